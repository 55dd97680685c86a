"""CPU-side checks of the C-ABI boundary: the library builds/loads, exports every symbol the header declares,
and its weight tables name exactly the reference's state-dict keys.  No compute calls (no GPU here)."""
import os
import re
from types import SimpleNamespace

import pytest
import torch

from ladiff_amd import _lib, schema
from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from ladiff_amd import build
    build.build()
    return _lib.lib()


def header_functions(name="ladiff_hip.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"#ifdef LADIFF_STAMPS.*?#endif", "", src, flags=re.S)      # the diagnostic twin's extra entries
    return sorted(set(re.findall(r"\b(ladiff_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree(lib):
    product, debug = header_functions(), header_functions("ladiff_hip_debug.h")
    assert not [n for n in product if n.startswith("ladiff_debug_")], "measurement switches belong in ladiff_hip_debug.h"
    assert all(n.startswith("ladiff_debug_") for n in debug)
    declared = sorted(product + debug)
    assert declared == sorted(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"


def test_library_exports_only_the_declared_entries(lib):
    """-fvisibility=hidden: `nm -D` shows the C ABI and nothing of the C++ inside (no ladiff:: symbols, no launch_* helpers)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = sorted(ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-2] in "TW")
    assert names == sorted(_lib.EXPORTS), sorted(set(names) ^ set(_lib.EXPORTS))[:10]


def test_version_and_errors(lib):
    assert lib.ladiff_version() == 6
    assert lib.ladiff_error_string(0) == b"ok"
    assert b"workspace" in lib.ladiff_error_string(-3)


def test_weight_tables_match_reference_schema(lib):
    den = _lib.param_names("denoiser")
    assert den == list(schema.denoiser_schema())            # same names, same order, all 360 tensors
    vs = schema.vae_schema(263)
    dec = _lib.param_names("decoder")
    assert sorted(dec) == sorted(schema.vae_decode_keys(vs))
    assert len(dec) == 175
    enc = _lib.param_names("encoder")
    assert len(enc) == 122 and set(enc) <= set(vs) and "skel_embedding.weight" in enc
    assert sorted(set(dec) | set(enc)) == sorted(vs)        # decode + encode tables cover all 297 tensors of the VAE


def test_clip_table_matches_transformers_key_names():
    clip = _lib.param_names("clip")
    sch = schema.clip_text_schema(49408, 12)
    assert len(clip) == 197 and sorted(clip) == sorted(sch)
    assert clip[:5] == ["text_model.embeddings.token_embedding.weight", "text_model.embeddings.position_embedding.weight",
                        "text_model.final_layer_norm.weight", "text_model.final_layer_norm.bias", "text_projection.weight"]
    assert sum(int(torch.Size(v).numel()) for v in sch.values()) == 123650304     # CLIPTextModelWithProjection, ViT-L/14


def test_t2m_evaluator_tables_match_checkpoint_keys():
    for kind, sch in (("t2m_movement", schema.t2m_movement_schema(259)), ("t2m_motion", schema.t2m_motion_schema()),
                      ("t2m_text", schema.t2m_text_schema())):
        assert sorted(_lib.param_names(kind)) == sorted(sch)
    from ladiff_amd import MotionEncoderBiGRUCo, MovementConvEncoder, TextEncoderBiGRUCo, synthetic as syn
    mv, mo, tx = syn.t2m_weights(263)
    MovementConvEncoder(259, 512, 512).load_state_dict(mv, strict=True)
    MotionEncoderBiGRUCo(512, 1024, 512).load_state_dict(mo, strict=True)
    TextEncoderBiGRUCo(300, 15, 512, 512).load_state_dict(tx, strict=True)
    with pytest.raises(NotImplementedError):
        MotionEncoderBiGRUCo(512, 512, 512)


def test_workspace_queries(lib):
    assert lib.ladiff_denoiser_tables_floats(50) == 50 * 9 * 1536
    assert lib.ladiff_denoiser_text_cache_floats(256, 50, 1) == 256 * 256 + 9 * 256 * 512 + 9 * 256 * 256 + 9 * 50 * 257 * 256
    assert lib.ladiff_denoiser_text_cache_floats(8, 50, 4) == 8 * 4 * 256 + 9 * 8 * 4 * 512 + 9 * 8 * 4 * 64 * 64
    assert lib.ladiff_reverse_workspace_bytes(128, 5, 50, 1) > 0
    # the c table is windowed: a 1000-step schedule needs about as much as a 50-step one, not 20x (2.4 GB at B = 128)
    assert lib.ladiff_reverse_workspace_bytes(128, 5, 1000, 1) < 1.3 * lib.ladiff_reverse_workspace_bytes(128, 5, 50, 1)
    assert lib.ladiff_decoder_workspace_bytes(128, 196, 5, 263) >= 128 * 196 * 4096 * 4


def test_block_plan_of_the_pipeline_loop(lib):
    """Host arithmetic of the pipeline loop's block plan (ladiff_reverse_plan): padded 32-row blocks hold both guidance branches
    of 3 prompts, the length-aware 16-row packing one branch of as many prompts as fit with only their valid latent rows."""
    import ctypes
    import math

    def plan(lengths, mode, masked=True, host=True, T=5, cfg=1):
        B = len(lengths)
        counts = (ctypes.c_int32 * B)(*[math.ceil(l / 48) for l in lengths])
        rows, nb = ctypes.c_int(0), ctypes.c_int(0)
        rc = lib.ladiff_reverse_plan(B, T, ctypes.cast(counts, ctypes.c_void_p) if host else None, int(masked), mode, 1, cfg,
                                     ctypes.byref(rows), ctypes.byref(nb))
        assert rc == 0
        return rows.value, nb.value

    uniform = [196] * 128
    assert plan(uniform, 3) == (32, 43)                       # ceil(128 / 3) blocks of 30 rows
    assert plan(uniform, 2) == (16, 86)                       # 43 groups of 3 prompts x 2 branches
    mixed = ([60, 120, 196] * 43)[:128]                       # latent counts 2 / 3 / 5
    rows, nb = plan(mixed, 2)
    assert rows == 16
    live = sum(math.ceil(l / 48) for l in mixed)
    assert 2 * math.ceil(live / 16) <= nb <= 62               # 427 valid rows per branch: >= 27 blocks, packing reaches 30
    assert plan(mixed, 1)[0] == 16 and plan(uniform, 1)[0] == 16          # the cost model takes the packed plan for both
    assert plan([196], 1) == (16, 2) and plan([196], 3) == (32, 1)    # one prompt: the 16-row plan's trip through the stages is shorter
    assert plan(mixed, 2, host=False) == (32, 43)              # counts on the device only: the packing needs them on the host
    assert plan([40] * 40, 2) == (16, 10)                      # one-row prompts: 8 per block (text K|V slots), 5 groups x 2 branches
    assert lib.ladiff_reverse_plan(0, 5, None, 0, 1, 1, 1, None, None) != 0
    # without guidance (ladiff.py:472-490): one-branch 16-row blocks, half as many; with device-only counts there is no plan
    assert plan(uniform, 1, cfg=0) == (16, 43) and plan(uniform, 3, cfg=0) == (16, 43)
    rows, nb = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.ladiff_reverse_plan(4, 5, None, 1, 1, 1, 0, ctypes.byref(rows), ctypes.byref(nb)) == -4


def test_product_library_has_no_garbage_result_kernels(lib):
    """the timing builds of the decoder kernels (results are garbage) live in the diagnostic twin only.  The product
    library rejects their selector values and exports no process-wide fault switch."""
    import ctypes
    for v in (0, 1, 2, 3):
        assert lib.ladiff_debug_set_mlp_variant(v) == 0
    for v in (-1, 4, 11, 17, 21, 26, 27):
        assert lib.ladiff_debug_set_mlp_variant(v) == -1
    assert lib.ladiff_debug_set_mlp_variant(0) == 0
    raw = ctypes.CDLL(lib._name)
    assert not hasattr(raw, "ladiff_debug_set_pipeline_fault") and not hasattr(raw, "ladiff_debug_set_stamps")
    assert hasattr(raw, "ladiff_sampler_set_fault")


def test_no_kernel_waits_for_an_lds_dma_stage_with_a_counted_vmcnt(lib):
    """LDS-DMA requests of one wave complete out of issue order when their latencies differ, so the only exact wait for an
    LDS-DMA stage is `s_waitcnt vmcnt(0)` with nothing younger in flight.  The shipped library's gfx950 code is disassembled and every
    kernel with a `global_load_lds` walked: a counted wait (N > 0) at which an LDS-DMA request may be outstanding is a failure
    (scripts/lds_dma_wait_lint.py; rounds 1 - 5's dec_mlp had 100+ of them, round 4's gemm_big two per K stage)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("lds_dma_wait_lint", os.path.join(ROOT, "scripts", "lds_dma_wait_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    bad, seen = lint.lint()
    assert seen >= 20, seen                                   # the walk found the LDS-DMA kernels at all
    assert not bad, {k: v[:4] for k, v in bad.items()}
    # ... and the rule is not vacuous: the classic two-stages-in-flight pattern is a finding
    assert lint.findings(["global_load_lds_dwordx4 v[0:1], off", "global_load_lds_dwordx4 v[2:3], off", "s_waitcnt vmcnt(1)"]) == [(2, 1)]
    # counted waits for plain loads with no LDS-DMA outstanding are fine; one whose loop's back edge carries an LDS-DMA request is not
    plain = ["global_load_dwordx4 v[4:7], v[0:1], off", "global_load_dwordx4 v[8:11], v[0:1], off", "s_waitcnt vmcnt(1)"]
    assert lint.findings(["global_load_lds_dwordx4 v[0:1], off", "s_waitcnt vmcnt(0)"] + plain) == []
    assert lint.findings(["s_waitcnt vmcnt(0)"] + plain + ["global_load_lds_dwordx4 v[0:1], off", "s_cbranch_scc1 65530"]) == []
    assert lint.findings(["s_waitcnt vmcnt(0)", "s_nop 0"] + plain + ["global_load_lds_dwordx4 v[0:1], off", "s_cbranch_scc1 65532"][:1] + plain) == [(8, 1)]


def test_argument_errors_do_not_touch_the_gpu(lib):
    assert lib.ladiff_layernorm(None, None, None, None, 4, None) == -1
    assert lib.ladiff_gemm(None, 0, None, 0, 0, None, 0, None, None, 0, None, None, None, 0, 1, 1, 32, 0, None) == -1


from ladiff_amd.schema import ABL, DEN_KW, VAE_KW  # noqa: E402,F401  (the shipped configuration; other tests import it from here)


def test_modules_have_reference_state_dict_schema():
    from ladiff_amd import LADiffDenoiser, LADiffVae, synthetic as syn
    den = LADiffDenoiser(ABL, **DEN_KW)
    sd = den.state_dict()
    assert len(sd) == 360 and sum(v.numel() for v in sd.values()) == 18426880
    den.load_state_dict(syn.denoiser_weights(), strict=True)
    vae = LADiffVae(ABL, **VAE_KW)
    sv = vae.state_dict()
    assert len(sv) == 297 and sum(v.numel() for v in sv.values()) == 18034183
    vae.load_state_dict(syn.vae_weights(263), strict=True)


def test_product_path_fails_loudly_without_gpu():
    from ladiff_amd import LADiffDenoiser, LADiffVae
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    den = LADiffDenoiser(ABL, **DEN_KW)
    with pytest.raises(_lib.LadiffHipError):
        den(torch.zeros(2, 5, 256), torch.tensor(1), torch.zeros(2, 1, 768), max_iter_elements=torch.tensor([5, 5]))
    vae = LADiffVae(ABL, **VAE_KW)
    with pytest.raises(_lib.LadiffHipError):
        vae.decode(torch.zeros(5, 2, 256), [60, 60])
    with pytest.raises(_lib.LadiffHipError):
        vae.encode(torch.zeros(2, 60, 263), [60, 60])


def test_unbuilt_config_branches_raise():
    from ladiff_amd import LADiffDenoiser
    with pytest.raises(TypeError):
        LADiffDenoiser(ABL, **{**DEN_KW, "condition": "action"})
    with pytest.raises(ValueError):
        LADiffDenoiser(ABL, **{**DEN_KW, "arch": "trans_dec2"})
    with pytest.raises(NotImplementedError):
        LADiffDenoiser(ABL, **{**DEN_KW, "num_layers": 7})


def test_scheduler_tables():
    from ladiff_amd import DDIMScheduler, DDPMScheduler
    from oracle import ladiff_oracle as orc
    kw = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
              clip_sample=False)
    d = DDIMScheduler(set_alpha_to_one=False, steps_offset=1, **kw)
    d.set_timesteps(50)
    o = orc.DDIM(); o.set_timesteps(50)
    assert torch.equal(d.timesteps, o.timesteps)
    assert torch.equal(d.alphas_cumprod, o.alphas_cumprod)
    tab = d.coef_table(0.0)
    assert tab.shape == (50, 8) and not d.needs_noise(0.0) and d.needs_noise(0.5)
    # the table reproduces the oracle's step on CPU numbers
    x = torch.randn(3, 5, 256); e = torch.randn(3, 5, 256)
    for i in (0, 17, 49):
        sa, sb, kx0, kx, ke, kn = tab[i, :6]
        got = kx0 * ((x - sb * e) / sa) + kx * x + ke * e
        assert torch.allclose(got, o.step(e, d.timesteps[i], x), atol=1e-5, rtol=1e-5)
    p = DDPMScheduler(variance_type="fixed_small", **kw)
    p.set_timesteps(1000)
    q = orc.DDPM(); q.set_timesteps(1000)
    assert torch.equal(p.timesteps, q.timesteps) and p.needs_noise()
    tab = p.coef_table()
    z = torch.randn(3, 5, 256)
    for i in (0, 500, 999):
        sa, sb, kx0, kx, ke, kn = tab[i, :6]
        got = kx0 * ((x - sb * e) / sa) + kx * x + ke * e + kn * z
        assert torch.allclose(got, q.step(e, p.timesteps[i], x, noise=z), atol=1e-5, rtol=1e-5)
    with pytest.raises(NotImplementedError):
        DDIMScheduler(clip_sample=True)
    with pytest.raises(NotImplementedError):      # diffusers' default is clip_sample=True: an omitted key must not sample silently
        DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")
    # DDPM with fewer inference steps than training steps: the two diffusers generations differ, both are restated
    for mode in ("t-1", "schedule"):
        p10 = DDPMScheduler(variance_type="fixed_small", prev_timestep=mode, **kw); p10.set_timesteps(10)
        q10 = orc.DDPM(prev_timestep=mode); q10.set_timesteps(10)
        tab = p10.coef_table()
        for i in (0, 5, 9):
            sa, sb, kx0, kx, ke, kn = tab[i, :6]
            got = kx0 * ((x - sb * e) / sa) + kx * x + ke * e + kn * z
            assert torch.allclose(got, q10.step(e, p10.timesteps[i], x, noise=z), atol=1e-5, rtol=1e-5)
    a = DDPMScheduler(variance_type="fixed_small", prev_timestep="t-1", **kw); a.set_timesteps(1000)
    b = DDPMScheduler(variance_type="fixed_small", prev_timestep="schedule", **kw); b.set_timesteps(1000)
    assert torch.allclose(a.coef_table(), b.coef_table(), atol=1e-4, rtol=1e-3)     # same schedule at N = 1000 (1 - a_t/a_prev vs betas[t]: rounding only)
    a.set_timesteps(10); b.set_timesteps(10)
    assert not torch.allclose(a.coef_table(), b.coef_table(), atol=1e-3)


def test_host_sinusoid_matches_oracle():
    from ladiff_amd.schedulers import timestep_sinusoid
    from oracle import ladiff_oracle as orc
    t = torch.tensor([981, 481, 1])
    assert torch.equal(timestep_sinusoid(t), orc.timestep_sinusoid(t))


def test_both_split_flavours_are_built_and_say_what_they_are(lib):
    """build_all(): libladiff_hip.so carries split operands as fp16 pairs, libladiff_hip_bf16.so as bf16 pairs; same exports."""
    import ctypes
    import subprocess
    from ladiff_amd import build
    build.build_all()
    assert lib.ladiff_split_format() == 1
    other = ctypes.CDLL(_lib.LIB_PATH_BF16)
    other.ladiff_split_format.restype = ctypes.c_int
    assert other.ladiff_split_format() == 0 and other.ladiff_version() == lib.ladiff_version()
    syms = lambda p: sorted(l.split()[-1] for l in subprocess.run(["nm", "-D", "--defined-only", p], capture_output=True, text=True).stdout.splitlines()
                            if " T " in l)
    assert syms(_lib.LIB_PATH) == syms(_lib.LIB_PATH_BF16) == syms(build.diag_lib())   # + the loop kernel's hand-off diagnostic build
    with pytest.raises(ValueError):
        _lib.select_split_format("fp8")
    with pytest.raises(_lib.LadiffHipError):            # this process has loaded the fp16 library already
        _lib.select_split_format("bf16")

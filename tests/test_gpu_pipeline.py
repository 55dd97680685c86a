"""The persistent pipeline loop (csrc/systolic.hip) against the launch-per-stage loop: same arithmetic per row, different
summation order of the MLP partials only.  The end-to-end goldens / oracle tests in test_gpu_path.py run through the pipeline
as well (it is the default loop); these tests pin the two loop forms to each other over the shapes the block geometry
depends on (prompts per block, latent rows per prompt, partial last block) and check the kernel's status word."""
import pytest
import torch

from ladiff_amd import LADIFF, DDIMScheduler, DDPMScheduler, LADiffDenoiser, LADiffVae, synthetic as syn
from test_abi import ABL, DEN_KW, VAE_KW

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SCHED_KW = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False)


@pytest.fixture(scope="module")
def nets():
    den = LADiffDenoiser(ABL, **DEN_KW); den.load_state_dict(syn.denoiser_weights(), strict=True)
    vae = LADiffVae(ABL, **VAE_KW); vae.load_state_dict(syn.vae_weights(263), strict=True)
    return den.to(DEV).eval(), vae.to(DEV).eval()


def run(nets, loop, precision, B, T, steps, lens, sched="ddim", step_noise=None, guidance=7.5):
    den, vae = nets
    s = (DDIMScheduler(set_alpha_to_one=False, steps_offset=1, **SCHED_KW) if sched == "ddim"
         else DDPMScheduler(variance_type="fixed_small", **SCHED_KW))
    pipe = LADIFF(denoiser=den, vae=vae, scheduler=s, guidance_scale=guidance, num_inference_timesteps=steps, eta=0.0, max_it=T,
                  precision=precision, loop=loop)
    text = syn.text_embeddings(B, seed=900 + B).to(DEV)
    if guidance <= 1.0:
        text = text[B:].contiguous()                               # no guidance: the conditional rows only (ladiff.py:472-490)
    noise = torch.randn(B, T, 256, generator=torch.Generator().manual_seed(B * 10 + T)).to(DEV)
    z = pipe._diffusion_reverse(text, lens, init_noise=noise, step_noise=step_noise)
    code, info = pipe.loop_status()
    assert (code, info) == (0, 0), f"pipeline kernel aborted: code {code}, workgroup {info}"
    assert pipe.last_loop()[0] == (loop != "launches")
    return z


@pytest.mark.parametrize("loop", ["pipeline32", "pipeline16", "pipeline"])
@pytest.mark.parametrize("precision,tol", [("f16x3", 2e-4), ("fp32", 1e-5)])
@pytest.mark.parametrize("B,T", [(1, 5), (2, 5), (3, 5), (4, 5), (7, 5), (43, 5), (5, 1), (9, 2), (6, 3), (5, 8)])
def test_pipeline_matches_launches(nets, loop, precision, tol, B, T):
    """32-row blocks of P = 32 / (2 T) prompts (one block, several, a partial last block), the length-aware 16-row packing,
    and whichever of the two the sampler picks by itself; every latent count the tiles allow."""
    lens = [max(1, min(196, 48 * ((i % T) + 1) - 5 * (i % 3))) for i in range(B)]          # latent counts 1..T, mixed
    za = run(nets, "launches", precision, B, T, 6, lens)
    zb = run(nets, loop, precision, B, T, 6, lens)
    scale = max(1.0, za.abs().max().item())
    assert torch.isfinite(zb).all()
    assert (za - zb).abs().max().item() < tol * scale
    for i, l in enumerate(lens):                                   # rows past a motion's latent count: exact zeros in both
        c = -(-l // 48)
        if c < T:
            assert zb[c:, i].abs().max().item() == 0


def test_pipeline16_variant_and_replay(nets):
    """The 16-row-block variant (one guidance branch of three prompts per block, the tails join the branches) gives the same
    rows as the 32-row blocks - the arithmetic per row is identical - and a replay is bit-identical."""
    lens = [196, 60, 120, 100, 48, 150, 196, 30, 77, 196, 13]
    z32 = run(nets, "pipeline", "f16x3", 11, 5, 8, lens)
    z16 = run(nets, "pipeline16", "f16x3", 11, 5, 8, lens)
    assert torch.equal(z32, z16)
    assert torch.equal(z32, run(nets, "pipeline", "f16x3", 11, 5, 8, lens))


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
@pytest.mark.parametrize("name,lens", [
    ("eight one-row prompts per block", [40] * 40),
    ("ten-row last block", [196] * 128),
    ("c5 per-rank mix", ([60, 120, 196] * 43)[:128]),
    ("all counts", ([196, 60, 120, 100, 48, 150, 196] * 19)[:128]),
    ("172 blocks: the partial planes' ring wraps off a slot boundary at every step", [196] * 256),
    ("an odd number of blocks", [196] * 127 + [60, 100]),
])
def test_packed_blocks_many(nets, precision, name, lens):
    """Length-aware packing at sizes where the stages are backlogged (blocks prefetched, flags deferred) and blocks leave
    whole waves idle in the attention stage: the packed plan is bit-identical to the 32-row plan, and both match the
    launch-per-stage loop."""
    B = len(lens)
    za = run(nets, "launches", precision, B, 5, 4, lens)
    z32 = run(nets, "pipeline32", precision, B, 5, 4, lens)
    z16 = run(nets, "pipeline16", precision, B, 5, 4, lens)
    assert torch.equal(z32, z16), name
    assert (za - z16).abs().max().item() < (2e-4 if precision == "f16x3" else 1e-5) * max(1.0, za.abs().max().item())


def test_ragged_decode_matches_padded_pass(nets):
    """LADiffVae.decode on a mixed-length batch computes only the valid frames (ragged rows, ladiff_vae_decode_ragged);
    against the padded pass: same frames on the valid rows, exact zeros after each motion's length."""
    _, vae = nets
    old = vae.precision
    for lens in (([60, 120, 196, 33, 150] * 13)[:64], [1, 196, 2, 47, 48, 49, 195], [17]):
        B = len(lens)
        z = torch.randn(5, B, 256, generator=torch.Generator().manual_seed(3)).to(DEV)
        for i, l in enumerate(lens):
            z[-(-l // 48):, i] = 0
        for precision, tol in (("fp32", 1e-5), ("f16x3", 2e-5)):
            vae.precision = precision
            vae.length_aware = False
            one = vae.decode(z, lens)
            vae.length_aware = True
            many = vae.decode(z, lens)
            assert one.shape == many.shape == (B, max(lens), 263)
            assert (one - many).abs().max().item() < tol * max(1.0, one.abs().max().item())
            for i, l in enumerate(lens):
                if l < max(lens):
                    assert many[i, l:].abs().max().item() == 0
    vae.precision = old


def test_stage_workgroups_of_four_and_eight_waves_agree(nets):
    """The 16-row plan runs its stage workgroups with two waves per SIMD (each stage's weight slice split over them); the
    one-wave-per-SIMD form of the same kernel is kept behind a measurement switch: identical bits."""
    from ladiff_amd import _lib
    lens = [196, 60, 120, 100, 48, 150, 196, 30, 77, 196, 13]
    L = _lib.lib()
    try:
        for precision in ("f16x3", "fp32"):
            assert L.ladiff_debug_set_stage_waves(2) == 0
            z8 = run(nets, "pipeline16", precision, 11, 5, 6, lens)
            assert L.ladiff_debug_set_stage_waves(1) == 0
            z4 = run(nets, "pipeline16", precision, 11, 5, 6, lens)
            assert torch.equal(z8, z4), precision
        assert L.ladiff_debug_set_stage_waves(3) != 0
    finally:
        L.ladiff_debug_set_stage_waves(2)


def test_xcd_placement_on_and_off_agree(nets):
    """The stage table is dealt to the XCDs in chain order and stages whose readers share their XCD store plainly (through that
    XCD's L2); with the placement switched off every hand-off writes through as before.  Same bits either way, in both plans and
    both arithmetic modes, on batches with more blocks than the partial planes' ring has slots (back-pressure active) - and the
    same bits when the call is repeated (the ring's slot sequence runs through the steps)."""
    from ladiff_amd import _lib
    L = _lib.lib()
    lens = [196] * 70 + [60, 120, 49, 1, 100, 150, 196, 48, 30, 77]
    try:
        for precision in ("f16x3", "fp32"):
            for loop in ("pipeline16", "pipeline32"):
                assert L.ladiff_debug_set_xcd_local(1) == 0
                za = run(nets, loop, precision, len(lens), 5, 7, lens)
                zb = run(nets, loop, precision, len(lens), 5, 7, lens)
                assert L.ladiff_debug_set_xcd_local(0) == 0
                zc = run(nets, loop, precision, len(lens), 5, 7, lens)
                assert torch.equal(za, zb) and torch.equal(za, zc), (precision, loop)
        assert L.ladiff_debug_set_xcd_local(3) != 0
        # one workgroup somewhere else than planned: the launch agrees to write through everywhere (status info -1), same bits
        den, vae = nets
        for precision in ("f16x3", "fp32"):
            outs = []
            for mode in (1, 2):
                assert L.ladiff_debug_set_xcd_local(mode) == 0
                pipe = LADIFF(denoiser=den, vae=vae, scheduler=DDIMScheduler(set_alpha_to_one=False, steps_offset=1, **SCHED_KW),
                              guidance_scale=7.5, num_inference_timesteps=5, eta=0.0, max_it=5, precision=precision, loop="pipeline16")
                text = syn.text_embeddings(len(lens), seed=31).to(DEV)
                noise = torch.randn(len(lens), 5, 256, generator=torch.Generator().manual_seed(32)).to(DEV)
                outs.append(pipe._diffusion_reverse(text, lens, init_noise=noise))
                assert pipe.loop_status() == ((0, 0) if mode == 1 else (0, -1)), (precision, mode, pipe.loop_status())
            assert torch.equal(outs[0], outs[1]), precision
    finally:
        L.ladiff_debug_set_xcd_local(1)


@pytest.mark.slow
@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_repeated_full_size_calls_are_identical(nets, precision):
    """The loop is deterministic: the benchmark-size call (84 length-aware blocks, 50 steps - the partial planes' 16-slot rings
    wrap off a slot boundary at every step) gives the same bits every time, alternating with a small batch on the same
    sampler.  (A producer running ahead over the step boundary once overwrote a ring slot its consumer had not read: only
    the last blocks of a step were affected, a few calls in a hundred, in the slower arithmetic mode.)"""
    den, vae = nets
    lens = [196] * 120 + [60, 120, 49, 1, 100, 150, 196, 48]
    sub = [196, 60, 120, 49, 1, 100]
    pipe = LADIFF(denoiser=den, vae=vae, scheduler=DDIMScheduler(set_alpha_to_one=False, steps_offset=1, **SCHED_KW),
                  guidance_scale=7.5, num_inference_timesteps=50, eta=0.0, max_it=5, precision=precision, loop="pipeline")
    text, noise = syn.text_embeddings(128, seed=5).to(DEV), syn.init_noise(lens, seed=6).to(DEV)
    text_s, noise_s = syn.text_embeddings(len(sub), seed=7).to(DEV), syn.init_noise(sub, seed=8).to(DEV)
    ref = ref_s = None
    for it in range(100):                                          # a soak: ~4 s per arithmetic mode
        z = pipe._diffusion_reverse(text, lens, init_noise=noise)
        assert pipe.loop_status() == (0, 0)
        zs = pipe._diffusion_reverse(text_s, sub, init_noise=noise_s)
        assert pipe.loop_status() == (0, 0)
        if ref is None:
            ref, ref_s = z.clone(), zs.clone()
        assert torch.equal(z, ref) and torch.equal(zs, ref_s), (precision, it)


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_old_step_graph_is_not_replayed_after_other_plans(nets, precision):
    """A hipGraph is replayed only while it is the newest graph instantiation of the process (api.hip, g_graph_epoch).  The sequence
    that failed before: launch-per-stage loop at 200 prompts (its ~150-node step graph instantiated), a blocking status read, two
    other batch shapes (two more samplers, their graphs, status reads), the first shape again - the replay of the FIRST step graph
    dispatched kernels with garbage pointers (memory access fault; the same sequence without graphs was clean, every captured
    pointer was alive).  Now the first shape captures again: same bits as its first run, for the three loop forms."""
    den, vae = nets
    pipe = LADIFF(denoiser=den, vae=vae, scheduler=DDIMScheduler(set_alpha_to_one=False, steps_offset=1, **SCHED_KW),
                  guidance_scale=7.5, num_inference_timesteps=2, eta=0.0, max_it=5, precision=precision, loop="launches")
    data = {B: (syn.text_embeddings(B, seed=B).to(DEV), syn.init_noise([196] * B, seed=B + 1).to(DEV)) for B in (200, 32, 16, 321)}

    def call(B, loop):
        pipe.loop = loop
        z = pipe._diffusion_reverse(data[B][0], [196] * B, init_noise=data[B][1])
        assert pipe.loop_status() == (0, 0)                       # a blocking device-to-host copy, as in the failing sequence
        return z

    first = call(200, "launches")
    call(32, "launches"); call(16, "launches")
    assert torch.equal(call(200, "launches"), first)
    # the same with the other shapes run by the pipeline kernel, a chunked batch (two samplers alternating) among them
    call(32, "pipeline16"); call(321, "pipeline16"); call(16, "pipeline32")
    assert torch.equal(call(200, "launches"), first)
    a = call(321, "pipeline16")
    call(200, "launches"); call(32, "pipeline16")
    assert torch.equal(call(321, "pipeline16"), a)


def test_two_samplers_on_two_streams(nets):
    """A pipeline kernel needs the whole chip resident: launches from different streams of one process are chained through an
    event (systolic.hip), so two samplers enqueued back to back on two streams both complete with the right result."""
    den, vae = nets
    lens = [196, 60, 120, 100, 48, 150, 196, 30]
    B = len(lens)
    text = syn.text_embeddings(B, seed=77).to(DEV)
    noise = torch.randn(B, 5, 256, generator=torch.Generator().manual_seed(5)).to(DEV)
    sched = lambda: DDIMScheduler(set_alpha_to_one=False, steps_offset=1, **SCHED_KW)
    ref = LADIFF(denoiser=den, vae=vae, scheduler=sched(), guidance_scale=7.5, num_inference_timesteps=20, eta=0.0, max_it=5,
                 precision="f16x3", loop="launches")._diffusion_reverse(text, lens, init_noise=noise)
    pipes = [LADIFF(denoiser=den, vae=vae, scheduler=sched(), guidance_scale=7.5, num_inference_timesteps=20, eta=0.0, max_it=5,
                    precision="f16x3", loop="pipeline") for _ in range(2)]
    streams = [torch.cuda.Stream(device=DEV) for _ in range(2)]
    torch.cuda.synchronize()
    outs = []
    for _ in range(3):                                             # interleaved launches, nothing synchronises in between
        for pipe, st in zip(pipes, streams):
            with torch.cuda.stream(st):
                outs.append(pipe._diffusion_reverse(text, lens, init_noise=noise))
    torch.cuda.synchronize()
    for pipe in pipes:
        assert pipe.loop_status() == (0, 0)
    for z in outs:
        assert (z - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())


def test_pipeline_ddpm_windows(nets):
    """A 200-step DDPM schedule runs as four 50-step windows (the c table is rebuilt per window, the latents carry over),
    with the per-step noise stream: pipeline == launches."""
    B, T, n = 5, 5, 200
    lens = [196, 60, 120, 100, 48]
    sn = syn.ddpm_noise(n, B, seed=5).to(DEV)
    za = run(nets, "launches", "f16x3", B, T, n, lens, sched="ddpm", step_noise=sn)
    zb = run(nets, "pipeline", "f16x3", B, T, n, lens, sched="ddpm", step_noise=sn)
    assert (za - zb).abs().max().item() < 5e-4 * max(1.0, za.abs().max().item())


def test_window_timing_switch_reaches_plans_made_later(nets):
    """`window_ms(enable=True)` before any plan exists: the sampler of the plan the next call creates still times its windows."""
    den, vae = nets
    B, T, n = 3, 5, 200                                             # four windows of 50 steps
    lens = [196, 60, 120]
    pipe = LADIFF(denoiser=den, vae=vae, scheduler=DDPMScheduler(variance_type="fixed_small", **SCHED_KW), guidance_scale=7.5,
                  num_inference_timesteps=n, eta=0.0, max_it=T, precision="f16x3", loop="pipeline")
    pipe.window_ms(enable=True)
    text, noise = syn.text_embeddings(B, seed=3), syn.init_noise(lens, seed=4)
    pipe._diffusion_reverse(text.to(DEV), lens, init_noise=noise.to(DEV), noise_seed=5)
    total = pipe.loop_ms()
    ms, windows = pipe.window_ms()
    assert windows == 4 and 0.0 < ms <= total * 1.05


# ---------------------------------------------------------------- an abandoned pipeline launch
def _fault_pipe(nets, **kw):
    den, vae = nets
    return LADIFF(denoiser=den, vae=vae, scheduler=DDIMScheduler(set_alpha_to_one=False, steps_offset=1, **SCHED_KW),
                  guidance_scale=7.5, num_inference_timesteps=8, eta=0.0, max_it=5, precision="f16x3", loop="pipeline", **kw)


def test_aborted_pipeline_launch_raises_poisons_and_recovers(nets):
    """One stage workgroup never publishes (test switch) and the waits are bounded to 20 ms: the loop is abandoned.  The product
    path must not hand back partial latents: z is NaN, `check()` / the next call / `loop_ms()` raise LadiffHipError, the status is
    sticky over a windowed schedule, and the next call - fault removed - is clean and bit-identical to a run that never failed."""
    from ladiff_amd import _lib
    L = _lib.lib()
    lens = [196, 60, 120, 100, 48, 150, 196, 30, 77, 196, 13]
    B = len(lens)
    text, noise = syn.text_embeddings(B, seed=41), syn.init_noise(lens, seed=42)
    clean = _fault_pipe(nets)
    good = clean._diffusion_reverse(text.to(DEV), lens, init_noise=noise.to(DEV))
    pipe = _fault_pipe(nets)
    fb = wp = None
    try:
        pipe.set_pipeline_fault(17, 20)                                # workgroup 17 of 255 leaves at once; waits time out after 20 ms
        z = pipe._diffusion_reverse(text.to(DEV), lens, init_noise=noise.to(DEV))
        assert torch.isnan(z).all()                                    # poisoned, not "plausible garbage"
        # the fault is a property of THAT object's samplers: another one in the same process runs clean meanwhile
        assert torch.equal(clean._diffusion_reverse(text.to(DEV), lens, init_noise=noise.to(DEV)), good)
        clean.check()
        with pytest.raises(_lib.LadiffHipError, match="abandoned"):
            pipe.check()
        pipe.check()                                                   # reported once
        # later calls notice by themselves, without the host ever blocking on a loop that is still running: the next call looks only
        # if the status words have arrived already (the aborting loop takes its 20 ms timeout), the one after it waits for them
        td, nd = text.to(DEV), noise.to(DEV)
        z = pipe._diffusion_reverse(td, lens, init_noise=nd)
        pipe.set_pipeline_fault(-1, 0)
        raised = 0
        for _ in range(2):
            try:
                pipe._diffusion_reverse(td, lens, init_noise=nd)
            except _lib.LadiffHipError:
                raised += 1
        assert raised == 1                                             # reported by the next call or the one after it, once
        pipe.check()                                                   # the clean call(s) in between: nothing to report
        pipe.set_pipeline_fault(17, 20)
        z = pipe._diffusion_reverse(text.to(DEV), lens, init_noise=noise.to(DEV))
        torch.cuda.synchronize()                                       # status arrived: the very next call reports it
        with pytest.raises(_lib.LadiffHipError):
            pipe._diffusion_reverse(text.to(DEV), lens, init_noise=noise.to(DEV))
        with pytest.raises(_lib.LadiffHipError):                       # sample() hands out checked frames only
            pipe.sample(text.to(DEV), lens, init_noise=noise.to(DEV))
        # fallback=True: the same call is re-run launch-per-stage in this process and matches the oracle
        fb = _fault_pipe(nets, fallback=True)
        fb.set_pipeline_fault(17, 20)
        zf = fb._diffusion_reverse(text.to(DEV), lens, init_noise=noise.to(DEV))
        assert fb.fallback_count == 1 and torch.isfinite(zf).all()
        assert (zf - good).abs().max().item() < 2e-4 * max(1.0, good.abs().max().item())     # launches vs pipeline: summation order only
        # a windowed schedule (200 DDPM steps = 4 launches): the abort of the first window is not erased by the later ones
        sn = syn.ddpm_noise(200, B, seed=5).to(DEV)
        den, vae = nets
        wp = LADIFF(denoiser=den, vae=vae, scheduler=DDPMScheduler(variance_type="fixed_small", **SCHED_KW), guidance_scale=7.5,
                    num_inference_timesteps=200, eta=0.0, max_it=5, precision="f16x3", loop="pipeline")
        wp.set_pipeline_fault(17, 20)
        zw = wp._diffusion_reverse(text.to(DEV), lens, init_noise=noise.to(DEV), step_noise=sn)
        assert torch.isnan(zw).all() and wp.loop_status()[0] == 2
        with pytest.raises(_lib.LadiffHipError):
            wp.check()
    finally:
        for o in (pipe, fb, wp):
            if o is not None:
                o.set_pipeline_fault(-1, 0)
    # fault removed: clean, and the same bits as the run that never failed
    z2 = pipe._diffusion_reverse(text.to(DEV), lens, init_noise=noise.to(DEV))
    pipe.check()
    assert pipe.loop_status() == (0, 0) and torch.equal(z2, good)
    assert fb.fallback_count == 1
    zf2 = fb._diffusion_reverse(text.to(DEV), lens, init_noise=noise.to(DEV))
    assert fb.fallback_count == 1 and torch.equal(zf2, good)


def test_fallback_result_matches_oracle(nets):
    """The in-process fallback (launch-per-stage re-run of an abandoned call) against the CPU oracle on the decoded frames."""
    from ladiff_amd import _lib
    from oracle import ladiff_oracle as orc
    L = _lib.lib()
    lens = [196, 60, 130]
    text, noise = syn.text_embeddings(3, seed=43), syn.init_noise(lens, seed=44)
    fb = _fault_pipe(nets, fallback=True)
    try:
        fb.set_pipeline_fault(100, 20)
        z, feats = fb.sample(text.to(DEV), lens, init_noise=noise.to(DEV))
    finally:
        fb.set_pipeline_fault(-1, 0)
    assert fb.fallback_count == 1
    z_o, f_o = orc.sample_motions(syn.denoiser_weights(), syn.vae_weights(263), text, lens, noise, 8, "ddim")
    assert (feats.cpu() - f_o).abs().max().item() < 1e-3


@pytest.mark.parametrize("precision,tol", [("f16x3", 2e-4), ("fp32", 1e-5)])
@pytest.mark.parametrize("B,T", [(1, 5), (3, 5), (7, 5), (40, 5), (130, 5), (9, 2), (5, 8)])
def test_pipeline_without_guidance_matches_launches(nets, precision, tol, B, T):
    """guidance_scale <= 1 (ladiff.py:472-490): the network sees the B latents once.  The pipeline runs it as one-branch 16-row
    blocks whose tail unit is the block itself; against the launch-per-stage loop over the block geometries (one block, a partial
    last block, more blocks than ring slots, every latent count)."""
    lens = [max(1, min(196, 48 * ((i % T) + 1) - 5 * (i % 3))) for i in range(B)]
    za = run(nets, "launches", precision, B, T, 6, lens, guidance=1.0)
    zb = run(nets, "pipeline", precision, B, T, 6, lens, guidance=1.0)
    assert torch.isfinite(zb).all()
    assert (za - zb).abs().max().item() < tol * max(1.0, za.abs().max().item())
    for i, l in enumerate(lens):
        c = -(-l // 48)
        if c < T:
            assert zb[c:, i].abs().max().item() == 0


# ---------------------------------------------------------------- round 4: the tagged hand-off against the flag protocol
def _with_handoff(tagged, fn):
    from ladiff_amd import _lib
    L = _lib.lib()
    assert L.ladiff_debug_set_handoff(1 if tagged else 0) == 0
    try:
        return fn()
    finally:
        L.ladiff_debug_set_handoff(1)


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
@pytest.mark.parametrize("B,T,steps,guidance", [
    (1, 5, 3, 7.5), (3, 5, 2, 7.5), (7, 5, 5, 7.5), (43, 5, 6, 7.5),      # one unit, padding rows in every tile, an odd step count
    (5, 1, 4, 7.5), (9, 2, 4, 7.5), (5, 8, 3, 7.5),                        # 1 ... 8 latent rows per prompt: blocks with idle waves and 16 live rows
    (130, 5, 4, 7.5),                                                     # 88 blocks: ring back-pressure and look-ahead are on
    (7, 5, 5, 1.0), (40, 5, 3, 1.0),                                      # no guidance: a block is a unit of its own
])
def test_tagged_handoff_gives_the_flag_protocol_s_bits(nets, precision, B, T, steps, guidance):
    """The parity-tagged hand-off (rows carry the step's parity in the last mantissa bit of every word, consumers load until every word
    shows it: no drain, no flag, no poll round trip) must give the SAME BITS as the flag protocol: both store the canonical value (bit
    cleared) and sum in the same order.  A consumer that ever computed on a stale, torn or half-written word would differ."""
    lens = [max(1, min(196, 48 * ((i % T) + 1) - 5 * (i % 3))) for i in range(B)]
    zf = _with_handoff(False, lambda: run(nets, "pipeline16", precision, B, T, steps, lens, guidance=guidance))
    zt = _with_handoff(True, lambda: run(nets, "pipeline16", precision, B, T, steps, lens, guidance=guidance))
    assert torch.isfinite(zt).all() and torch.equal(zf, zt)


def test_tagged_handoff_ddpm_windows_and_replays(nets):
    """A 200-step DDPM schedule = four launches of 50 steps (the hand-off buffers start every launch at parity 1), then ten replays of a
    50-step DDIM call: identical bits to the flag protocol, identical from call to call."""
    B, T = 11, 5
    lens = [196, 60, 120, 100, 48, 150, 196, 30, 77, 196, 13]
    sn = syn.ddpm_noise(200, B, seed=5).to(DEV)
    zf = _with_handoff(False, lambda: run(nets, "pipeline16", "f16x3", B, T, 200, lens, sched="ddpm", step_noise=sn))
    zt = _with_handoff(True, lambda: run(nets, "pipeline16", "f16x3", B, T, 200, lens, sched="ddpm", step_noise=sn))
    assert torch.equal(zf, zt)
    first = run(nets, "pipeline16", "f16x3", B, T, 50, lens)
    for _ in range(10):
        assert torch.equal(run(nets, "pipeline16", "f16x3", B, T, 50, lens), first)


def test_tagged_handoff_full_batch_soak(nets):
    """128 prompts x 50 steps x 30 calls through the tagged hand-off (86 blocks x 59 hops x 50 steps x 30 = 7.6 million tile hand-offs, ~1.2e11
    tagged words) while alternating with the flag protocol: every call bit-identical to the first."""
    B, T = 128, 5
    lens = [196] * B
    ref = _with_handoff(False, lambda: run(nets, "pipeline16", "f16x3", B, T, 50, lens))
    for i in range(30):
        z = _with_handoff(i % 5 != 4, lambda: run(nets, "pipeline16", "f16x3", B, T, 50, lens))
        assert torch.equal(z, ref), f"call {i} differs"


@pytest.mark.parametrize("B,calls", [(256, 20), (1024, 12)])
def test_tagged_handoff_many_blocks_of_mixed_lengths(nets, B, calls):
    """Round 6: 256 / 1,024 prompts of 2, 3 and 5 latents (114 / 452 blocks: config c5's shape) - launches in which the attention
    stages run dozens of blocks ahead of the MLP stages.  RED2 has no barrier, and a wave whose row is padding in block after block (row 15
    of every 15-row block) used to be held back by its ticket row only, i.e. by OUT, not by LIN: more than a ring (16 blocks) ahead of LIN,
    its next live row - the first blocks of the next step, which have 16 live rows - found a ring slot two uses old, whose one-bit tag is
    the one expected, and summed another block's partial products.  Seen as single prompts differing from call to call once QKV / OUT had
    become faster than the MLP stages (every call at 1,024 prompts).  Every call must give the flag protocol's bits."""
    lens = syn.mixed_lengths(B)
    ref = _with_handoff(False, lambda: run(nets, "pipeline16", "f16x3", B, 5, 50, lens))
    for i in range(calls):
        z = _with_handoff(True, lambda: run(nets, "pipeline16", "f16x3", B, 5, 50, lens))
        bad = torch.nonzero((z - ref).abs().amax(dim=(0, 2))).flatten().tolist()                 # latents are [T, B, 256]
        assert not bad, f"call {i}: prompts {bad[:12]} differ from the flag protocol's result"


def test_no_stage_ever_looks_at_a_slot_more_than_one_use_old():
    """The reader's side of the tagged hand-off, checked by the kernel itself: the diagnostic build of the loop kernel (libladiff_hip_diag.so:
    4-bit generation tags, every look that a one-bit tag would have accepted from another generation counted per stage type, accepted rows
    reloaded and compared) over many mixed-length blocks, padding in every tile, one prompt, no guidance - zero events.  With round 5's
    RED2 this records hundreds of looks per call at 1,024 prompts (profiles/r6/22g_*).  Child process: one library per process."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "handoff_diag_check.py"), "quick"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "hand-off diagnostics: clean" in r.stdout, (r.stdout + r.stderr)[-1500:]


# ---------------------------------------------------------------- round 4: measurement switches of the pipeline that must not change a bit
@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_stage_plan_and_poll_pause_switches_keep_the_bits(nets, precision):
    """`ladiff_debug_set_stage_plan(1)` (OUT as two workgroups on alternating blocks, STYL as one group: 246 workgroups instead of 255)
    and `ladiff_debug_set_poll_pause` (waves of chosen stage types rest between polls) re-deal work and change timing only: the
    latents are the same bits - in both hand-off protocols, on a batch with ring back-pressure (88 blocks)."""
    from ladiff_amd import _lib
    L = _lib.lib()
    B, T, steps = 130, 5, 3
    lens = [max(1, min(196, 48 * ((i % T) + 1) - 5 * (i % 3))) for i in range(B)]
    ref = run(nets, "pipeline16", precision, B, T, steps, lens)
    try:
        assert L.ladiff_debug_set_stage_plan(2) != 0 and L.ladiff_debug_set_poll_pause(256, 1) != 0      # rejected
        assert L.ladiff_debug_set_stage_plan(1) == 0
        assert torch.equal(run(nets, "pipeline16", precision, B, T, steps, lens), ref)                    # a new LADIFF object: new stage table
        assert torch.equal(_with_handoff(False, lambda: run(nets, "pipeline16", precision, B, T, steps, lens)), ref)
        assert L.ladiff_debug_set_stage_plan(0) == 0
        assert L.ladiff_debug_set_poll_pause(1 | 2 | 4 | 8 | 16 | 32 | 64, 3) == 0
        assert L.ladiff_debug_set_stage_delay(1 | 2 | 4 | 8 | 64, 2) == 0
        assert torch.equal(run(nets, "pipeline16", precision, B, T, steps, lens), ref)
        for eighths, mask in ((0, 0), (8, 4), (6, 255)):                 # the pacing of STYL's polling off, at its longest, for every type asked
            assert L.ladiff_debug_set_pacing(eighths, mask) == 0
            assert torch.equal(run(nets, "pipeline16", precision, B, T, steps, lens), ref)
        assert L.ladiff_debug_set_pacing(9, 4) != 0
        # a small launch (44 blocks): by default its LIN / FFN workgroups rest after every block - the same bits as without
        Bs = 64
        lens_s = [196] * Bs
        assert L.ladiff_debug_set_stage_delay(0, 0) == 0
        small = run(nets, "pipeline16", precision, Bs, T, steps, lens_s)
        assert L.ladiff_debug_set_stage_delay(-1, 0) == 0
        assert torch.equal(run(nets, "pipeline16", precision, Bs, T, steps, lens_s), small)
    finally:
        L.ladiff_debug_set_stage_plan(0)
        L.ladiff_debug_set_poll_pause(0, 0)
        L.ladiff_debug_set_stage_delay(-1, 0)
        L.ladiff_debug_set_pacing(-1, 0)              # the built-in choice by launch size

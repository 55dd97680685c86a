"""The per-step noise generator of the stochastic schedulers (csrc/noise_gen.h, oracle/ladiff_oracle.py:device_noise): the integer
part against the Random123 known-answer vectors on the CPU, the device values against the numpy restatement on the GPU, and the loop
drawing its own noise against the same loop reading the generator's values from a tensor."""
import numpy as np
import pytest
import torch

from oracle import ladiff_oracle as orc

DEV = "cuda:0"

# Random123 (Salmon, Moraes, Dror, Shaw, SC'11), kat_vectors: philox4x32 10 <counter x4> <key x2> <expected x4>
PHILOX_KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


@pytest.mark.parametrize("counter,key,expected", PHILOX_KAT)
def test_philox_known_answers(counter, key, expected):
    out = orc.philox4x32_10(np.array(counter, dtype=np.uint64), np.array(key, dtype=np.uint64))
    assert tuple(int(v) for v in out) == expected


def test_oracle_noise_is_keyed_by_global_prompt_and_step():
    """A shard / chunk / window draws what the whole batch draws; different seeds, steps and prompts are different streams; moments."""
    whole = orc.device_noise(77, 0, 0, 6, 7, 5)
    assert np.array_equal(orc.device_noise(77, 3, 2, 3, 2, 5), whole[2:5, 3:5])
    assert not np.array_equal(orc.device_noise(78, 0, 0, 1, 7, 5), whole[:1])
    flat = whole.reshape(-1, 256)
    assert len({r.tobytes() for r in flat}) == flat.shape[0]
    big = orc.device_noise(1, 0, 0, 40, 16, 5)
    assert abs(float(big.mean())) < 5e-3 and abs(float(big.std()) - 1.0) < 5e-3 and np.isfinite(big).all()
    assert abs(float((big ** 4).mean()) - 3.0) < 0.05                                     # a normal's fourth moment


@pytest.mark.gpu
def test_device_noise_matches_the_numpy_restatement():
    from ladiff_amd import LADIFF
    for seed, first_prompt, first_step, n, B, T in [(0, 0, 0, 3, 4, 5), (0x1234567890ABCDEF, 1000, 950, 50, 9, 5), (5, 7, 11, 2, 3, 3)]:
        got = LADIFF.noise_tensor(seed, n, B, T, first_prompt=first_prompt, first_step=first_step, device=DEV).cpu().numpy()
        ref = orc.device_noise(seed, first_prompt, first_step, n, B, T)
        err = float(np.abs(got - ref).max())
        print(f"device noise vs numpy: seed {seed:#x} max abs diff {err:.2e}")
        assert err < 4e-6                    # identical integers; ln / cos / sin differ by the libraries' last bits (|z| < 6)


# ---------------------------------------------------------------- the loop drawing its own noise
def _pipe(nets, sched, steps, loop, eta=0.0, precision="f16x3", **kw):
    from ladiff_amd import LADIFF, DDIMScheduler, DDPMScheduler
    from test_gpu_pipeline import SCHED_KW
    den, vae = nets
    s = (DDIMScheduler(set_alpha_to_one=False, steps_offset=1, **SCHED_KW) if sched == "ddim"
         else DDPMScheduler(variance_type="fixed_small", **SCHED_KW))
    return LADIFF(denoiser=den, vae=vae, scheduler=s, guidance_scale=7.5, num_inference_timesteps=steps, eta=eta, max_it=5,
                  precision=precision, loop=loop, **kw)


@pytest.fixture(scope="module")
def nets():
    from ladiff_amd import LADiffDenoiser, LADiffVae, synthetic as syn
    from test_abi import ABL, DEN_KW, VAE_KW
    den = LADiffDenoiser(ABL, **DEN_KW); den.load_state_dict(syn.denoiser_weights(), strict=True)
    vae = LADiffVae(ABL, **VAE_KW); vae.load_state_dict(syn.vae_weights(263), strict=True)
    return den.to(DEV).eval(), vae.to(DEV).eval()


@pytest.mark.gpu
@pytest.mark.parametrize("loop", ["pipeline", "launches"])
@pytest.mark.parametrize("sched,steps,eta", [("ddpm", 120, 0.0), ("ddim", 9, 0.7)])
def test_loop_drawing_its_noise_equals_the_loop_reading_it(nets, loop, sched, steps, eta):
    """noise_seed = S (values computed inside the TAIL stage / the tail launch) against step_noise = the generator's tensor for S: the
    same device function on both sides, so the SAME BITS - over a windowed schedule (120 steps = three launches), in a chunked batch and
    on a shard that starts at global prompt 4."""
    from ladiff_amd import LADIFF, synthetic as syn
    lens = [196, 60, 120, 100, 48, 150, 196, 30, 77]
    B, T, S = len(lens), 5, 0xC0FFEE1234
    text, init = syn.text_embeddings(B, seed=11).to(DEV), syn.init_noise(lens, seed=12).to(DEV)
    pipe = _pipe(nets, sched, steps, loop, eta=eta)
    n = pipe._get_plan(B, T, steps, eta, torch.device(DEV), 1)["n"]
    tensor = LADIFF.noise_tensor(S, n, B, T, device=DEV)
    z_gen = pipe._diffusion_reverse(text, lens, init_noise=init, noise_seed=S)
    assert pipe.last_noise_seed == S and pipe.last_loop()[0] == (loop == "pipeline")
    z_ten = pipe._diffusion_reverse(text, lens, init_noise=init, step_noise=tensor)
    pipe.check()
    assert torch.isfinite(z_gen).all() and torch.equal(z_gen, z_ten)
    assert not torch.equal(z_gen, pipe._diffusion_reverse(text, lens, init_noise=init, noise_seed=S + 1))      # the seed matters
    assert torch.equal(z_gen, pipe._diffusion_reverse(text, lens, init_noise=init, noise_seed=S))              # and only the seed
    if loop == "pipeline":
        # chunks of <= 4 prompts: every chunk draws for its GLOBAL prompt indices
        chunked = _pipe(nets, sched, steps, loop, eta=eta, max_prompts_per_launch=4)
        z_ch = chunked._diffusion_reverse(text, lens, init_noise=init, noise_seed=S)
        chunked.check()
        assert (z_ch - z_gen).abs().max().item() < 1e-5 * max(1.0, z_gen.abs().max().item())      # other block packing: summation order of nothing, but be lenient
        # a rank that owns prompts 4 .. 8 of the batch
        shard = _pipe(nets, sched, steps, loop, eta=eta)
        shard.noise_first_prompt = 4
        tg = torch.cat([text[4:B], text[B + 4:]])
        z_sh = shard._diffusion_reverse(tg, lens[4:], init_noise=init[4:], noise_seed=S)
        shard.check()
        assert (z_sh - z_gen[:, 4:]).abs().max().item() < 1e-5 * max(1.0, z_gen.abs().max().item())


@pytest.mark.gpu
def test_default_noise_comes_from_torch_s_seed(nets):
    """No step_noise, no noise_seed: the seed is taken from torch's CPU generator - torch.manual_seed makes a DDPM run reproducible, and
    no [n,B,T,256] tensor is allocated for it."""
    from ladiff_amd import synthetic as syn
    lens = [196, 60, 120]
    text, init = syn.text_embeddings(3, seed=11).to(DEV), syn.init_noise(lens, seed=12).to(DEV)
    pipe = _pipe(nets, "ddpm", 100, "pipeline")
    torch.manual_seed(5)
    a = pipe._diffusion_reverse(text, lens, init_noise=init)
    seed_a = pipe.last_noise_seed
    b = pipe._diffusion_reverse(text, lens, init_noise=init)
    torch.manual_seed(5)
    c = pipe._diffusion_reverse(text, lens, init_noise=init)
    pipe.check()
    assert seed_a is not None and pipe.last_noise_seed == seed_a
    assert torch.equal(a, c) and not torch.equal(a, b)
    assert all(p["step_noise"] is None for p in pipe._plans.values())


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("fp32", 2e-4), ("f16x3", 1e-3)])
def test_ddpm_with_device_noise_matches_the_oracle(nets, precision, tol):
    """200 DDPM steps with the noise drawn on the device against the CPU oracle consuming the NUMPY generator's tensor."""
    from ladiff_amd import synthetic as syn
    lens = [196, 60, 130]
    B, T, S = 3, 5, 424242
    text, init = syn.text_embeddings(B, seed=43), syn.init_noise(lens, seed=44)
    pipe = _pipe(nets, "ddpm", 200, "pipeline", precision=precision)
    z, feats = pipe.sample(text.to(DEV), lens, init_noise=init.to(DEV), noise_seed=S)
    sn = torch.from_numpy(orc.device_noise(S, 0, 0, 200, B, T))
    _, ref = orc.sample_motions(syn.denoiser_weights(), syn.vae_weights(263), text, lens, init, n_steps=200, scheduler="ddpm", step_noise=sn)
    err = (feats.cpu() - ref).abs().max().item()
    print(f"DDPM-200 with device noise vs oracle ({precision}): max |frames - oracle| = {err:.3e}")
    assert err < tol * max(1.0, ref.abs().max().item())

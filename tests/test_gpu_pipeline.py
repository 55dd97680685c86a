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


def run(nets, loop, precision, B, T, steps, lens, sched="ddim", step_noise=None):
    den, vae = nets
    s = (DDIMScheduler(set_alpha_to_one=False, steps_offset=1, **SCHED_KW) if sched == "ddim"
         else DDPMScheduler(variance_type="fixed_small", **SCHED_KW))
    pipe = LADIFF(denoiser=den, vae=vae, scheduler=s, guidance_scale=7.5, num_inference_timesteps=steps, eta=0.0, max_it=T,
                  precision=precision, loop=loop)
    text = syn.text_embeddings(B, seed=900 + B).to(DEV)
    noise = torch.randn(B, T, 256, generator=torch.Generator().manual_seed(B * 10 + T)).to(DEV)
    z = pipe._diffusion_reverse(text, lens, init_noise=noise, step_noise=step_noise)
    code, info = pipe.loop_status()
    assert (code, info) == (0, 0), f"pipeline kernel aborted: code {code}, workgroup {info}"
    return z


@pytest.mark.parametrize("precision,tol", [("bf16x3", 2e-4), ("fp32", 1e-5)])
@pytest.mark.parametrize("B,T", [(1, 5), (2, 5), (3, 5), (4, 5), (7, 5), (43, 5), (5, 1), (9, 2), (6, 3), (5, 8)])
def test_pipeline_matches_launches(nets, precision, tol, B, T):
    """Blocks of P = 32 / (2 T) prompts: one block, several, a partial last block; every latent count the tiles allow."""
    lens = [max(1, min(196, 48 * ((i % T) + 1) - 5 * (i % 3))) for i in range(B)]          # latent counts 1..T, mixed
    za = run(nets, "launches", precision, B, T, 6, lens)
    zb = run(nets, "pipeline", precision, B, T, 6, lens)
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
    z32 = run(nets, "pipeline", "bf16x3", 11, 5, 8, lens)
    z16 = run(nets, "pipeline16", "bf16x3", 11, 5, 8, lens)
    assert torch.equal(z32, z16)
    assert torch.equal(z32, run(nets, "pipeline", "bf16x3", 11, 5, 8, lens))


def test_pipeline_ddpm_windows(nets):
    """A 200-step DDPM schedule runs as four 50-step windows (the c table is rebuilt per window, the latents carry over),
    with the per-step noise stream: pipeline == launches."""
    B, T, n = 5, 5, 200
    lens = [196, 60, 120, 100, 48]
    sn = syn.ddpm_noise(n, B, seed=5).to(DEV)
    za = run(nets, "launches", "bf16x3", B, T, n, lens, sched="ddpm", step_noise=sn)
    zb = run(nets, "pipeline", "bf16x3", B, T, n, lens, sched="ddpm", step_noise=sn)
    assert (za - zb).abs().max().item() < 5e-4 * max(1.0, za.abs().max().item())

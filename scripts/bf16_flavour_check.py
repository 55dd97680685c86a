"""The bf16-pair flavour of the library (ladiff_amd/libladiff_hip_bf16.so, _lib.select_split_format("bf16")) through the Python loop owner:
a 5-step guided sample of two prompts in split mode against the CPU oracle (test infrastructure: tests/test_gpu_path.py runs this in a
child process, because the format is chosen once per process)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ladiff_amd import _lib
_lib.select_split_format("bf16")
from ladiff_amd import LADIFF, DDIMScheduler, LADiffDenoiser, LADiffVae, synthetic as syn
from ladiff_amd.schema import ABL, DEN_KW, VAE_KW
from oracle import ladiff_oracle as orc
dev = "cuda:0"
den = LADiffDenoiser(ABL, **DEN_KW); den.load_state_dict(syn.denoiser_weights())
vae = LADiffVae(ABL, **VAE_KW); vae.load_state_dict(syn.vae_weights(263))
sch = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                    set_alpha_to_one=False, steps_offset=1)
pipe = LADIFF(denoiser=den.to(dev), vae=vae.to(dev), scheduler=sch, guidance_scale=7.5, num_inference_timesteps=5, precision="bf16x3")
lens = [60, 196]
text, noise = syn.text_embeddings(2), syn.init_noise(lens)
_, feats = pipe.sample(text.to(dev), lens, init_noise=noise.to(dev))
torch.cuda.synchronize()
_, f_o = orc.sample_motions(syn.denoiser_weights(), syn.vae_weights(263), text, lens, noise, 5, "ddim")
err = (feats.cpu() - f_o).abs().max().item()
print(f"split format {_lib.lib().ladiff_split_format()} ({_lib.split_mode_name()}): max |hip - oracle| = {err:.3e}")
assert _lib.lib().ladiff_split_format() == 0 and err < 1e-3

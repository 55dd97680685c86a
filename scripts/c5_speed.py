"""c5's per-rank workload (128 prompts of mixed lengths {60, 120, 196}, KIT-ML 251-dim decoder, 50-step DDIM) against the
same batch padded to 196 frames: whole sampling passes (loop + decode) timed with HIP events, outputs compared on the valid
frames.  Run on the GPU box: python scripts/c5_speed.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from ladiff_amd import synthetic as syn
from ladiff_amd.modules import LADiffDenoiser, LADiffVae
from ladiff_amd.pipeline import LADIFF
from ladiff_amd.schedulers import DDIMScheduler
from test_abi import ABL, DEN_KW, VAE_KW

dev = torch.device("cuda", 0)
NF = 251
den = LADiffDenoiser(ABL, **DEN_KW); den.load_state_dict(syn.denoiser_weights())
vae = LADiffVae(ABL, **dict(VAE_KW, nfeats=NF)); vae.load_state_dict(syn.vae_weights(NF))
sch = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                    clip_sample=False, set_alpha_to_one=False, steps_offset=1)
pipe = LADIFF(denoiser=den.to(dev).eval(), vae=vae.to(dev).eval(), scheduler=sch, guidance_scale=7.5,
              num_inference_timesteps=50, eta=0.0)
pipe.precision = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
B = 128
mixed = ([60, 120, 196] * 43)[:B]
text = syn.text_embeddings(B).to(dev)
noise = syn.init_noise([196] * B).to(dev)
stream = torch.cuda.Stream(device=dev)


def run(lens, loop, length_aware, reps=6):
    pipe.loop = loop
    pipe.vae.length_aware = length_aware
    ts, tl = [], []
    with torch.cuda.stream(stream), torch.no_grad():
        for i in range(reps + 2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            _, feats = pipe.sample(text, lens, init_noise=noise)
            e1.record(stream)
            torch.cuda.synchronize(dev)
            if i >= 2:
                ts.append(e0.elapsed_time(e1)); tl.append(pipe.loop_ms())
    ts.sort(); tl.sort()
    return feats, ts[len(ts) // 2], tl[len(tl) // 2]


f_pad, t_pad, l_pad = run([196] * B, "pipeline", False)
f_old, t_old, l_old = run(mixed, "pipeline32", False)
f_new, t_new, l_new = run(mixed, "pipeline", True)
f_ref, t_ref, l_ref = run(mixed, "launches", False)
print(f"{pipe.precision}: padded to 196: {t_pad:.2f} ms/pass (loop {l_pad:.2f}); mixed, masked only (32-row blocks, padded decode): "
      f"{t_old:.2f} (loop {l_old:.2f}); mixed, length-aware: {t_new:.2f} (loop {l_new:.2f}); launch-per-stage: {t_ref:.2f}")
print(f"speed-up over padding: {t_pad / t_new:.3f}x; over masked-only: {t_old / t_new:.3f}x")
print("frame rows computed:", sum(mixed), "of", B * 196)
d = (f_new - f_ref).abs().max().item()
tail = max(f_new[i, mixed[i]:].abs().max().item() if mixed[i] < 196 else 0.0 for i in range(B))
print(f"max |length-aware - launch-per-stage| = {d:.3e}; max |padded frames| = {tail:.1e}")
# the valid frames of a 196-frame prompt do not depend on the other prompts of the batch
same = [i for i in range(B) if mixed[i] == 196]
print(f"196-frame prompts vs the padded batch: {(f_new[same] - f_pad[same]).abs().max().item():.3e}")
# decode alone per bucket geometry
z = torch.randn(5, B, 256, device=dev)
for nb, F in ((128, 196), (86, 120), (43, 120), (43, 60), (42, 196), (16, 196), (8, 60)):
    with torch.cuda.stream(stream), torch.no_grad():
        pipe.vae.length_aware = False
        for i in range(3):
            pipe.vae.decode(z[:, :nb].contiguous(), [F] * nb)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(5):
            pipe.vae.decode(z[:, :nb].contiguous(), [F] * nb)
        e1.record(stream)
        torch.cuda.synchronize(dev)
    print(f"decode B={nb} F={F}: {e0.elapsed_time(e1) / 5:.3f} ms ({nb * F} rows)")
# the mixed batch's decode alone: single padded pass vs buckets
for la in (False, True):
    pipe.vae.length_aware = la
    with torch.cuda.stream(stream), torch.no_grad():
        for i in range(3):
            pipe.vae.decode(z, mixed)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(5):
            pipe.vae.decode(z, mixed)
        e1.record(stream)
        torch.cuda.synchronize(dev)
    print(f"decode of the mixed batch, length_aware={la}: {e0.elapsed_time(e1) / 5:.3f} ms")
for lens, name in (([196] * B, "padded"), (mixed, "mixed")):
    with torch.cuda.stream(stream), torch.no_grad():
        pipe.loop = "pipeline"
        for i in range(3):
            pipe._diffusion_reverse(text, lens, init_noise=noise)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(5):
            pipe._diffusion_reverse(text, lens, init_noise=noise)
        e1.record(stream)
        torch.cuda.synchronize(dev)
    print(f"reverse only, {name}: {e0.elapsed_time(e1) / 5:.3f} ms (loop {pipe.loop_ms():.3f})")

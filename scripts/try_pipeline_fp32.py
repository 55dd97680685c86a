"""fp32 mode: persistent pipeline loop against the launch-per-stage loop (both exact fp32 fma chains; only summation order differs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from ladiff_amd import synthetic as syn
dev = torch.device("cuda", 0)
pipe = bench.build_pipe(dev, 128)
pipe.precision = "fp32"
stream = torch.cuda.Stream(device=dev)
for B, steps in [(3, 5), (7, 5), (128, 50)]:
    lens = ([196, 60, 120, 100, 48, 150, 196] * 20)[:B]
    text = syn.text_embeddings(B).to(dev); noise = syn.init_noise(lens).to(dev)
    pipe.num_inference_timesteps = steps
    out = {}
    for loop in ("launches", "pipeline"):
        pipe.loop = loop
        with torch.cuda.stream(stream), torch.no_grad():
            z = pipe._diffusion_reverse(text, lens, init_noise=noise)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): z = pipe._diffusion_reverse(text, lens, init_noise=noise)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        out[loop] = z
        print(f"fp32 B={B} steps={steps} {loop}: {dt * 1e3:.3f} ms per loop, status {pipe.loop_status()}", flush=True)
    d = (out["pipeline"] - out["launches"]).abs().max().item()
    print(f"fp32 B={B} steps={steps}: max |z_pipeline - z_launches| = {d:.3e} (|z| max {out['launches'].abs().max().item():.1f})", flush=True)

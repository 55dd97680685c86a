"""Bring-up check of the persistent pipeline loop against the launch-per-stage loop (same bf16x3 arithmetic)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from ladiff_amd import synthetic as syn

dev = torch.device("cuda", 0)
pipe = bench.build_pipe(dev, 128)
pipe.precision = "fp32" if "fp32" in sys.argv[1:] else "bf16x3"
UNIFORM = "uniform" in sys.argv[1:]                      # every prompt 196 frames (the benchmark workload) instead of mixed lengths
FP32 = "fp32" in sys.argv[1:]                            # the strict-parity arithmetic mode instead of bf16x3
NOLOCAL = "nolocal" in sys.argv[1:]                      # every hand-off writes through (no XCD placement)
args = [a for a in sys.argv[1:] if a not in ("uniform", "fp32", "nolocal")]
if NOLOCAL:
    from ladiff_amd import _lib
    _lib.check(_lib.lib().ladiff_debug_set_xcd_local(0))
cases = [(3, 2), (3, 5), (7, 5), (128, 50)] if not args else [tuple(int(v) for v in a.split(",")) for a in args]
stream = torch.cuda.Stream(device=dev)
for B, steps in cases:
    lens = [196] * B if UNIFORM else ([196, 60, 120, 100, 48, 150, 196] * 20)[:B]
    text = syn.text_embeddings(B).to(dev)
    noise = syn.init_noise(lens).to(dev)
    pipe.num_inference_timesteps = steps
    out = {}
    for loop in ("launches", "pipeline32", "pipeline16"):
        pipe.loop = loop
        with torch.cuda.stream(stream), torch.no_grad():
            z = pipe._diffusion_reverse(text, lens, init_noise=noise)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                z = pipe._diffusion_reverse(text, lens, init_noise=noise)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
        out[loop] = z
        print(f"B={B} steps={steps} {loop}: {dt * 1e3:.3f} ms per call (loop kernel {pipe.loop_ms():.3f}), status {pipe.loop_status()}, finite {bool(torch.isfinite(z).all())}", flush=True)
    for k in ("pipeline32", "pipeline16"):
        d = (out[k] - out["launches"]).abs().max().item()
        print(f"B={B} steps={steps}: max |z_{k} - z_launches| = {d:.3e} (|z| max {out['launches'].abs().max().item():.1f})", flush=True)

"""Same-box A/B of two round-4 host-side changes: (1) `sample()` waiting for the loop's status words before it returns against
check=False (the host queues ahead), headline workload; (2) a DDPM loop drawing its noise in the TAIL stage against the same loop
reading the generator's values from a tensor (200 steps of config c3's shape).  python scripts/ab_check_noise.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from ladiff_amd import LADIFF

dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["headline"]
wl = bench.Workload(cfg, dev, 0, 1)
pipe = bench.build_pipe(dev, wl.B, cfg)
pipe.precision = "f16x3"
stream = torch.cuda.Stream(device=dev)
for chk in (True, False, True, False):
    with torch.cuda.stream(stream), torch.no_grad():
        for _ in range(4):
            pipe.sample(wl.text, wl.lens, init_noise=wl.noise, check=chk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            pipe.sample(wl.text, wl.lens, init_noise=wl.noise, check=chk)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
    pipe.check()
    print(f"headline: sample(check={chk!s:5}) {dt * 1e3:7.3f} ms per pass, {wl.B / dt:8.1f} motions/s", flush=True)

cfg = dict(bench.CONFIGS["c3"]); cfg["steps"] = 200
wl = bench.Workload(cfg, dev, 0, 1)
pipe = bench.build_pipe(dev, wl.B, cfg)
pipe.precision = "f16x3"
tensor = LADIFF.noise_tensor(7, 200, wl.B, 5, device=dev)
with torch.cuda.stream(stream), torch.no_grad():
    for mode in ("generator", "tensor", "generator", "tensor"):
        kw = dict(noise_seed=7) if mode == "generator" else dict(step_noise=tensor)
        ms = []
        for _ in range(4):
            z = pipe._diffusion_reverse(wl.text, wl.lens, init_noise=wl.noise, **kw)
            ms.append(pipe.loop_ms())
        print(f"ddpm-200 x 128 prompts: noise from the {mode:9}: loop {min(ms[1:]):8.3f} ms", flush=True)
    a = pipe._diffusion_reverse(wl.text, wl.lens, init_noise=wl.noise, noise_seed=7)
    b = pipe._diffusion_reverse(wl.text, wl.lens, init_noise=wl.noise, step_noise=tensor)
    print("same bits:", bool(torch.equal(a, b)))

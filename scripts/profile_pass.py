"""Profiling target: `python scripts/profile_pass.py <fp32|f16x3> [passes] [config]` runs 1 warm-up + N passes of a
benchmark workload (default: 128 prompts, 196 frames, 50-step DDIM + decode) in ONE precision mode and nothing else (no CPU
baseline, no second mode), so rocprofv3 --stats / --pmc summaries are per-mode.  It runs bench.py's own Workload."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
mode = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = bench.CONFIGS[sys.argv[3] if len(sys.argv) > 3 else "headline"]
dev = torch.device("cuda", 0)
wl = bench.Workload(cfg, dev, 0, 1)
pipe = bench.build_pipe(dev, wl.B, cfg)
pipe.precision = mode
s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s), torch.no_grad():
    for _ in range(1 + passes):
        feats = wl.one_pass(pipe)
    torch.cuda.synchronize()
if not cfg["decode_only"]:
    pipe.check()
print("profiled", mode, passes + 1, "passes", float(feats.abs().max()))

"""Profiling target: `python scripts/profile_pass.py <fp32|bf16x3> [passes]` runs 1 warm-up + N passes of the benchmark
workload (128 prompts, 196 frames, 50-step DDIM + decode) in ONE precision mode and nothing else (no CPU baseline, no
second mode), so rocprofv3 --stats / --pmc summaries are per-mode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ladiff_amd import synthetic as syn
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
pipe = bench.build_pipe(dev, bench.BATCH)
pipe.precision = mode
lens = [bench.FRAMES] * bench.BATCH
text = syn.text_embeddings(bench.BATCH).to(dev)
noise = syn.init_noise(lens).to(dev)
s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s), torch.no_grad():
    for _ in range(1 + passes):
        z, feats = pipe.sample(text, lens, init_noise=noise)
    torch.cuda.synchronize()
print("profiled", mode, passes + 1, "passes", float(feats.abs().max()))

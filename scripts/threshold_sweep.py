"""Loop kernel time of the pipeline over batch sizes around the block-count thresholds of its polling policy (csrc/systolic.hip:
LOOK_AHEAD_BLOCKS, SMALL_LAUNCH_BLOCKS), uniform 196-frame and mixed {60,120,196} batches, tagged hand-off against flags; calls queued
back to back.  Optional arguments "la=<blocks>" / "small=<blocks>" override the thresholds (ladiff_debug_set_loop_thresholds).
python scripts/threshold_sweep.py [la=72] [small=60]      env SIZES=8,16,... KINDS=u,m"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes
import torch
import bench
from ladiff_amd import _lib, synthetic as syn

dev = torch.device("cuda", 0)
L = _lib.lib()
la = small = -1
for a in sys.argv[1:]:
    if a.startswith("la="): la = int(a[3:])
    if a.startswith("small="): small = int(a[6:])
_lib.check(L.ladiff_debug_set_loop_thresholds(la, small))
sizes = [int(v) for v in os.environ.get("SIZES", "8,16,32,48,64,80,96,100,112,128,160,192,256,320").split(",")]
kinds = os.environ.get("KINDS", "u,m").split(",")
pipe = bench.build_pipe(dev, 128)
pipe.precision = "f16x3"; pipe.loop = "pipeline16"; pipe.num_inference_timesteps = 50
stream = torch.cuda.Stream(device=dev)
print(f"thresholds: look-ahead from {la if la >= 0 else 'built-in'} blocks, LIN / FFN rest up to {small if small >= 0 else 'built-in'} blocks")
print(f"{'prompts':>8s} {'kind':>5s} {'blocks':>6s} | {'flags ms':>9s} {'tags ms':>9s} {'tags motions/s (loop only)':>27s} {'us per block and step':>22s}")
for kind in kinds:
    for B in sizes:
        lens = [196] * B if kind == "u" else ([196, 60, 120] * 200)[:B]
        text, noise = syn.text_embeddings(B).to(dev), syn.init_noise(lens).to(dev)
        row = {}
        for ho in (0, 1):
            _lib.check(L.ladiff_debug_set_handoff(ho))
            with torch.cuda.stream(stream), torch.no_grad():
                for _ in range(3):
                    z = pipe._diffusion_reverse(text, lens, init_noise=noise)
                for _ in range(8):
                    z = pipe._diffusion_reverse(text, lens, init_noise=noise)
                torch.cuda.synchronize()
            assert pipe.loop_status()[0] == 0
            row[ho] = (pipe.loop_ms(), z.clone())
        nb = pipe.last_loop()[2]
        same = torch.equal(row[0][1], row[1][1])
        print(f"{B:8d} {kind:>5s} {nb:6d} | {row[0][0]:9.3f} {row[1][0]:9.3f} {B / row[1][0] * 1e3:27.0f} {row[1][0] * 1e3 / 50 / max(nb, 1):22.2f}"
              + ("" if same else "  BITS DIFFER"), flush=True)
_lib.check(L.ladiff_debug_set_handoff(1))
_lib.check(L.ladiff_debug_set_loop_thresholds(-1, -1))

"""Whole-pass A/B of the two hand-off protocols (loop + decode queued back to back, as bench.py runs them): pass time and the loop
kernel's own time inside the pass.  python scripts/pass_ab.py [config ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from ladiff_amd import _lib

dev = torch.device("cuda", 0)
L = _lib.lib()
for name in (sys.argv[1:] or ["headline", "c2", "c5"]):
    cfg = bench.CONFIGS[name]
    wl = bench.Workload(cfg, dev, 0, 1)
    pipe = bench.build_pipe(dev, wl.B, cfg)
    pipe.precision = "fp32" if os.environ.get("PRECISION") == "fp32" else "f16x3"
    stream = torch.cuda.Stream(device=dev)
    for ho in (0, 1, 0, 1):
        _lib.check(L.ladiff_debug_set_handoff(ho))
        with torch.cuda.stream(stream), torch.no_grad():
            for _ in range(4):
                wl.one_pass(pipe)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(10):
                wl.one_pass(pipe)
            e1.record(stream)
            torch.cuda.synchronize()
        print(f"{name}: handoff={'tags ' if ho else 'flags'} pass {e0.elapsed_time(e1) / 10:7.3f} ms, loop kernel in the last pass {pipe.loop_ms():7.3f} ms, "
              f"{wl.B * 10 / e0.elapsed_time(e1) * 1e3:8.1f} motions/s", flush=True)
_lib.check(L.ladiff_debug_set_handoff(1))

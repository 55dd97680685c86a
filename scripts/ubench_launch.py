"""Micro-benchmark: device time per dependent kernel in a captured graph vs eager (diagnostic, GPU box only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ladiff_amd import _lib
L = _lib.lib()
dev = "cuda:0"
step = torch.zeros(1, dtype=torch.int32, device=dev)
x = torch.randn(1280, 256, device=dev); y = torch.empty_like(x); g = torch.ones(256, device=dev); b = torch.zeros(256, device=dev)
s = torch.cuda.Stream()
def chain(n, kind):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(n):
        if kind == "advance":
            L.ladiff_advance_step(step.data_ptr(), st)
        else:
            L.ladiff_layernorm(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), 1280, st)
for kind in ("advance", "layernorm"):
    with torch.cuda.stream(s):
        chain(10, kind); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            chain(100, kind)
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20): gr.replay()
        e1.record(s); torch.cuda.synchronize()
        print(f"{kind}: graph  {e0.elapsed_time(e1) * 1e3 / 2000:.2f} us/kernel")
        e0.record(s); t0 = time.perf_counter()
        chain(2000, kind)
        e1.record(s); torch.cuda.synchronize()
        print(f"{kind}: eager  {e0.elapsed_time(e1) * 1e3 / 2000:.2f} us/kernel (host {1e6*(time.perf_counter()-t0)/2000:.2f})")

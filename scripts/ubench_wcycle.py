"""Diagnostic (GPU box): does the denoiser GEMM care whether its weights are L2-warm?  50 launches per graph of the
N=1024, K=256 bf16x3 GEMM with (a) the same weight matrix every launch, (b) 72 different matrices in turn (72 MiB, the
size of one denoiser's S-format weights: they come from the Infinity Cache), (c) as (b) with the activations rotating too."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ladiff_amd import _lib
L = _lib.lib(); dev = "cuda:0"
s = torch.cuda.Stream()
M, N, K = 1280, 1024, 256
Ws = [torch.randn(N, K, device=dev) * 0.05 for _ in range(72)]
As = [torch.randn(M, K, device=dev) for _ in range(8)]
b = torch.randn(N, device=dev); Ys = [torch.empty(M, N, device=dev) for _ in range(8)]
def run(wsel, asel, reps=72):
    def one(i):
        W, A, Y = Ws[wsel(i)], As[asel(i)], Ys[asel(i)]
        _lib.check(L.ladiff_gemm_resident(A.data_ptr(), K, None, 0, K, W.data_ptr(), K, b.data_ptr(), None, 0, None, N, M, N, K, 2, 1,
                                          Y.data_ptr(), s.cuda_stream))
    with torch.cuda.stream(s):
        one(0); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(reps): one(i)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(10): g.replay()
        e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (10 * reps)
print(f"same W, same A         : {run(lambda i: 0, lambda i: 0):.2f} us / launch")
print(f"72 W in turn, same A   : {run(lambda i: i % 72, lambda i: 0):.2f} us / launch")
print(f"72 W in turn, 8 A      : {run(lambda i: i % 72, lambda i: i % 8):.2f} us / launch")

"""Diagnostic (GPU box): cadence of the bf16x3 denoiser GEMM shapes (gemm_kp_kernel), 50 launches back to back inside a
graph, result checked against fp64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ladiff_amd import _lib
L = _lib.lib(); dev = "cuda:0"
s = torch.cuda.Stream()
def timeit(fn, reps=50):
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(10): g.replay()
        e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (10 * reps)
M = 1280
torch.manual_seed(0)
out = []
for (N, K) in [(1024, 256), (256, 1024), (768, 256), (256, 512), (256, 256)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    As, Ws = torch.empty_like(A), torch.empty_like(W)
    st = s.cuda_stream
    with torch.cuda.stream(s):
        _lib.check(L.ladiff_split_rows(A.data_ptr(), As.data_ptr(), M, K, st)); _lib.check(L.ladiff_split_rows(W.data_ptr(), Ws.data_ptr(), N, K, st))
    splits = K // 256
    Y = torch.zeros(splits, M, N, device=dev); res = torch.randn(M, N, device=dev)
    fn = lambda: _lib.check(L.ladiff_gemm_resident(As.data_ptr(), K, None, 0, K, Ws.data_ptr(), K, b.data_ptr() if splits == 1 else None,
                                                   res.data_ptr() if splits == 1 else None, N, Y.data_ptr(), N, M, N, K, 0, 1, None, st))
    t = timeit(fn)
    ref = A.double() @ W.double().t()
    got = Y.double().sum(0)
    if splits == 1: ref = ref + b.double() + res.double()
    err = (got - ref).abs().max().item()
    out.append(f"N={N:5d} K={K:5d}: {t:6.2f} us  err {err:.2e}")
print(" | ".join(out))

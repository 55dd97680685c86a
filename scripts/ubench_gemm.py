"""Diagnostic (GPU box): steady-state time of the denoiser GEMM shapes, old staged kernel vs K-resident kernel,
each launched 50x back to back inside a captured graph (so L2 is as warm as it gets)."""
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
for (N, K, ln) in [(768, 256, 0), (1024, 256, 0), (256, 256, 0), (256, 1024, 0), (256, 512, 0)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    Y = torch.empty(4, M, N, device=dev); res = torch.randn(M, N, device=dev)
    g_ = torch.ones(256, device=dev); be = torch.zeros(256, device=dev); xo = torch.empty(M, 256, device=dev)
    st = s.cuda_stream
    old = lambda: L.ladiff_gemm(A.data_ptr(), K, None, 0, K, W.data_ptr(), K, b.data_ptr(), res.data_ptr(), N, None, None, Y.data_ptr(), N, M, N, K, 0, 0, None, st)
    new = lambda: L.ladiff_gemm_resident(A.data_ptr(), K, None, 0, K, W.data_ptr(), K, b.data_ptr(), res.data_ptr(), N, Y.data_ptr(), N, M, N, K, 0, 0, None, st)
    t_old = timeit(old) if not ln else float("nan")
    t_new = timeit(new)
    spl = lambda: L.ladiff_gemm_resident(A.data_ptr(), K, None, 0, K, W.data_ptr(), K, b.data_ptr(), res.data_ptr(), N, Y.data_ptr(), N, M, N, K, 0, 1, None, st)
    t_spl = timeit(spl)
    fl = 2.0 * M * N * K
    print(f"N={N:5d} K={K:5d} ln={ln}: staged {t_old:7.2f} us   resident {t_new:7.2f} us  bf16x3 {t_spl:7.2f} us  ({fl / t_new / 1e6:6.1f} TF/s; ideal {fl / 157.3e6:.2f} us)")
P = torch.randn(4, M, 256, device=dev); out = torch.empty(M, 256, device=dev); tab = torch.randn(257, 256, device=dev)
cnt = torch.full((128,), 5, dtype=torch.int32, device=dev)
comb = lambda: L.ladiff_combine_rows(P.data_ptr(), 4, M, b.data_ptr(), res.data_ptr(), 2, g_.data_ptr(), be.data_ptr(), tab.data_ptr(), cnt.data_ptr(), 128, 5, 256, out.data_ptr(), s.cuda_stream)
print(f"combine_rows(4 planes, LN+add): {timeit(comb):.2f} us")

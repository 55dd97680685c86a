"""Diagnostic (GPU box): large-M f16x3 GEMM shapes of the decoder / CLIP (M = 25088 by default, env M overrides) through
ladiff_gemm_split; checks against fp64 on a row sample."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ladiff_amd import _lib
L = _lib.lib(); dev = "cuda:0"
s = torch.cuda.Stream()
def timeit(fn, reps=20):
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): fn()
        e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
M = int(os.environ.get("M", 25088))
torch.manual_seed(0)
out = []
for (N, K, act) in [(768, 256, 0), (256, 256, 0), (1024, 256, 2), (256, 1024, 0), (256, 512, 0), (3072, 768, 4), (768, 3072, 0)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    As, Ws = torch.empty_like(A), torch.empty_like(W)
    st = s.cuda_stream
    with torch.cuda.stream(s):
        _lib.check(L.ladiff_split_rows(A.data_ptr(), As.data_ptr(), M, K, st)); _lib.check(L.ladiff_split_rows(W.data_ptr(), Ws.data_ptr(), N, K, st))
    Y = torch.zeros(M, N, device=dev); Ys = torch.zeros(M, N, device=dev)
    fn = lambda: _lib.check(L.ladiff_gemm_split(As.data_ptr(), K, None, 0, K, Ws.data_ptr(), K, b.data_ptr(), None, 0, Y.data_ptr(), Ys.data_ptr(), N, M, N, K, act, st))
    t = timeit(fn)
    idx = torch.randint(0, M, (256,), device=dev); idx[0] = M - 1
    ref = A[idx].double() @ W.double().t() + b.double()
    if act == 2: ref = torch.nn.functional.gelu(ref)
    if act == 4: ref = ref * torch.sigmoid(1.702 * ref)
    err = (Y[idx].double() - ref).abs().max().item()
    Yb = torch.empty_like(Y)
    # S-format twin decodes to the same values (16 significant bits)
    out.append(f"N={N:4d} K={K:4d}: {t:7.2f} us {2.0 * M * N * K / t / 1e6:6.1f} TF/s err {err:.1e}")
print(f"M={M}  " + " | ".join(out))

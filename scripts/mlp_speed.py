"""The fused decoder feed-forward kernel alone (ladiff_mlp_ln_fused) at several row counts and measurement variants, against the
three-launch form (linear1 GEMM + linear2 GEMM + LayerNorm): HIP-event us per call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ladiff_amd import _lib
if os.environ.get("LADIFF_LIB"):                      # an experiment build of the library (same ABI)
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["LADIFF_LIB"])
L = _lib.lib()
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
def r(*s, sc=1.0): return (sc * torch.randn(*s, generator=g)).to(dev)
def split(t):
    s = torch.empty_like(t); _lib.check(L.ladiff_split_rows(_lib.ptr(t), _lib.ptr(s), t.shape[0], t.shape[1], _lib.stream_ptr())); return s
w1, b1, w2, b2 = r(1024, 256, sc=1 / 16), r(1024), r(256, 1024, sc=1 / 32), r(256)
g3, be3 = 1 + 0.1 * r(256), 0.1 * r(256)
w1s, w2s = split(w1), split(w2)
st = torch.cuda.Stream()
def timeit(fn, n=30):
    with torch.cuda.stream(st):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(n): fn()
        e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (25088, 480):
    x = r(M, 256, sc=2.0); xs = split(x)
    y, ys, hid = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev), torch.empty(M, 1024, device=dev)
    sp = st.cuda_stream
    def fused():
        _lib.check(L.ladiff_mlp_ln_fused(_lib.ptr(xs), _lib.ptr(x), _lib.ptr(w1s), _lib.ptr(b1), _lib.ptr(w2s), _lib.ptr(b2), _lib.ptr(g3), _lib.ptr(be3),
                                         None, None, _lib.ptr(y), _lib.ptr(ys), M, sp))
    def three():
        _lib.check(L.ladiff_gemm_split(_lib.ptr(xs), 256, None, 0, 256, _lib.ptr(w1s), 256, _lib.ptr(b1), None, 0, None, _lib.ptr(hid), 1024, M, 1024, 256, 2, sp))
        _lib.check(L.ladiff_gemm_split(_lib.ptr(hid), 1024, None, 0, 1024, _lib.ptr(w2s), 1024, _lib.ptr(b2), _lib.ptr(x), 256, _lib.ptr(y), None, 256, M, 256, 1024, 0, sp))
        _lib.check(L.ladiff_layernorm(_lib.ptr(y), _lib.ptr(g3), _lib.ptr(be3), _lib.ptr(ys), M, sp))
    out = [f"M={M:6d}: three launches {timeit(three):7.1f} us"]
    for v in (1, 2, 3):
        L.ladiff_debug_set_mlp_variant(v)
        out.append(f"v{v} {timeit(fused):7.1f}")
        print(out[-1], file=sys.stderr, flush=True)
    L.ladiff_debug_set_mlp_variant(0)
    print(" | ".join(out), flush=True)

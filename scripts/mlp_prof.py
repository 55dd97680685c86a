"""Profiling target: the fused decoder feed-forward kernel alone, `python scripts/mlp_prof.py <variant> [M]` (20 launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ladiff_amd import _lib
L = _lib.lib()
dev = "cuda:0"
v = int(sys.argv[1]) if len(sys.argv) > 1 else 0
M = int(sys.argv[2]) if len(sys.argv) > 2 else 25088
g = torch.Generator().manual_seed(0)
def r(*s, sc=1.0): return (sc * torch.randn(*s, generator=g)).to(dev)
def split(t):
    s = torch.empty_like(t); _lib.check(L.ladiff_split_rows(_lib.ptr(t), _lib.ptr(s), t.shape[0], t.shape[1], _lib.stream_ptr())); return s
w1, b1, w2, b2 = r(1024, 256, sc=1 / 16), r(1024), r(256, 1024, sc=1 / 32), r(256)
g3, be3 = 1 + 0.1 * r(256), 0.1 * r(256)
w1s, w2s = split(w1), split(w2)
x = r(M, 256, sc=2.0); xs = split(x)
y, ys = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev)
L.ladiff_debug_set_mlp_variant(v)
for _ in range(20):
    _lib.check(L.ladiff_mlp_ln_fused(_lib.ptr(xs), _lib.ptr(x), _lib.ptr(w1s), _lib.ptr(b1), _lib.ptr(w2s), _lib.ptr(b2), _lib.ptr(g3), _lib.ptr(be3),
                                     None, None, _lib.ptr(y), _lib.ptr(ys), M, _lib.stream_ptr()))
torch.cuda.synchronize()
print("done", v, M, float(y.abs().max()))

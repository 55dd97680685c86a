"""The fused feed-forward kernel (csrc/dec_mlp.hip: a ring of eight LDS-DMA weight stages; rounds 3 - 5 six in flight behind COUNTED vmcnt
waits, since round 6 one batch per wave behind vmcnt(0)) beside kernels that sweep the caches on another stream: large device-to-device copies (the weights then
miss the L2 again and again: their pieces' latencies spread) and the split GEMM.  Every launch's output against the first one's bits.
usage: mlp_under_memory_pressure.py [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ladiff_amd import _lib
if os.environ.get("LADIFF_LIB"):                      # an experiment build of the library (same ABI)
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["LADIFF_LIB"])
L = _lib.lib()
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
g = torch.Generator().manual_seed(0)
def r(*s, sc=1.0): return (sc * torch.randn(*s, generator=g)).to(dev)
def split(t):
    s = torch.empty_like(t); _lib.check(L.ladiff_split_rows(_lib.ptr(t), _lib.ptr(s), t.shape[0], t.shape[1], _lib.stream_ptr())); return s
w1, b1, w2, b2 = r(1024, 256, sc=1 / 16), r(1024), r(256, 1024, sc=1 / 32), r(256)
g3, be3 = 1 + 0.1 * r(256), 0.1 * r(256)
w1s, w2s = split(w1), split(w2)
M = 25088
x = r(M, 256, sc=2.0); xs = split(x)
ys = [torch.empty(M, 256, device=dev) for _ in range(8)]
big_a = torch.empty(96 << 20, device=dev); big_b = torch.empty(96 << 20, device=dev)        # 384 MB each: more than L2 + Infinity Cache
A, Bs = torch.cuda.Stream(), torch.cuda.Stream()
def mlp(y, sp):
    _lib.check(L.ladiff_mlp_ln_fused(_lib.ptr(xs), _lib.ptr(x), _lib.ptr(w1s), _lib.ptr(b1), _lib.ptr(w2s), _lib.ptr(b2), _lib.ptr(g3), _lib.ptr(be3), None, None, _lib.ptr(y), None, M, sp))
with torch.cuda.stream(A): mlp(ys[0], A.cuda_stream)
torch.cuda.synchronize(); ref = ys[0].clone()
for name in ("nothing", "384-MB copies", "split GEMMs (gemm_big_split, K = 256)"):
    bad = 0
    for it in range(n // 8):
        with torch.cuda.stream(Bs):
            if name.startswith("384"):
                for _ in range(3): big_b.copy_(big_a)
            elif name.startswith("split"):
                for _ in range(10):
                    _lib.check(L.ladiff_gemm_split(_lib.ptr(xs), 256, None, 0, 256, _lib.ptr(w1s), 256, _lib.ptr(b1), None, 0, None, _lib.ptr(big_a), 1024, M, 1024, 256, 0, Bs.cuda_stream))
        with torch.cuda.stream(A):
            for k in range(8): mlp(ys[k], A.cuda_stream)
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(y, ref) else 1 for y in ys)
    print(f"beside {name}: {bad} of {n // 8 * 8} launches differ from the first launch's bits", flush=True)

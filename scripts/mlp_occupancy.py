"""How does the fused feed-forward kernel's time depend on how many CUs it occupies?  The unbalanced <4, 2> form (one 128-row block per
workgroup, 8 hidden slices each) at 64 ... 256 row blocks: if the time per launch grows with the workgroup count the kernel is
bound by something the workgroups share (clock / power, the weight stream through the L2), and dealing 196 blocks' work to 256 CUs
cannot shorten it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ladiff_amd import _lib
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
L.ladiff_debug_set_mlp_variant(3)
for nb in (32, 64, 98, 128, 160, 196, 224, 256, 320, 392):
    M = nb * 128
    x = r(M, 256, sc=2.0); xs = split(x)
    y, ys = torch.empty(M, 256, device=dev), torch.empty(M, 256, device=dev)
    sp = st.cuda_stream
    with torch.cuda.stream(st):
        for _ in range(3):
            _lib.check(L.ladiff_mlp_ln_fused(_lib.ptr(xs), _lib.ptr(x), _lib.ptr(w1s), _lib.ptr(b1), _lib.ptr(w2s), _lib.ptr(b2), _lib.ptr(g3), _lib.ptr(be3), None, None, _lib.ptr(y), _lib.ptr(ys), M, sp))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(30):
            _lib.check(L.ladiff_mlp_ln_fused(_lib.ptr(xs), _lib.ptr(x), _lib.ptr(w1s), _lib.ptr(b1), _lib.ptr(w2s), _lib.ptr(b2), _lib.ptr(g3), _lib.ptr(be3), None, None, _lib.ptr(y), _lib.ptr(ys), M, sp))
        e1.record(st)
    torch.cuda.synchronize()
    print(f"{nb:4d} row blocks = workgroups of the unbalanced <4, 2> form: {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us per launch", flush=True)
L.ladiff_debug_set_mlp_variant(0)

"""Phase timeline of the decoder's fused out_proj + cross-attention kernel (diagnostic twin build: python -m ladiff_amd.build --stamps;
microseconds between the stamps of workgroup (3, 0), thread 0, per 32-row pass."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes
import torch
from ladiff_amd import LADiffVae, _lib, synthetic as syn
from ladiff_amd import build as _build
_lib.LIB_PATH = _build.stamps_lib(os.environ.get("LADIFF_STAMPS_LEVEL", "1"))
from test_abi import ABL, VAE_KW
dev = "cuda:0"
vae = LADiffVae(ABL, **VAE_KW); vae.load_state_dict(syn.vae_weights(263)); vae = vae.to(dev).eval()
vae.precision = "f16x3"
L = _lib.lib()
B, F = 128, 196
lens = [F] * B
z = torch.randn(5, B, 256, generator=torch.Generator().manual_seed(1)).to(dev)
st = torch.zeros(128, dtype=torch.int64, device=dev)
L.ladiff_debug_set_sys_stamps.argtypes = [ctypes.c_void_p]
with torch.no_grad():
    for _ in range(3):
        vae.decode(z, lens)
    L.ladiff_debug_set_sys_stamps(ctypes.c_void_p(st.data_ptr()))
    vae.decode(z, lens)
    torch.cuda.synchronize()
    L.ladiff_debug_set_sys_stamps(ctypes.c_void_p(0))
t = st.cpu().tolist()                      # the LAST layer's launch wrote last
names = ["loop top", "att tile whole", "product 1 done", "staged + barrier", "norm1 + tiles written", "barrier", "scores + barrier", "softmax + barrier", "product 3 staged", "stored"]
print("prologue (kernel start -> first loop top): %.2f us" % ((t[0] - t[15]) / 100.0))
for ps in range(6):
    r = t[ps * 16: ps * 16 + 10]
    if r[0] == 0: break
    print(f"pass {ps}: " + "  ".join(f"{names[i]} +{(r[i] - r[i - 1]) / 100.0:.2f}" for i in range(1, 10)) + f"   | whole pass {(r[9] - r[0]) / 100.0:.2f} us")

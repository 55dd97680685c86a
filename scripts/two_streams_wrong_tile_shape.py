"""Victim cut after 21 launches (layer 5's skip GEMM, gemm_big_split K = 512 -> P3 | Ps3), aggressor = a whole decode on another stream:
the shape of what differs in P3 (rows, columns, magnitudes) when it differs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ladiff_amd import LADiffVae, _lib, synthetic as syn
from test_abi import ABL, VAE_KW
if os.environ.get("LADIFF_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["LADIFF_LIB"])
dev = "cuda:0"
vae = LADiffVae(ABL, **VAE_KW); vae.load_state_dict(syn.vae_weights(263)); vae = vae.to(dev).eval()
vae.precision = "f16x3"
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
cut = int(sys.argv[2]) if len(sys.argv) > 2 else 21
g = torch.Generator().manual_seed(1)
B, F, T, C = 64, 196, 5, 263
z = torch.randn(T, B, 256, generator=g).to(dev); z2 = torch.randn(T, B, 256, generator=g).to(dev)
A, Bs = torch.cuda.Stream(), torch.cuda.Stream()
lens = [F] * B; counts = [5] * B
L = _lib.lib()
wsb = L.ladiff_decoder_workspace_bytes(B, F, T, C)
wt = vae._weight_table(); wsp = wt.split_array()
with torch.cuda.stream(A):
    featsA = torch.zeros(B, F, C, device=dev); wsA = torch.zeros((wsb + 3) // 4, device=dev)
    lA = _lib.device_ints(lens, dev); cA = _lib.device_ints(counts, dev)
with torch.cuda.stream(Bs):
    featsB = torch.zeros(B, F, C, device=dev); wsB = torch.zeros((wsb + 3) // 4, device=dev)
    lB = _lib.device_ints(lens, dev); cB = _lib.device_ints(counts, dev)
torch.cuda.synchronize()
def victim(c):
    os.environ["LADIFF_DEC_CUT"] = str(c)
    _lib.check(L.ladiff_vae_decode(wt.array, wsp, z.data_ptr(), lA.data_ptr(), cA.data_ptr(), B, F, T, C, featsA.data_ptr(), wsA.data_ptr(), wsb, A.cuda_stream))
def agg():
    os.environ["LADIFF_DEC_CUT"] = "100"
    _lib.check(L.ladiff_vae_decode(wt.array, wsp, z2.data_ptr(), lB.data_ptr(), cB.data_ptr(), B, F, T, C, featsB.data_ptr(), wsB.data_ptr(), wsb, Bs.cuda_stream))
M = B * F; MD = M * 256
victim(cut); torch.cuda.synchronize(); wref = wsA.clone()
P3 = lambda w: w[3 * MD:4 * MD].view(M, 256)
shown = 0; bad = 0
for it in range(runs):
    victim(cut); agg(); torch.cuda.synchronize()
    if not torch.equal(wsA, wref):
        bad += 1
        if shown < 12:
            shown += 1
            d = (P3(wsA) != P3(wref))
            rows = d.any(1).nonzero().flatten(); cols = d.any(0).nonzero().flatten()
            mag = (P3(wsA) - P3(wref)).abs()
            rl = rows.tolist()
            print(f"run {it}: {len(rl)} rows {rl[0]}..{rl[-1]} (tile {rl[0] // 128}, row in tile {rl[0] % 128}..{rl[-1] % 128}), cols {int(cols.min())}..{int(cols.max())} ({cols.numel()}), "
                  f"differing floats {int(d.sum())}, max |diff| {mag.max().item():.3e}, median |diff| over differing {mag[d].median().item():.3e}; rows%32 set {sorted(set(r % 32 for r in rl))[:40]}", flush=True)
print(f"cut {cut}: {bad} of {runs} runs differ")

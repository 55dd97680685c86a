"""Does a decode write outside the workspace / output it was given?  The C entry is called with a workspace and an output that sit
inside larger buffers filled with a canary word; afterwards the guard zones must be untouched.  usage: decode_guard.py [B] [F]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ladiff_amd import LADiffVae, _lib, synthetic as syn
from test_abi import ABL, VAE_KW
dev = "cuda:0"
vae = LADiffVae(ABL, **VAE_KW); vae.load_state_dict(syn.vae_weights(263)); vae = vae.to(dev).eval()
L = _lib.lib()
CAN = 0x7fc12345
G = 16 << 20          # guard words (64 MB) on each side
bad_total = 0
for B, F in ((64, 196), (128, 196), (8, 60), (37, 100)):
    lens = [F] * B
    T = 5
    z = torch.randn(T, B, 256, generator=torch.Generator().manual_seed(1)).to(dev)
    counts = [-(-l // 48) for l in lens]
    for prec in ("f16x3", "fp32"):
        vae.precision = prec
        wt = vae._weight_table(); wsplit = wt.split_array() if prec == "f16x3" else None
        wsb = L.ladiff_decoder_workspace_bytes(B, F, T, 263)
        nw = (wsb + 3) // 4
        big = torch.full((nw + 2 * G,), CAN, dtype=torch.int32, device=dev)
        nf = B * F * 263
        fbig = torch.full((nf + 2 * G,), CAN, dtype=torch.int32, device=dev)
        lens_t = _lib.device_ints(lens, dev); counts_t = _lib.device_ints(counts, dev)
        torch.cuda.synchronize()
        _lib.check(L.ladiff_vae_decode(wt.array, wsplit, z.data_ptr(), lens_t.data_ptr(), counts_t.data_ptr(), B, F, T, 263,
                                       fbig.data_ptr() + 4 * G, big.data_ptr() + 4 * G, wsb, _lib.stream_ptr()))
        torch.cuda.synchronize()
        msg = []
        for name, t, n in (("workspace", big, nw), ("feats", fbig, nf)):
            lo = (t[:G] != CAN).nonzero().flatten(); hi = (t[G + n:] != CAN).nonzero().flatten()
            msg.append(f"{name}: {lo.numel()} words written below (nearest at -{G - int(lo.max()) if lo.numel() else 0}), {hi.numel()} above (furthest at +{int(hi.max()) if hi.numel() else 0}, first +{int(hi.min()) if hi.numel() else 0})")
        print(f"B {B} F {F} {prec}: " + "; ".join(msg), flush=True)
        bad_total += sum(1 for m in msg if not m.split(": ")[1].startswith("0 words written below (nearest at -0), 0 above"))
print(f"guard zones touched: {bad_total}")
sys.exit(1 if bad_total else 0)

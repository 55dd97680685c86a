"""Does the decoder gain from running sample groups on concurrent streams (one group's feed-forward kernel - 196 of 256 CUs at the full
batch - beside another group's attention kernels)?  ms per decode of 128 x 196 frames: one stream against 2 / 3 / 4 sample groups on
streams of their own.  python scripts/decode_split.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ladiff_amd import LADiffVae, synthetic as syn
from test_abi import ABL, VAE_KW
dev = "cuda:0"
vae = LADiffVae(ABL, **VAE_KW); vae.load_state_dict(syn.vae_weights(263)); vae = vae.to(dev).eval()
vae.precision = "f16x3"
B, F = 128, 196
lens = [F] * B
z = torch.randn(5, B, 256, generator=torch.Generator().manual_seed(1)).to(dev)
main = torch.cuda.Stream()
side = [torch.cuda.Stream() for _ in range(4)]

def split_decode(n):
    if n == 1:
        return vae.decode(z, lens)
    outs, step = [], (B + n - 1) // n
    ev = torch.cuda.Event(); ev.record(main)
    for g in range(n):
        lo, hi = g * step, min(B, (g + 1) * step)
        with torch.cuda.stream(side[g]):
            side[g].wait_event(ev)
            outs.append(vae.decode(z[:, lo:hi].contiguous(), lens[lo:hi]))
    for g in range(n):
        main.wait_stream(side[g])
    return torch.cat(outs)

ref = None
with torch.cuda.stream(main), torch.no_grad():
    for n in (1, 2, 3, 4, 1, 2):
        for _ in range(3):
            out = split_decode(n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for _ in range(20):
            out = split_decode(n)
        e1.record(main)
        torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        print(f"{n} group(s): {e0.elapsed_time(e1) / 20:.3f} ms per decode, same bits as one group: {bool(torch.equal(out, ref))}", flush=True)

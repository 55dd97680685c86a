"""Bisect (diagnostic twin, LADIFF_LIB=ladiff_amd/libladiff_hip_stamps.so): victim = a whole decode 64 x 196 f16x3 launched first on
stream A; aggressor = the FIRST n launches of a second decode (LADIFF_DEC_CUT=n, csrc/decoder.hip) launched right after on stream B.
usage: decode_victim3.py [runs] [cut ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ladiff_amd import LADiffVae, _lib, synthetic as syn
from test_abi import ABL, VAE_KW
if os.environ.get("LADIFF_LIB"):                      # the diagnostic twin (same ABI)
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["LADIFF_LIB"])
dev = "cuda:0"
vae = LADiffVae(ABL, **VAE_KW); vae.load_state_dict(syn.vae_weights(263)); vae = vae.to(dev).eval()
vae.precision = "f16x3"
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 500
cuts = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4, 5, 6, 7, 8, 11, 20, 100]
g = torch.Generator().manual_seed(1)
z = torch.randn(5, 64, 256, generator=g).to(dev)
z2 = torch.randn(5, 64, 256, generator=g).to(dev)
A, Bs = torch.cuda.Stream(), torch.cuda.Stream()
lens = [196] * 64
with torch.no_grad():
    with torch.cuda.stream(A): ref = vae.decode(z, lens)
    with torch.cuda.stream(Bs): vae.decode(z2, lens)
    torch.cuda.synchronize()
    for cut in cuts:
        bad = 0; worst = 0.0; where = []
        for it in range(runs):
            os.environ.pop("LADIFF_DEC_CUT", None)
            with torch.cuda.stream(A): out = vae.decode(z, lens)
            os.environ["LADIFF_DEC_CUT"] = str(cut)
            with torch.cuda.stream(Bs): vae.decode(z2, lens)
            torch.cuda.synchronize()
            d = (out - ref).abs().amax(dim=(1, 2))
            if d.max().item() > 0: bad += 1; worst = max(worst, d.max().item()); where += torch.nonzero(d).flatten().tolist()
        print(f"aggressor = first {cut:3d} launches: victim differs in {bad} of {runs} runs (worst {worst:.3e}) samples {where[:14]}", flush=True)

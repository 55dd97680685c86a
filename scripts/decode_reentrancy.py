"""Check: two decodes of disjoint sample ranges running CONCURRENTLY on separate streams (separate workspaces) give the bits the same
two calls give one after another, over N runs; per decoder-fusion switch (ladiff_debug_set_decoder_fusion).  Exits non-zero when a
concurrent run differs.  This is the test that found the counted `vmcnt` waits behind LDS-DMA stages in flight (gemm_big.hip, header:
1 - 2 % of the concurrent decodes had a wrong GEMM tile; profiles/r5/11_*).
usage: decode_reentrancy.py [runs] [switch ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ladiff_amd import LADiffVae, _lib, synthetic as syn
if os.environ.get("LADIFF_LIB"):                      # an experiment build of the library (same ABI)
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["LADIFF_LIB"])
from test_abi import ABL, VAE_KW
dev = "cuda:0"
vae = LADiffVae(ABL, **VAE_KW); vae.load_state_dict(syn.vae_weights(263)); vae = vae.to(dev).eval()
vae.precision = "f16x3"
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
switches = [int(a) for a in sys.argv[2:]] or [1, 65, 17, 81, 2 + 64 + 16, 0, 64 + 16]
B, F = 128, 196
lens = [F] * B
z = torch.randn(5, B, 256, generator=torch.Generator().manual_seed(1)).to(dev)
parts = 2
cut = [B * i // parts for i in range(parts + 1)]
zs = [z[:, cut[i]:cut[i + 1]].contiguous() for i in range(parts)]
torch.cuda.synchronize()
side = [torch.cuda.Stream() for _ in range(parts)]
L = _lib.lib()
def both(concurrent):
    outs = []
    for i in range(parts):
        with torch.cuda.stream(side[i]):
            outs.append(vae.decode(zs[i], lens[cut[i]:cut[i + 1]]))
        if not concurrent:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return torch.cat(outs, 0)
total_bad = 0
with torch.no_grad():
    for sw in switches:
        L.ladiff_debug_set_decoder_fusion(sw)
        ref = both(False)
        for concurrent in (False, True):
            bad = 0; worst = 0.0; where = []
            for it in range(runs):
                d = (both(concurrent) - ref).abs().amax(dim=(1, 2))
                if d.max().item() > 0:
                    bad += 1; worst = max(worst, d.max().item()); where += torch.nonzero(d).flatten().tolist()
            total_bad += bad
            print(f"switch {sw:3d} concurrent {concurrent}: {bad} of {runs} runs differ, worst {worst:.3e}, samples {where[:16]}", flush=True)
L.ladiff_debug_set_decoder_fusion(1)
sys.exit(1 if total_bad else 0)

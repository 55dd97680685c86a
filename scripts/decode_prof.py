"""Ten f16x3 decodes at the benchmark size (128 x 196 frames) for `rocprofv3 --kernel-trace --stats -- python3 scripts/decode_prof.py [switch]`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ladiff_amd import LADiffVae, _lib, synthetic as syn
from test_abi import ABL, VAE_KW
if os.environ.get("LADIFF_LIB"):                      # an experiment build of the library (same ABI)
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["LADIFF_LIB"])
dev = "cuda:0"
vae = LADiffVae(ABL, **VAE_KW); vae.load_state_dict(syn.vae_weights(263)); vae = vae.to(dev).eval()
vae.precision = "f16x3"
_lib.lib().ladiff_debug_set_decoder_fusion(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
B, F = 128, 196
lens = [F] * B
z = torch.randn(5, B, 256, generator=torch.Generator().manual_seed(1)).to(dev)
for i, m in enumerate(syn.max_iter_elements(lens)):
    z[m:, i] = 0
with torch.no_grad():
    for _ in range(12):
        out = vae.decode(z, lens)
torch.cuda.synchronize()
print("ok", out.abs().max().item())

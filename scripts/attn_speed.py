"""Decode at the benchmark size with timing builds of the attention kernel (dec_qkv_attn.hip): ms per decode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ladiff_amd import LADiffVae, _lib, synthetic as syn
from test_abi import ABL, VAE_KW
dev = "cuda:0"
vae = LADiffVae(ABL, **VAE_KW); vae.load_state_dict(syn.vae_weights(263)); vae = vae.to(dev).eval()
vae.precision = "f16x3"
L = _lib.lib()
B, F = 128, 196
lens = [F] * B
z = torch.randn(5, B, 256, generator=torch.Generator().manual_seed(1)).to(dev)
for i, m in enumerate(syn.max_iter_elements(lens)):
    z[m:, i] = 0
for v in (0, 21, 22, 23, 24, 25, 26):
    L.ladiff_debug_set_mlp_variant(v)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(3):
            out = vae.decode(z, lens)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20):
            out = vae.decode(z, lens)
        e1.record(s)
        torch.cuda.synchronize()
    print(f"variant {v}: {e0.elapsed_time(e1) / 20:.3f} ms per decode (0 = real, 21 = no DMA waits, 22 = no attention core, 23 = no projection MFMAs, 24 = no projection, 25 = no x loads, 26 = x loads + one store only)", flush=True)
L.ladiff_debug_set_mlp_variant(0)

"""LA-VAE decode alone at the benchmark size (128 x 196 frames) and at config c1's (8 x 60), both arithmetic modes, with the
decoder's feed-forward block fused (csrc/dec_mlp.hip) and as three launches: HIP-event ms per decode."""
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
L = _lib.lib()
for B, F in ((128, 196), (8, 60)):
    lens = [F] * B
    z = torch.randn(5, B, 256, generator=torch.Generator().manual_seed(1)).to(dev)
    for i, m in enumerate(syn.max_iter_elements(lens)):
        z[m:, i] = 0
    for prec in ("f16x3", "fp32"):
        vae.precision = prec
        for fused in ((1, 65, 17, 9, 2, 0, 4) if prec == "f16x3" else (1,)):          # 65 = out_proj GEMM + cross-attention row kernel as two launches, 17 = in_proj + attention as two launches, 9 = final_layer on the fp32-input kernel
            L.ladiff_debug_set_decoder_fusion(fused)
            vae.graph_rows = 4096 if fused == 2 and B * F < 4096 else 0     # fused == 2 at a small size: also replayed from a hipGraph
            s = torch.cuda.Stream()
            with torch.cuda.stream(s), torch.no_grad():
                for _ in range(3):
                    out = vae.decode(z, lens)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 20
                e0.record(s)
                for _ in range(n):
                    out = vae.decode(z, lens)
                e1.record(s)
                torch.cuda.synchronize()
            print(f"decode B={B} F={F} {prec} fused_mlp={fused}: {e0.elapsed_time(e1) / n:.3f} ms  (max |feats| {out.abs().max().item():.3f})")
L.ladiff_debug_set_decoder_fusion(1)

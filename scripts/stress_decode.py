"""Stress (GPU box): random batch shapes through LADiffVae.decode - f16x3 default path against the same path with each fusion switched
off (attention with in_proj inside, fused feed-forward kernel, final_layer on f16x3 tiles, small-M routing), against fp32 mode
(tolerance), every call twice (same bits), padded frames exactly zero.  python scripts/stress_decode.py [cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ladiff_amd import LADiffVae, _lib, synthetic as syn
from test_abi import ABL, VAE_KW

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
dev = "cuda:0"
L = _lib.lib()
vaes = {}
for nf in (263, 251):
    v = LADiffVae(ABL, **{**VAE_KW, "nfeats": nf}); v.load_state_dict(syn.vae_weights(nf)); vaes[nf] = v.to(dev).eval()
bad = 0
for case in range(cases):
    nf = rng.choice([263, 251])
    vae = vaes[nf]
    B = rng.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 33, 64, 100, 128, 129, 200, 257])
    kind = rng.choice(["full", "mixed", "short", "bimodal", "edge"])
    lens = [196 if kind == "full" else rng.randint(1, 196) if kind == "mixed" else rng.randint(1, 60) if kind == "short"
            else rng.choice([20, 196]) if kind == "bimodal" else rng.choice([1, 31, 32, 33, 48, 49, 64, 96, 97, 192, 193, 196]) for _ in range(B)]
    vae.length_aware = rng.random() < 0.7
    z = torch.randn(5, B, 256, generator=torch.Generator().manual_seed(100 + case)).to(dev)
    for i, l in enumerate(lens):
        z[-(-l // 48):, i] = 0
    out = {}
    with torch.no_grad():
        vae.precision = "f16x3"
        for sw in (1, 1, 1 + 16, 0, 1 + 8, 1 + 4, 2 + 32, 1 + 64):
            L.ladiff_debug_set_decoder_fusion(sw)
            out.setdefault(sw, []).append(vae.decode(z, lens))
        L.ladiff_debug_set_decoder_fusion(1)
        vae.precision = "fp32"
        ref = vae.decode(z, lens)
    torch.cuda.synchronize()
    scale = max(1.0, ref.abs().max().item())
    a = out[1][0]
    msgs = []
    if not torch.isfinite(a).all(): msgs.append("non-finite")
    if not torch.equal(a, out[1][1]): msgs.append("repeat differs")
    for sw in (17, 0, 9, 5, 34, 65):
        d = (out[sw][0] - a).abs().max().item()
        if d > 1e-4 * scale: msgs.append(f"switch {sw}: {d:.2e}")
    d = (ref - a).abs().max().item()
    if d > 5e-4 * scale: msgs.append(f"vs fp32: {d:.2e}")
    for i, l in enumerate(lens):
        if l < a.shape[1] and a[i, l:].abs().max().item() != 0.0: msgs.append(f"frames past length of sample {i} not zero"); break
    if msgs:
        bad += 1
        print(f"case {case}: B={B} {kind} nfeats={nf} length_aware={vae.length_aware} rows={sum(lens)}: " + "; ".join(msgs), flush=True)
print(f"{cases} decode shapes done, {bad} bad", flush=True)
sys.exit(1 if bad else 0)

"""Diagnostic (GPU box): the length-aware 16-row plan against the launch-per-stage loop over batch shapes that stress the packing
(8 one-row prompts per block, partial last blocks, large batches).  python scripts/try_packed.py [fp32]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
from ladiff_amd import synthetic as syn
dev = torch.device("cuda", 0)
pipe = bench.build_pipe(dev, 128); pipe.precision = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
pipe.num_inference_timesteps = 4
stream = torch.cuda.Stream(device=dev)
cases = {"u196x128": [196] * 128, "u40x16": [40] * 16, "u40x17": [40] * 17, "u90x24": [90] * 24, "mix9": [196, 60, 120, 100, 48, 150, 196, 30, 77],
         "u40x40": [40] * 40, "mix512": ([196, 60, 120, 100, 48, 150, 196, 13] * 64)[:512], "u196x600": [196] * 600,
         "mix1000": ([60, 120, 196, 33, 150] * 200)[:1000]}
for name, lens in cases.items():
    B = len(lens)
    text = syn.text_embeddings(B).to(dev); noise = syn.init_noise(lens).to(dev)
    out = {}
    for loop in ("launches", "pipeline16", "pipeline32"):
        pipe.loop = loop
        with torch.cuda.stream(stream), torch.no_grad():
            out[loop] = pipe._diffusion_reverse(text, lens, init_noise=noise)
            torch.cuda.synchronize()
        st = pipe.loop_status()
    d = (out["pipeline16"] - out["launches"]).abs()
    bad = (d.amax(dim=(0, 2)) > 1e-2).nonzero().flatten().tolist()
    print(name, "max diff", f"{d.max().item():.3e}", "16 == 32 rows:", bool(torch.equal(out["pipeline16"], out["pipeline32"])), "status", st,
          "bad prompts", bad[:20], flush=True)

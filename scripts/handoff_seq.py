"""Per-call loop time over a sequence of flag (F) / tagged (T) calls of one shape: does a call's time depend on what ran before?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from ladiff_amd import _lib, synthetic as syn
if os.environ.get("LADIFF_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["LADIFF_LIB"])
dev = torch.device("cuda", 0)
pipe = bench.build_pipe(dev, 128)
pipe.precision = "f16x3"; pipe.loop = "pipeline16"; pipe.num_inference_timesteps = 50
B = int(sys.argv[1]); seq = sys.argv[2]
lens = [196] * B
text, noise = syn.text_embeddings(B).to(dev), syn.init_noise(lens).to(dev)
stream = torch.cuda.Stream(device=dev)
L = _lib.lib()
out = []
for c in seq:
    if c == "s":
        torch.cuda.synchronize(); import time; time.sleep(0.5); out.append("sleep"); continue
    _lib.check(L.ladiff_debug_set_handoff(1 if c == "T" else 0))
    with torch.cuda.stream(stream), torch.no_grad():
        pipe._diffusion_reverse(text, lens, init_noise=noise)
        torch.cuda.synchronize()
    out.append(f"{c}{pipe.loop_ms():.2f}")
print(f"B={B}: " + " ".join(out), flush=True)

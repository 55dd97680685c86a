"""16-row length-aware blocks (tagged hand-off, 8 waves) against 32-row padded blocks (flag protocol, 4 waves) per batch size: loop ms."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from ladiff_amd import _lib, synthetic as syn
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
for B in (64, 128, 192, 256):
    lens = [196] * B
    text, noise = syn.text_embeddings(B).to(dev), syn.init_noise(lens).to(dev)
    row = []
    for loop in ("pipeline16", "pipeline32", "pipeline"):
        pipe = bench.build_pipe(dev, B)
        pipe.precision = "f16x3"; pipe.loop = loop; pipe.num_inference_timesteps = 50; pipe.max_prompts_per_launch = None
        with torch.cuda.stream(stream), torch.no_grad():
            for _ in range(5):
                pipe._diffusion_reverse(text, lens, init_noise=noise)
            ms = pipe.loop_ms()
        row.append(f"{loop} {ms:7.3f} ms {pipe.last_loop()[1:]}")
    print(f"{B:4d} prompts: " + " | ".join(row), flush=True)

"""Loop + decode time per 128 prompts from 128 to 1,024 prompts in ONE call on one GPU (196 frames, 50-step DDIM): batches above
`LADIFF.max_prompts_per_launch` run as balanced chunks of <= 256 prompts (several pipeline launches), so the per-prompt cost does not
grow with the batch.  Also prints the single-launch time (chunking off) beside it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from ladiff_amd import synthetic as syn

dev = torch.device("cuda", 0)
pipe = bench.build_pipe(dev, 128)
pipe.precision = "fp32" if "fp32" in sys.argv[1:] else "f16x3"
stream = torch.cuda.Stream(device=dev)
for B in (128, 192, 256, 320, 384, 512, 768, 1024):
    lens = [196] * B
    text, noise = syn.text_embeddings(B).to(dev), syn.init_noise(lens).to(dev)
    row = [f"B={B:5d}"]
    for cap in (320, None):
        if cap is None and B > 512:
            continue
        pipe.max_prompts_per_launch = cap
        with torch.cuda.stream(stream), torch.no_grad():
            for _ in range(2):
                z = pipe._diffusion_reverse(text, lens, init_noise=noise)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(3):
                z = pipe._diffusion_reverse(text, lens, init_noise=noise)
            e1.record(stream)
            torch.cuda.synchronize()
        pipe.check()
        ms = e0.elapsed_time(e1) / 3
        row.append(f"{'chunked' if cap else 'one launch'} {len(pipe._chunks(B))} launch(es): {ms:7.2f} ms = {ms * 128 / B:6.2f} per 128 (finite {bool(torch.isfinite(z).all())})")
    print(" | ".join(row), flush=True)

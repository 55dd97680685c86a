"""Stress (GPU box): the persistent pipeline loop is deterministic - the same call must give the same bits every time.  Runs the
B = 128 mixed-length workload and a 6-prompt sub-batch alternately, with the decoder in between (other kernels around the
pipeline launches), and reports every call whose latents differ from the first one, with the loop's status word.
python scripts/stress_pipeline.py [iterations] [fp32]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from ladiff_amd import synthetic as syn

iters = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100
dev = torch.device("cuda", 0)
bad = 0
for precision in (["fp32"] if "fp32" in sys.argv[1:] else ["f16x3", "fp32"]):
    pipe = bench.build_pipe(dev, 128)
    pipe.precision = precision
    pipe.num_inference_timesteps = 50
    lens = [196] * 120 + [60, 120, 49, 1, 100, 150, 196, 48]
    idx = [0, 57, 120, 121, 123, 127]
    text, noise = syn.text_embeddings(128), syn.init_noise(lens)
    sub_text = torch.cat([text[:128][idx], text[128:][idx]]).to(dev)
    sub_lens, sub_noise = [lens[i] for i in idx], noise[idx].to(dev)
    text, noise = text.to(dev), noise.to(dev)
    ref = ref_s = None
    with torch.no_grad():
        for it in range(iters):
            z, feats = pipe.sample(text, lens, init_noise=noise)
            st = pipe.loop_status()
            zs, fs = pipe.sample(sub_text, sub_lens, init_noise=sub_noise)
            st_s = pipe.loop_status()
            if ref is None:
                ref, ref_s = z.clone(), zs.clone()
            for name, a, b, s in (("full", z, ref, st), ("sub", zs, ref_s, st_s)):
                if not torch.equal(a, b) or s != (0, 0):
                    bad += 1
                    d = (a - b).abs()
                    cols = (d.amax(dim=(0, 2)) > 0).nonzero().flatten().tolist()
                    print(f"{precision} call {it} ({name}): status {s}, max diff {d.max().item():.3e}, prompts that differ {cols[:16]} ({len(cols)})", flush=True)
    print(f"{precision}: {iters} x (full + sub-batch) calls done, {bad} bad so far", flush=True)
sys.exit(1 if bad else 0)

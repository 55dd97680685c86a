"""Two launch-per-stage loops (loop='launches': the kernels of gemm_kr / gemm_rowln / qkv_attn, LDS-DMA of activations and weights) on
two streams at once against the same two calls one after another, bit for bit, both arithmetic modes; the small decode (8 x 60 frames:
gemm_kr's path) likewise.  Exits non-zero when a concurrent run differs.  usage: launch_path_reentrancy.py [runs] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from ladiff_amd import _lib, synthetic as syn
if os.environ.get("LADIFF_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["LADIFF_LIB"])
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
pipes = [bench.build_pipe(dev, 64), bench.build_pipe(dev, 64)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
B = 64
lens = [196] * B
data = [(syn.text_embeddings(B, seed=11 + i).to(dev),
         syn.init_noise(lens).to(dev) * (1 + 0.02 * i)) for i in range(2)]
total_bad = 0
for prec in ("f16x3", "fp32"):
    for p in pipes:
        p.loop = "launches"; p.precision = prec; p.num_inference_timesteps = steps
    def both(concurrent):
        zs = []
        for i in range(2):
            with torch.cuda.stream(streams[i]), torch.no_grad():
                zs.append(pipes[i]._diffusion_reverse(data[i][0], lens, init_noise=data[i][1]).clone())
            if not concurrent: torch.cuda.synchronize()
        torch.cuda.synchronize()
        return torch.stack(zs)
    ref = both(False)
    for concurrent in (False, True):
        bad = 0; worst = 0.0
        for it in range(runs):
            d = (both(concurrent) - ref).abs().max().item()
            if d > 0: bad += 1; worst = max(worst, d)
        if concurrent: total_bad += bad
        print(f"launch-per-stage loop, 64 prompts x {steps} steps, {prec}, concurrent {concurrent}: {bad} of {runs} runs differ (worst {worst:.3e})", flush=True)
# the launch path beside a LARGE decode on the other stream (the load that spread the GEMM's LDS-DMA latencies: DESIGN.md 4b)
zbig = torch.randn(5, 128, 256, generator=torch.Generator().manual_seed(9)).to(dev)
pipes[1].vae.precision = "f16x3"
for prec in ("f16x3", "fp32"):
    pipes[0].precision = prec
    def loop_beside_decode(concurrent):
        with torch.cuda.stream(streams[1]), torch.no_grad():
            for _ in range(2): pipes[1].vae.decode(zbig, [196] * 128)
        if not concurrent: torch.cuda.synchronize()
        with torch.cuda.stream(streams[0]), torch.no_grad():
            z = pipes[0]._diffusion_reverse(data[0][0], lens, init_noise=data[0][1]).clone()
        torch.cuda.synchronize()
        return z
    ref = loop_beside_decode(False)
    for concurrent in (False, True):
        bad = 0; worst = 0.0
        for it in range(runs):
            d = (loop_beside_decode(concurrent) - ref).abs().max().item()
            if d > 0: bad += 1; worst = max(worst, d)
        if concurrent: total_bad += bad
        print(f"launch-per-stage loop beside two decodes of 128 x 196 frames, {prec}, concurrent {concurrent}: {bad} of {runs} runs differ (worst {worst:.3e})", flush=True)
# the small decode
vaes = [p.vae for p in pipes]
zz = [torch.randn(5, 8, 256, generator=torch.Generator().manual_seed(3 + i)).to(dev) for i in range(2)]
for prec in ("f16x3", "fp32"):
    for v in vaes: v.precision = prec
    def both_d(concurrent):
        outs = []
        for i in range(2):
            with torch.cuda.stream(streams[i]), torch.no_grad():
                for _ in range(3): o = vaes[i].decode(zz[i], [60] * 8)
                outs.append(o)
            if not concurrent: torch.cuda.synchronize()
        torch.cuda.synchronize()
        return torch.stack(outs)
    ref = both_d(False)
    for concurrent in (False, True):
        bad = 0
        for it in range(runs):
            if not torch.equal(both_d(concurrent), ref): bad += 1
        if concurrent: total_bad += bad
        print(f"decode 8 x 60 frames x 3, {prec}, concurrent {concurrent}: {bad} of {runs} runs differ", flush=True)
sys.exit(1 if total_bad else 0)

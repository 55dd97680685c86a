"""The launch-per-stage loop (loop='launches': N text tokens, device-only counts, the in-process fallback) and the small decode of config
c1, ms per call: the kernels of gemm_kr.hip / gemm_rowln.hip / qkv_attn.hip, which the pipeline kernel's numbers do not show.
usage: launch_path_speed.py [git tag for the log line]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from ladiff_amd import LADiffVae, _lib, synthetic as syn
from test_abi import ABL, VAE_KW
if os.environ.get("LADIFF_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["LADIFF_LIB"])
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
def timed(fn, n):
    with torch.cuda.stream(stream), torch.no_grad():
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n): fn()
        e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
pipe = bench.build_pipe(dev, 128)
pipe.loop = "launches"
for B in (128, 32):
    lens = [196] * B
    text, noise = syn.text_embeddings(B).to(dev), syn.init_noise(lens).to(dev)
    for prec in ("f16x3", "fp32"):
        pipe.precision = prec
        ms = timed(lambda: pipe._diffusion_reverse(text, lens, init_noise=noise), 3)
        print(f"{sys.argv[1] if len(sys.argv) > 1 else ''} launch-per-stage loop, {B} prompts x 50 steps, {prec}: {ms:8.2f} ms", flush=True)
vae = LADiffVae(ABL, **VAE_KW); vae.load_state_dict(syn.vae_weights(263)); vae = vae.to(dev).eval()
for B, F in ((8, 60), (16, 196)):
    z = torch.randn(5, B, 256, generator=torch.Generator().manual_seed(1)).to(dev)
    for prec in ("f16x3", "fp32"):
        vae.precision = prec
        ms = timed(lambda: vae.decode(z, [F] * B), 20)
        print(f"{sys.argv[1] if len(sys.argv) > 1 else ''} decode {B} x {F} frames (small-rows path), {prec}: {ms:8.3f} ms", flush=True)

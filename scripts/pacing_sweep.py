"""STYL's polling pace (ladiff_debug_set_pacing: it sleeps eighths / 8 of its last observed wait before it polls again; default 4) swept
at the kernels of the round's end: loop kernel ms per shape, bits against the default."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from ladiff_amd import _lib, synthetic as syn
dev = torch.device("cuda", 0)
pipe = bench.build_pipe(dev, 128)
pipe.precision = "bf16x3"; pipe.loop = "pipeline16"; pipe.num_inference_timesteps = 50
L = _lib.lib()
stream = torch.cuda.Stream(device=dev)
shapes = [(64, "u"), (128, "u"), (128, "m"), (256, "u")]
data = {}
for B, kind in shapes:
    lens = [196] * B if kind == "u" else ([196, 60, 120] * 200)[:B]
    data[(B, kind)] = (lens, syn.text_embeddings(B).to(dev), syn.init_noise(lens).to(dev))
def run(B, kind, reps=10):
    lens, text, noise = data[(B, kind)]
    with torch.cuda.stream(stream), torch.no_grad():
        for _ in range(3 + reps): z = pipe._diffusion_reverse(text, lens, init_noise=noise)
        torch.cuda.synchronize()
    return pipe.loop_ms(), z.clone()
ref = {}
for eighths, mask in ((4, 4), (0, 0), (2, 4), (6, 4), (8, 4), (4, 4 | 8), (4, 4)):
    _lib.check(L.ladiff_debug_set_pacing(eighths, mask))
    row = []
    for sh in shapes:
        ms, z = run(*sh)
        if sh not in ref: ref[sh] = z
        row.append(f"{sh[0]}{sh[1]} {ms:7.3f}{'' if torch.equal(z, ref[sh]) else ' BITS DIFFER'}")
    print(f"pace {eighths}/8 roles {mask:#x}: " + " | ".join(row), flush=True)
_lib.check(L.ladiff_debug_set_pacing(4, 4))

"""Polling-policy sweeps of the pipeline loop at the current kernels (loop kernel ms per shape):
  policy_sweep.py pace   STYL's polling pace (ladiff_debug_set_pacing: it sleeps eighths / 8 of its last observed wait before it polls
                         again; default 4 / 8, STYL only), bits against the default
  policy_sweep.py rest   the small-launch rest of the LIN / FFN workgroups (ladiff_debug_set_stage_delay: mask of stage types, length in
                         s_sleep(2) units; default LIN | FFN, 4, in launches of <= 60 blocks) at the trip-bound shapes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from ladiff_amd import _lib, synthetic as syn
mode = sys.argv[1] if len(sys.argv) > 1 else "pace"
dev = torch.device("cuda", 0)
pipe = bench.build_pipe(dev, 128)
pipe.precision = "f16x3"; pipe.loop = "pipeline16"; pipe.num_inference_timesteps = 50
L = _lib.lib()
stream = torch.cuda.Stream(device=dev)
shapes = [(64, "u"), (128, "u"), (128, "m"), (256, "u")] if mode == "pace" else [(32, "u"), (64, "u"), (100, "m"), (128, "m")]
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
if mode == "pace":
    for eighths, mask in ((4, 4), (0, 0), (2, 4), (6, 4), (8, 4), (4, 4 | 8), (4, 4)):
        _lib.check(L.ladiff_debug_set_pacing(eighths, mask))
        row = []
        for sh in shapes:
            ms, z = run(*sh)
            if sh not in ref: ref[sh] = z
            row.append(f"{sh[0]}{sh[1]} {ms:7.3f}{'' if torch.equal(z, ref[sh]) else ' BITS DIFFER'}")
        print(f"pace {eighths}/8 roles {mask:#x}: " + " | ".join(row), flush=True)
    _lib.check(L.ladiff_debug_set_pacing(-1, 0))
else:
    for mask, ln in ((-1, 0), (0, 0), (9, 2), (9, 4), (9, 6), (9, 8), (1, 4), (8, 4), (9 | 64, 4), (-1, 0)):
        _lib.check(L.ladiff_debug_set_stage_delay(mask, ln))
        print(f"rest mask {mask:3d} len {ln}: " + " | ".join(f"{B}{k} {run(B, k)[0]:7.3f}" for B, k in shapes), flush=True)
    _lib.check(L.ladiff_debug_set_stage_delay(-1, 0))

"""Loop-kernel time of the two hand-off protocols of the pipeline loop (csrc/systolic.hip) at several batch shapes, calls queued back to
back as bench.py queues its passes; every tagged result is compared bit for bit with the flag protocol's.
python scripts/handoff_knobs.py [fp32] tags=      (experiment builds of round 4 had switches: name=k0,k1,...; profiles/r4)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from ladiff_amd import _lib, synthetic as syn

if os.environ.get("LADIFF_LIB"):                      # an experiment build of the library (scripts/build_variant.sh; same ABI)
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["LADIFF_LIB"])
dev = torch.device("cuda", 0)
pipe = bench.build_pipe(dev, 128)
pipe.precision = "fp32" if "fp32" in sys.argv[1:] else "f16x3"
pipe.loop = "pipeline16"
pipe.num_inference_timesteps = 50
cfgs = [a for a in sys.argv[1:] if "=" in a]
shapes = [(64, "u"), (128, "u"), (128, "m"), (256, "u")]
if os.environ.get("SHAPES"):
    shapes = [(int(t[:-1]), t[-1]) for t in os.environ["SHAPES"].split(",")]
stream = torch.cuda.Stream(device=dev)
L = _lib.lib()
ref = {}
data = {}
for B, kind in shapes:
    lens = [196] * B if kind == "u" else ([196, 60, 120] * 200)[:B]
    data[(B, kind)] = (lens, syn.text_embeddings(B).to(dev), syn.init_noise(lens).to(dev))
def run(B, kind, reps=10):
    """Back-to-back calls with no host synchronisation in between (as bench.py queues its passes): per-call time of the whole
    _diffusion_reverse (prologue + loop), and the loop kernel's own time in the last call."""
    lens, text, noise = data[(B, kind)]
    with torch.cuda.stream(stream), torch.no_grad():
        for _ in range(4):
            z = pipe._diffusion_reverse(text, lens, init_noise=noise)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            z = pipe._diffusion_reverse(text, lens, init_noise=noise)
        e1.record(stream)
        torch.cuda.synchronize()
    assert pipe.loop_status()[0] == 0, pipe.loop_status()
    return pipe.loop_ms(), z.clone()
_lib.check(L.ladiff_debug_set_handoff(0))
row = []
for B, kind in shapes:
    ms, z = run(B, kind)
    ref[(B, kind)] = z
    row.append(f"{B}{kind} {ms:7.3f}")
print(f"{'flags':28s} " + " | ".join(row), flush=True)
_lib.check(L.ladiff_debug_set_handoff(1))
for c in cfgs:
    name, vals = c.split("=")
    for i, v in enumerate(v for v in vals.split(",") if v and hasattr(L, "ladiff_debug_set_knob")):   # experiment builds only
        _lib.check(L.ladiff_debug_set_knob(i, int(v)))
    row = []
    for B, kind in shapes:
        ms, z = run(B, kind)
        row.append(f"{B}{kind} {ms:7.3f}{'' if torch.equal(z, ref[(B, kind)]) else ' BITS DIFFER'}")
    print(f"{name + '=' + vals:28s} " + " | ".join(row), flush=True)

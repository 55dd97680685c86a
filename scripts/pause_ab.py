"""Targeted poll pauses of the tagged pipeline (ladiff_debug_set_poll_pause): loop kernel ms per batch shape, calls queued back to back;
results compared bit for bit with no pause.  python scripts/pause_ab.py mask:len [mask:len ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from ladiff_amd import _lib, synthetic as syn

dev = torch.device("cuda", 0)
L = _lib.lib()
knob = os.environ.get("KNOB", "pause")          # pause | delay | pace (pace: "eighths:0")
delay = knob == "delay"       # KNOB=delay: mask:len are ladiff_debug_set_stage_delay arguments (idle after every block)
cfgs = [(0, 0)] + [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] + [(0, 0)]
if knob not in ("delay", "none"): _lib.check(L.ladiff_debug_set_stage_delay(0, 0))       # the other knobs: against a loop without the small-launch rest

shapes = [(64, "u"), (128, "u"), (128, "m"), (256, "u")]
if os.environ.get("SHAPES"):
    shapes = [(int(t[:-1]), t[-1]) for t in os.environ["SHAPES"].split(",")]
stream = torch.cuda.Stream(device=dev)
pipe = bench.build_pipe(dev, 128)
pipe.precision = "f16x3"; pipe.loop = "pipeline16"; pipe.num_inference_timesteps = 50
ref = {}
data = {}
for B, kind in shapes:
    lens = [196] * B if kind == "u" else ([196, 60, 120] * 200)[:B]
    data[(B, kind)] = (lens, syn.text_embeddings(B).to(dev), syn.init_noise(lens).to(dev))
for mask, ln in cfgs:
    if knob == "none": pass                                                      # the library's defaults (thresholds by environment: scripts/thresholds_ab.sh)
    elif knob == "pace": _lib.check(L.ladiff_debug_set_pacing(ln, mask))          # mask:eighths
    else: _lib.check((L.ladiff_debug_set_stage_delay if delay else L.ladiff_debug_set_poll_pause)(mask, ln))
    row = []
    for B, kind in shapes:
        lens, text, noise = data[(B, kind)]
        with torch.cuda.stream(stream), torch.no_grad():
            for _ in range(8):
                z = pipe._diffusion_reverse(text, lens, init_noise=noise)
            ms = pipe.loop_ms()
        assert pipe.loop_status()[0] == 0, pipe.loop_status()
        if (B, kind) in ref:
            assert torch.equal(ref[(B, kind)], z), "different bits"
        else:
            ref[(B, kind)] = z.clone()
        row.append(f"{B}{kind} {ms:7.3f}")
    print(f"{knob} mask {mask:3d} len {ln:2d}: " + " | ".join(row), flush=True)
_lib.check(L.ladiff_debug_set_poll_pause(0, 0)); _lib.check(L.ladiff_debug_set_stage_delay(-1, 0)); _lib.check(L.ladiff_debug_set_pacing(-1, 0))          # the defaults

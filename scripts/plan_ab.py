"""Stage plans of the 16-row pipeline (ladiff_debug_set_stage_plan): loop kernel ms per batch shape, calls queued back to back, a fresh
LADIFF object (new samplers, new stage tables) per plan; results compared bit for bit.  python scripts/plan_ab.py [plans, e.g. 0,1,0,1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from ladiff_amd import _lib, synthetic as syn

dev = torch.device("cuda", 0)
L = _lib.lib()
plans = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,1,0,1").split(",")]
shapes = [(64, "u"), (128, "u"), (128, "m"), (256, "u")]
stream = torch.cuda.Stream(device=dev)
ref = {}
for plan in plans:
    _lib.check(L.ladiff_debug_set_stage_plan(plan))
    pipe = bench.build_pipe(dev, 128)
    pipe.precision = "fp32" if os.environ.get("PRECISION") == "fp32" else "f16x3"
    pipe.loop = "pipeline16"; pipe.num_inference_timesteps = 50
    row = []
    for B, kind in shapes:
        lens = [196] * B if kind == "u" else ([196, 60, 120] * 200)[:B]
        text, noise = syn.text_embeddings(B).to(dev), syn.init_noise(lens).to(dev)
        with torch.cuda.stream(stream), torch.no_grad():
            for _ in range(8):
                z = pipe._diffusion_reverse(text, lens, init_noise=noise)
            ms = pipe.loop_ms()
        assert pipe.loop_status()[0] == 0, pipe.loop_status()
        same = ""
        if (B, kind) in ref:
            same = "=" if torch.equal(ref[(B, kind)], z) else "!DIFFERENT!"
        else:
            ref[(B, kind)] = z.clone()
        row.append(f"{B}{kind} {ms:7.3f}{same}")
    print(f"plan {plan}: " + " | ".join(row), flush=True)
_lib.check(L.ladiff_debug_set_stage_plan(0))

"""A/B of the pipeline loop's two hand-off protocols on one GPU (csrc/systolic.hip): parity tags in the data against flags.
Same bits required; prints the loop kernel's time for each.  python scripts/handoff_ab.py [fp32] [B,steps ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from ladiff_amd import _lib, synthetic as syn

if os.environ.get("LADIFF_LIB"):                      # an experiment build of the library (same ABI)
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["LADIFF_LIB"])
dev = torch.device("cuda", 0)
pipe = bench.build_pipe(dev, 128)
pipe.precision = "fp32" if "fp32" in sys.argv[1:] else "f16x3"
pipe.loop = "pipeline16"
args = [a for a in sys.argv[1:] if a not in ("fp32",)]
cases = [(3, 2, "u"), (7, 5, "m"), (64, 50, "u"), (128, 50, "u"), (128, 50, "m"), (256, 50, "u")] if not args else \
    [(int(a.split(",")[0]), int(a.split(",")[1]), a.split(",")[2] if a.count(",") > 1 else "u") for a in args]
stream = torch.cuda.Stream(device=dev)
L = _lib.lib()
for B, steps, kind in cases:
    lens = [196] * B if kind == "u" else ([196, 60, 120] * 200)[:B]
    text = syn.text_embeddings(B).to(dev)
    noise = syn.init_noise(lens).to(dev)
    pipe.num_inference_timesteps = steps
    out, ms = {}, {}
    for ho in (0, 1, 0, 1):
        _lib.check(L.ladiff_debug_set_handoff(ho))
        with torch.cuda.stream(stream), torch.no_grad():
            z = pipe._diffusion_reverse(text, lens, init_noise=noise)
            torch.cuda.synchronize()
            t = []
            for _ in range(5):
                z = pipe._diffusion_reverse(text, lens, init_noise=noise)
                torch.cuda.synchronize()
                t.append(pipe.loop_ms())
        st = pipe.loop_status()
        out[ho] = z.clone(); ms[ho] = sorted(t)[len(t) // 2]
        print(f"B={B} steps={steps} {kind} handoff={'tags ' if ho else 'flags'}: loop {ms[ho]:.3f} ms (min {min(t):.3f}), status {st}, blocks {pipe.last_loop()}, finite {bool(torch.isfinite(z).all())}", flush=True)
    same = bool(torch.equal(out[0], out[1]))
    d = (out[0] - out[1]).abs().max().item()
    print(f"B={B} steps={steps} {kind}: tags == flags bit for bit: {same} (max diff {d:.3e}); tags / flags time {ms[1] / ms[0]:.3f}", flush=True)
_lib.check(L.ladiff_debug_set_handoff(1))

"""Debug aid: one tiny pipeline call per configuration; prints status instead of raising."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from ladiff_amd import _lib, synthetic as syn
if os.environ.get("LADIFF_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["LADIFF_LIB"])
dev = torch.device("cuda", 0)
L = _lib.lib()
_lib.check(L.ladiff_debug_set_pipeline_fault(-1, 100))
if "nolocal" in sys.argv[1:]:
    _lib.check(L.ladiff_debug_set_xcd_local(0))
pipe = bench.build_pipe(dev, 128)
pipe.precision = "bf16x3"; pipe.loop = "pipeline16"
stream = torch.cuda.Stream(device=dev)
for B, steps in ((3, 1), (3, 2), (1, 2), (7, 3)):
    lens = [196] * B
    text, noise = syn.text_embeddings(B).to(dev), syn.init_noise(lens).to(dev)
    pipe.num_inference_timesteps = steps
    for ho in (0, 1):
        _lib.check(L.ladiff_debug_set_handoff(ho))
        with torch.cuda.stream(stream), torch.no_grad():
            try:
                z = pipe._diffusion_reverse(text, lens, init_noise=noise)
                torch.cuda.synchronize()
                st = pipe.loop_status()
            except Exception as e:
                st = str(e)[:120]
            try:
                pipe.check()
            except Exception as e:
                pass
        print(f"B={B} steps={steps} ho={ho}: status {st}", flush=True)

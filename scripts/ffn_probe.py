"""Which stage type bounds the pipeline loop?  Diagnostic twin build: FFN and / or LIN stages run half of their MFMAs and no activation
(garbage results, timing only); loop kernel ms at several batch sizes, calls queued back to back.  python scripts/ffn_probe.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
subprocess.check_call([sys.executable, "-m", "ladiff_amd.build", "--stamps"], cwd=ROOT, stdout=subprocess.DEVNULL)
import torch
from ladiff_amd import _lib, synthetic as syn
from ladiff_amd import build as _build
_lib.LIB_PATH = _build.stamps_lib(os.environ.get("LADIFF_STAMPS_LEVEL", "1"))
import bench
dev = torch.device("cuda", 0)
L = _lib.lib()
pipe = bench.build_pipe(dev, 128)
pipe.precision = "f16x3"; pipe.loop = "pipeline16"; pipe.num_inference_timesteps = 50
stream = torch.cuda.Stream(device=dev)
for B in (64, 128, 256):
    lens = [196] * B
    text, noise = syn.text_embeddings(B).to(dev), syn.init_noise(lens).to(dev)
    row = []
    for probe in ([int(v) for v in os.environ["PROBES"].split(",")] if os.environ.get("PROBES") else (0, 1, 4, 16, 5, 20, 21, 0)):
        L.ladiff_debug_set_probe(probe)
        with torch.cuda.stream(stream), torch.no_grad():
            for _ in range(6):
                pipe._diffusion_reverse(text, lens, init_noise=noise)
            ms = pipe.loop_ms()
        row.append(f"probe {probe}: {ms:7.3f}")
    print(f"{B:4d} prompts  " + "   ".join(row) + "    (bits: 1 FFN at half its MFMAs and no GELU, 2 LIN likewise, 4 OUT's projection at a quarter, 8 STYL's product at a quarter, 16 QKV's projection at a quarter)", flush=True)
L.ladiff_debug_set_probe(0)

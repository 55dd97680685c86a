"""CLIP text tower alone on the end-to-end bench's guidance batch (128 empty + 128 prompts of <= 30 words): ms per encode, and the max
difference to the first run's output of this process (env LADIFF_GEMM_BM64 selects the split GEMM's 64-row-tile threshold per process)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ladiff_amd import synthetic as syn
from ladiff_amd import _lib
if os.environ.get("LADIFF_LIB"):                      # an experiment build of the library (same ABI)
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["LADIFF_LIB"])
from ladiff_amd.text_encoder import MldTextEncoder
dev = "cuda:0"
B = 128
ids = syn.clip_token_ids(2 * B, empty_first=B).to(dev)
enc = MldTextEncoder(precision=os.environ.get("PRECISION", "f16x3"))
enc.text_model.load_state_dict(syn.clip_weights(), strict=True)
enc = enc.to(dev).eval()
s = torch.cuda.Stream()
with torch.cuda.stream(s), torch.no_grad():
    for _ in range(3):
        out = enc.encode_ids(ids)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(10):
        out = enc.encode_ids(ids)
    e1.record(s)
    torch.cuda.synchronize()
print(f"LADIFF_GEMM_BM64={os.environ.get('LADIFF_GEMM_BM64', '-')}: {e0.elapsed_time(e1) / 10:.3f} ms per encode, checksum {out.double().sum().item():.6f} max {out.abs().max().item():.4f}")

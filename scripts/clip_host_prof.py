"""Host and device time of MldTextEncoder.encode_ids per call on the end-to-end bench's guidance batch (128 empty + 128 prompts of <= 30
words), token ids resident on the host (as a tokenizer delivers them) and on the device (costs a device -> host copy = a stream sync)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ladiff_amd import synthetic as syn
from ladiff_amd.text_encoder import MldTextEncoder
dev = torch.device("cuda:0")
B = 128
ids_h = syn.clip_token_ids(2 * B, empty_first=B)
enc = MldTextEncoder(precision=os.environ.get("PRECISION", "f16x3"))
enc.text_model.load_state_dict(syn.clip_weights(), strict=True)
enc = enc.to(dev).eval()
s = torch.cuda.Stream()
for name, x in (("host ids", ids_h), ("device ids", ids_h.to(dev))):
    for ragged in (True, False):
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(5):
                enc.encode_ids(x, ragged=ragged)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(s)
            for _ in range(20):
                out = enc.encode_ids(x, ragged=ragged)
            e1.record(s)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
        print(f"{name}, {'ragged rows' if ragged else 'padded to the longest prompt'}: host {(t1 - t0) / 20 * 1e3:.3f} ms per call, "
              f"device {e0.elapsed_time(e1) / 20:.3f} ms per call, checksum {out.double().sum().item():.6f}", flush=True)

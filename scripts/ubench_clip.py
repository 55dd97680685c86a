"""Time the CLIP text tower on the guidance batch of the benchmark (B empty + B prompts), both precisions."""
import sys, time
import torch
sys.path.insert(0, ".")
from ladiff_amd import synthetic as syn
from ladiff_amd.text_encoder import MldTextEncoder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
enc = MldTextEncoder(vocab_size=49408, num_layers=12)
enc.text_model.load_state_dict(syn.clip_weights(), strict=True)
enc = enc.to("cuda:0").eval()
ids = syn.clip_token_ids(2 * B, empty_first=B)
print("max eos", int(ids.argmax(1).max()), "unique rows", torch.unique(ids, dim=0).shape[0])
for prec in ("fp32", "bf16x3"):
    enc.precision = prec
    for label, kw in (("trunc+dedup", {}), ("full 77 x 2B", dict(full_length=True, dedup=False))):
        for _ in range(2):
            out = enc.encode_ids(ids, **kw)
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 5
        for _ in range(n):
            out = enc.encode_ids(ids, **kw)
        torch.cuda.synchronize()
        print(f"{prec:7s} {label:14s} {(time.perf_counter() - t) / n * 1e3:8.2f} ms  ({2 * B} prompts)")

"""Where the host time of MldTextEncoder.encode_ids goes (line timings of its pieces, host- and device-resident ids)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ladiff_amd import _lib, synthetic as syn
from ladiff_amd.text_encoder import MldTextEncoder
dev = torch.device("cuda:0")
B = 128
ids_h = syn.clip_token_ids(2 * B, empty_first=B)
enc = MldTextEncoder(precision="bf16x3")
enc.text_model.load_state_dict(syn.clip_weights(), strict=True)
enc = enc.to(dev).eval()
L = _lib.lib()
s = torch.cuda.Stream()
def one(x, T):
    t = [time.perf_counter()]
    def lap(name):
        t.append(time.perf_counter()); T[name] = T.get(name, 0.0) + t[-1] - t[-2]
    return enc.encode_ids(x)  # (pieces below: the torch-on-CPU form of round 5's first try)
    ids = torch.as_tensor(x).to(torch.int64)
    if ids.device.type != "cpu": ids = ids.cpu()
    lap("to cpu")
    mn, mx = int(ids.min()), int(ids.max())
    lap("min/max")
    uniq, inverse = torch.unique(ids, dim=0, return_inverse=True); ids = uniq
    lap("unique")
    n, S = ids.shape
    eos = ids.argmax(dim=1); Lx = int(eos.max()) + 1
    wt = enc.text_model._weight_table(); split = wt.split_array()
    lap("weight table")
    seq_len = (eos + 1).to(torch.int32)
    row_off = torch.zeros(n + 1, dtype=torch.int32); row_off[1:] = torch.cumsum(seq_len, 0)
    total = int(row_off[-1])
    row_seq = torch.repeat_interleave(torch.arange(n, dtype=torch.int32), seq_len.to(torch.int64))
    meta = torch.cat([seq_len, row_off, row_seq])
    if meta.numel() % 2: meta = torch.cat([meta, torch.zeros(1, dtype=torch.int32)])
    packed = torch.cat([ids.reshape(-1), meta.view(torch.int64)])
    lap("layout")
    d_packed = enc._stage(packed, dev)
    lap("stage")
    d_ids = d_packed[:n * S].view(n, S); d_meta = d_packed[n * S:].view(torch.int32)
    d_len, d_off, d_seq = d_meta[:n], d_meta[n:2 * n + 1], d_meta[2 * n + 1:2 * n + 1 + total]
    res = torch.empty(n, 768, dtype=torch.float32, device=dev)
    wsb = L.ladiff_clip_workspace_bytes_ragged(n, total)
    ws = _lib.workspace(wsb, dev)
    lap("workspace")
    _lib.check(L.ladiff_clip_text_encode_ragged(wt.array, split, 12, 49408, _lib.ptr(d_ids, torch.int64), n, S, Lx, _lib.ptr(d_len, torch.int32),
                                                _lib.ptr(d_off, torch.int32), _lib.ptr(d_seq, torch.int32), total, _lib.ptr(res), _lib.ptr(ws), wsb, _lib.stream_ptr()))
    lap("C call (launches)")
    out = res[enc._stage(inverse, dev)]
    lap("gather")
    return out
for name, x in (("host ids", ids_h), ("device ids", ids_h.to(dev))):
    T = {}
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(5): one(x, {})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): one(x, T)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
    print(f"{name}: host {(t1 - t0) / 20 * 1e3:.3f} ms per call: " + ", ".join(f"{k} {v / 20 * 1e3:.3f}" for k, v in T.items()))

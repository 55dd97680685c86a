"""Multi-GPU sharding of the sampling path: one process per GPU, prompts split contiguously, weights replicated,
no collective inside the loop, ONE all-gather of the decoded frames at the end.

The reference has no multi-device inference (test.py:100 pins one device); prompts are independent
(attention is per sample, guidance pairs row i with row i+B of the same prompt), so sharding is exact.
`backend="nccl"` is RCCL over xGMI on ROCm; the CPU tests use gloo.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group torchrun set up (RANK / WORLD_SIZE / MASTER_* / LOCAL_RANK). Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or "RANK" in os.environ) and not dist.is_initialized():   # launched by torchrun (any world size)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def shard_range(total, rank, world):
    """Contiguous [lo, hi) of `total` prompts for `rank`; the first total % world ranks take one extra."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_prompts(text_emb, lengths, init_noise, rank, world):
    """Slice a GLOBAL batch for one rank.  text_emb [2B,1,E] keeps its uncond|cond halves paired per prompt."""
    B = len(lengths)
    lo, hi = shard_range(B, rank, world)
    text = torch.cat([text_emb[:B][lo:hi], text_emb[B:][lo:hi]], dim=0)
    noise = None if init_noise is None else init_noise[lo:hi]
    return text, list(lengths[lo:hi]), noise, (lo, hi)


_PAD_CACHE = {}


def _pad_buffer(shape, dtype, dev):
    """Pre-allocated, zeroed staging buffer per (shape, dtype, device): no allocation inside a timed pass."""
    key = (tuple(shape), dtype, str(dev))
    buf = _PAD_CACHE.get(key)
    if buf is None:
        if len(_PAD_CACHE) >= 8:
            _PAD_CACHE.pop(next(iter(_PAD_CACHE)))
        buf = torch.zeros(*shape, dtype=dtype, device=dev)
        _PAD_CACHE[key] = buf
    return buf


def gather_feats(feats, total, world, out=None, lengths=None):
    """All-gather per-rank frames [b_r, F_r, C] into [total, F_max, C] in global prompt order: ONE collective.

    `lengths` = the GLOBAL list of frame counts (every rank has it: prompts are sharded from one global batch by
    `shard_range`).  With it every rank derives all shard shapes on the host, so the pass issues exactly one
    `all_gather_into_tensor`, no metadata exchange and no host synchronisation; a shard that already has the global
    shape (uniform batches: the benchmark) goes into the collective directly, otherwise through a pre-allocated
    zero-padded staging buffer (frames past a motion's length are zero anyway, ladiff_vae.py:358).
    Without `lengths` the shard shapes are exchanged first (one extra small collective + a host sync).
    """
    if not dist.is_initialized():
        return feats
    dev = feats.device
    C = feats.shape[2]
    if lengths is not None:
        if len(lengths) != total:
            raise ValueError(f"{len(lengths)} global lengths for {total} prompts")
        spans = [shard_range(total, r, world) for r in range(world)]
        bs = [hi - lo for lo, hi in spans]
        fmax = max(lengths)
    else:
        meta = torch.tensor([feats.shape[0], feats.shape[1]], dtype=torch.int64, device=dev)
        metas = [torch.empty_like(meta) for _ in range(world)]
        dist.all_gather(metas, meta)
        bs = [int(m[0]) for m in metas]
        fmax = max(int(m[1]) for m in metas)
    bmax = max(bs)
    if feats.shape[0] == bmax and feats.shape[1] == fmax and feats.is_contiguous():
        src = feats
    else:
        src = _pad_buffer((bmax, fmax, C), feats.dtype, dev)
        src.zero_()
        src[:feats.shape[0], :feats.shape[1]] = feats
    buf = torch.empty(world * bmax, fmax, C, dtype=feats.dtype, device=dev) if out is None else out
    dist.all_gather_into_tensor(buf, src)
    if all(b == bmax for b in bs):
        return buf
    return torch.cat([buf[r * bmax:r * bmax + bs[r]] for r in range(world)], dim=0)

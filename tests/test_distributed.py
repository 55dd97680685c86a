"""world_size-2 gloo tests (CPU) of the prompt sharding + final gather used for N>1 GPUs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ladiff_amd import distributed as D, synthetic as syn


def test_shard_ranges_cover_and_are_disjoint():
    for total in (1, 7, 128, 1024, 1023):
        for world in (1, 2, 4, 8):
            spans = [D.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_shards_keep_cfg_pairs_and_global_noise():
    B = 10
    lens = syn.mixed_lengths(B)
    text, noise = syn.text_embeddings(B), syn.init_noise(lens)
    seen = []
    for r in range(4):
        t, l, n, (lo, hi) = D.shard_prompts(text, lens, noise, r, 4)
        b = hi - lo
        assert torch.equal(t[:b], text[lo:hi]) and torch.equal(t[b:], text[B + lo:B + hi])
        assert torch.equal(n, noise[lo:hi]) and l == lens[lo:hi]
        # sliced global noise == noise drawn with offset/total (what a rank generates for itself)
        assert torch.equal(n, syn.init_noise(l, offset=lo, total=B))
        seen += l
    assert seen == lens


def _worker(rank, world, port, total, ret, with_lengths=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = D.init_from_env("gloo")
    lens = syn.mixed_lengths(total)
    lo, hi = D.shard_range(total, r, w)
    my = lens[lo:hi]
    # stand-in for the decoded frames of this rank: value encodes (global prompt, frame), zero past the length
    F = max(my)
    feats = torch.zeros(len(my), F, 3)
    for i, l in enumerate(my):
        feats[i, :l] = (lo + i) * 1000 + torch.arange(l, dtype=torch.float32)[:, None]
    calls = {"all_gather": 0, "into": 0}
    real_ag, real_into = dist.all_gather, dist.all_gather_into_tensor
    dist.all_gather = lambda *a, **k: (calls.__setitem__("all_gather", calls["all_gather"] + 1), real_ag(*a, **k))[1]
    dist.all_gather_into_tensor = lambda *a, **k: (calls.__setitem__("into", calls["into"] + 1), real_into(*a, **k))[1]
    try:
        out = D.gather_feats(feats, total, w, lengths=lens if with_lengths else None)
    finally:
        dist.all_gather, dist.all_gather_into_tensor = real_ag, real_into
    # with the global lengths: exactly ONE collective per pass
    if with_lengths and (calls["all_gather"] != 0 or calls["into"] != 1):
        ret[rank] = False
        return
    ok = out.shape == (total, max(lens), 3)
    for i, l in enumerate(lens):
        ok = ok and bool((out[i, :l, 0] == i * 1000 + torch.arange(l)).all()) and bool((out[i, l:] == 0).all())
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total,with_lengths", [(8, False), (7, False), (8, True), (7, True)])
def test_gather_feats_world2_gloo(total, with_lengths):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, total, ret, with_lengths), nprocs=2, join=True)
    assert ret[0] and ret[1]

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


# ---------------------------------------------------------------- bench.py's own pass plumbing at world size 4, uneven shards
class _MockPipe:
    """Stands in for LADIFF on a CPU rank: `sample` returns frames that encode which text row / noise row / length each prompt
    was given, so the gathered result shows that every rank ran ITS slice of the global batch, in global order."""
    class _Vae:
        def decode(self, z, lens):
            F = max(lens)
            out = torch.zeros(len(lens), F, 251)
            for i, l in enumerate(lens):
                out[i, :l, 0] = z[0, i, 0]
            return out
    vae = _Vae()

    def __init__(self):
        self.calls = 0
        self.noise_first_prompt = -1
        self.first_prompt_seen = None

    def sample(self, text, lens, init_noise=None, step_noise=None, noise_seed=None):
        self.calls += 1
        self.first_prompt_seen = self.noise_first_prompt          # bench sets the rank's global prompt offset: keys the device noise
        B = len(lens)
        assert text.shape == (2 * B, 1, 768) and init_noise.shape == (B, 5, 256)
        F = max(lens)
        feats = torch.zeros(B, F, 251)
        for i, l in enumerate(lens):
            feats[i, :l, 0] = text[i, 0, 0]               # unconditional row of prompt i
            feats[i, :l, 1] = text[B + i, 0, 0]           # its conditional row: the pair stays together on one rank
            feats[i, :l, 2] = init_noise[i, 0, 0]
            feats[i, :l, 3] = float(l)
        return None, feats


def _bench_worker(rank, world, port, total, ret):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = D.init_from_env("gloo")
    calls = {"into": 0, "other": 0}
    real_into, real_ag = dist.all_gather_into_tensor, dist.all_gather
    dist.all_gather_into_tensor = lambda *a, **k: (calls.__setitem__("into", calls["into"] + 1), real_into(*a, **k))[1]
    dist.all_gather = lambda *a, **k: (calls.__setitem__("other", calls["other"] + 1), real_ag(*a, **k))[1]
    try:
        wl = bench.Workload(bench.CONFIGS["c5"], "cpu", r, w, total=total)        # mixed {60,120,196}, 251 features
        pipe = _MockPipe()
        out = wl.one_pass(pipe)
        out2 = wl.one_pass(pipe)
    finally:
        dist.all_gather_into_tensor, dist.all_gather = real_into, real_ag
    lens = syn.mixed_lengths(total)
    gtext, gnoise = syn.text_embeddings(total), syn.init_noise(lens)
    ok = out.shape == (total, max(lens), 251) and calls == {"into": 2, "other": 0} and pipe.calls == 2 and torch.equal(out, out2)
    ok = ok and wl.glens == lens and (wl.lo, wl.hi) == D.shard_range(total, r, w) and wl.B == wl.hi - wl.lo
    ok = ok and pipe.first_prompt_seen == wl.lo
    for i, l in enumerate(lens):
        want = torch.tensor([gtext[i, 0, 0], gtext[total + i, 0, 0], gnoise[i, 0, 0], float(l)])
        ok = ok and bool((out[i, :l, :4] == want).all()) and bool((out[i, l:] == 0).all())
    ok = ok and torch.equal(wl.local_rows(out), out[wl.lo:wl.hi])
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [16, 21, 1024])
def test_bench_pass_plumbing_world8_config_c5(total):
    """BASELINE config c5's 8-GPU form on 8 gloo ranks: `Workload.one_pass` with the config's cycling lengths {60,120,196}.  1,024 = the
    config's stated batch (128 prompts per rank); 16 / 21 prompts give ranks whose own longest motion is SHORTER than the batch's (rank
    slices of 2 - 3 prompts: the frames of such a rank go through gather_feats' staging buffer) and uneven shards (21 -> 3,3,3,3,3,2,2,2)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bench_worker, args=(8, port, total, ret), nprocs=8, join=True)
    assert all(ret[r] for r in range(8)), dict(ret)


@pytest.mark.parametrize("total", [10, 7])
def test_bench_pass_plumbing_world4_uneven_shards(total):
    """bench.py's `Workload.one_pass` (what the timed loop calls) on 4 gloo ranks with a batch that does not divide evenly
    (10 -> 3,3,2,2; 7 -> 2,2,2,1): global inputs sliced per rank, guidance pairs kept together, ONE all_gather_into_tensor per
    pass, frames back in global prompt order with zeros past each length."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bench_worker, args=(4, port, total, ret), nprocs=4, join=True)
    assert all(ret[r] for r in range(4)), dict(ret)


def test_bench_n_gt_1_line_refers_to_a_recorded_cpu_baseline(tmp_path, monkeypatch):
    """The N > 1 bench line carries a `cpu_baseline` object: the newest N = 1 line of the config under profiles/, marked as not
    measured in this run - or an explicit "not available" record."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    cb = bench.recorded_cpu_baseline("headline")
    assert cb["measured_in_this_run"] is False and cb["kind"] == "port" and "sample" in cb
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    cb = bench.recorded_cpu_baseline("headline")
    assert cb["value"] is None and cb["measured_in_this_run"] is False
    os.makedirs(tmp_path / "profiles" / "r9")
    line = {"n_gpus": 1, "cpu_baseline": {"value": 17.5, "unit": "motions/s", "cores": 16, "kind": "port", "sample": "x", "gpu_over_cpu": 500}}
    (tmp_path / "profiles" / "r9" / "bench_headline_n1.json").write_text(__import__("json").dumps(line) + "\n")
    cb = bench.recorded_cpu_baseline("headline")
    assert cb["value"] == 17.5 and cb["cores"] == 16 and "gpu_over_cpu" not in cb and "bench_headline_n1.json" in cb["source"]


# ---------------------------------------------------------------- a plain `python bench.py --gpus N` starts the launcher itself
def _run_bench(args, env_extra):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], capture_output=True, text=True, env=env, timeout=300)


def test_bench_self_launch_builds_the_torchrun_command_without_touching_torch():
    """`python bench.py --gpus 8` is how the driver calls it.  The parent must start
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 ... bench.py <same args>` as a child
    and must not have imported torch (or the HIP library) when it does."""
    import json
    r = _run_bench(["--gpus", "8", "--steps", "5", "--warmup", "2", "--config", "c4"], {"LADIFF_BENCH_PRINT_LAUNCH": "1"})
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    cmd = d["cmd"]
    assert d["torch_imported"] is False
    assert cmd[1:5] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8"]
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(next(c for c in cmd if c.endswith("bench.py")))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2", "--config", "c4"]
    # --gpus=N spelling; N = 1 and a process that already is a rank do not launch anything
    r = _run_bench(["--gpus=2"], {"LADIFF_BENCH_PRINT_LAUNCH": "1"})
    assert json.loads(r.stdout)["cmd"][4] == "--nproc-per-node=2"
    import bench
    assert bench.self_launch(["--gpus", "1"]) is None
    os.environ["WORLD_SIZE"] = "2"
    try:
        assert bench.self_launch(["--gpus", "2"]) is None
    finally:
        del os.environ["WORLD_SIZE"]


def test_bench_self_launch_propagates_the_ranks_failure():
    """No GPU here: both ranks stop with "not enough devices", torchrun exits non-zero and so does the parent - and NOT with the
    old `--gpus 2 but WORLD_SIZE=1` SystemExit."""
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode != 0
    assert "not enough devices" in r.stderr
    assert "WORLD_SIZE=1" not in r.stderr
    assert r.stdout.strip() == ""


def _scale_worker(rank, world, port, ret):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = D.init_from_env("gloo")
    bench.recorded_n1_value = lambda name: (11000.0, "profiles/rX/bench_headline_n1.json")
    rep = bench.scale_report("cpu", r, w, True, loop_ms=9.0 + 0.1 * r, gather_ms=0.5 + 0.01 * r, pass_ms=11.0 + r,
                             motions_per_s=8 * 11000.0 * 0.9, config_name="headline")
    ret[rank] = rep
    dist.barrier()
    dist.destroy_process_group()


def test_scale_report_world8_validates_itself():
    """The fields an N > 1 bench line carries so that the first multi-GPU run checks itself (bench.scale_report) on 8 gloo ranks:
    `ranks_seen` from an all-reduce of ones, min / max / slowest rank of the per-rank loop, gather and pass times, efficiency against
    the recorded N = 1 line; only rank 0 gets the report."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_scale_worker, args=(8, port, ret), nprocs=8, join=True)
    assert all(ret[r] is None for r in range(1, 8))
    rep = ret[0]
    assert rep["ranks_seen"] == 8 and rep["ranks_expected"] == 8 and rep["ranks_ok"] is True
    assert rep["loop_kernel_ms_over_ranks"] == {"min": 9.0, "max": 9.7, "slowest_rank": 7}
    assert rep["final_gather_ms_over_ranks"] == {"min": 0.5, "max": 0.57, "slowest_rank": 7}
    assert rep["pass_device_ms_over_ranks"]["max"] == 18.0
    assert rep["scaling_efficiency_vs_recorded_n1"] == 0.9 and rep["n1_value"] == 11000.0

"""The random-shape stress programs of scripts/ as GPU tests (short runs; the long runs are logged under profiles/r3/16_*).  Each runs in a
child process of its own - they build their own objects and switch process-wide measurement switches - and exits non-zero on the first
shape that disagrees."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), *map(str, args)], cwd=ROOT, capture_output=True, text=True,
                       timeout=timeout)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-12:])
    assert r.returncode == 0, tail
    return r.stdout


@pytest.mark.slow
def test_random_decode_shapes():
    """LADiffVae.decode on 80 random batches (1 ... 257 samples, five length patterns, both feature counts, ragged and padded): the default
    f16x3 path against each fusion switched off, against fp32 mode, twice (same bits), frames past each length exactly zero."""
    out = _run("stress_decode.py", 80, 17)
    assert "80 decode shapes done, 0 bad" in out


@pytest.mark.slow
def test_random_loop_shapes_and_forms():
    """The reverse loop on random batches (1 ... 520 prompts, chunked batches included, with and without guidance), both arithmetic modes:
    pipeline against launch-per-stage loop, repeat identical, 16- against 32-row plan identical, status clean after every call - the run
    that found the stale step graph (api.hip, g_graph_epoch)."""
    out = _run("stress_shapes.py", 24, 7)
    assert "fp32: 24 shapes done, 0 bad so far" in out


@pytest.mark.slow
def test_random_sample_sequences_on_one_object():
    """One LADIFF object through 60 random (arithmetic mode, loop form, guidance, batch shape, steps) calls of sample(), each against a
    second object on the fp32 launch-per-stage path."""
    out = _run("stress_sample.py", 60, 29)
    assert "60 sample() cases done, 0 bad" in out


def test_old_graph_execs_replay_correctly_without_the_epoch_rule():
    """The round-3 fault, bisected in round 4 (api.hip, g_graph_epoch; profiles/r4/06_*): an OLDER exec's memset node was what the runtime
    replayed wrongly.  No graph of this library holds a memset node any more, so the formerly failing sequence - plan A, two other plans
    with blocking status reads in between, plan A again from its old execs - must be clean with the re-instantiation rule switched OFF."""
    env = dict(os.environ, LADIFF_GRAPH_EPOCH_OFF="1", STATUS="1")
    for _ in range(2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "repro_graph.py"), "plans"], cwd=ROOT, capture_output=True, text=True,
                           timeout=300, env=env)
        lines = [l.split() for l in r.stdout.splitlines() if l.startswith("ok ")]
        assert r.returncode == 0 and len(lines) == 4, (r.stdout + r.stderr)[-600:]
        assert lines[0][:3] == lines[3][:3]                           # plan A's result: the same before and after the other plans
        # ... and it really came from plan A's OLD execs: the last call instantiated nothing (the Python loop owner draws a fresh noise
        # seed per call; a generator that is off must not be part of the graph key - ADVICE r4)
        assert lines[3][-1] == lines[2][-1], lines


def test_two_decodes_on_two_streams_are_the_decodes_alone():
    """include/ladiff_hip.h: calls are re-entrant across distinct (workspace, stream) pairs.  Two decodes of 64 x 196 frames launched back
    to back on two streams, 400 times, default fusion and every fusion off: each pair must give the bits the two calls give one after
    another.  With the GEMM's counted `vmcnt` waits of rounds 1 - 4 (two LDS-DMA stages in flight) 1 - 2 % of such pairs had a wrong
    128-row tile in the first-launched decode (csrc/gemm_big.hip header); the second kernel's traffic is what makes LDS-DMA requests
    complete out of issue order."""
    out = _run("decode_reentrancy.py", 400, 1, 80)
    assert out.count("concurrent True: 0 of 400 runs differ") == 2, out


def test_dec_mlp_beside_cache_sweeping_kernels():
    """csrc/dec_mlp.hip streams its weights through LDS-DMA stages; since round 6 a wave has ONE batch of them in flight and waits with
    vmcnt(0) (rounds 3 - 5: six stages in flight behind counted waits - the pattern that broke gemm_big beside a second stream).  The
    aggressors here are chosen for THIS kernel: 384-MB copies sweep the L2 and the memory-side cache so that its weight pieces miss
    again and again, split GEMMs stream the same weight rows from 196 other workgroups.  1600 launches per aggressor against the
    first launch's bits."""
    out = _run("mlp_under_memory_pressure.py", 1600)
    assert out.count(": 0 of 1600 launches differ") == 3, out


def test_decode_writes_only_inside_its_workspace_and_output():
    """Workspace and output inside larger canary-filled buffers (64 MB of guard words on each side), four batch shapes, both modes."""
    out = _run("decode_guard.py")
    assert "guard zones touched: 0" in out


def test_launch_path_and_small_decode_on_two_streams():
    """The launch-per-stage loop (gemm_kr / gemm_kp / gemm_rowln / qkv_attn: LDS-DMA of activations and weights) on two streams at once,
    beside a large decode, and the 8 x 60 decode on two streams: the bits of the same calls one after another, both arithmetic modes.
    (These kernels were never SEEN wrong with their counted waits - profiles/r5/19_* - they got the exact ones on principle)"""
    out = _run("launch_path_reentrancy.py", 40, 3)
    assert out.count("concurrent True: 0 of 40 runs differ") == 6, out


def test_captures_beside_the_process_group_watchdog():
    """With a process group up (RCCL, world size 1): a collective on the stream and at once a call that captures several ms of graphs,
    100 times.  The library captures on a stream of the handle's own; with captures on the caller's stream the watchdog thread's
    hipEventQuery of the collective's end event (it belongs to the current stream) was refused, the capture invalidated and the process
    ended - within 150 pairs, and once in ~15 bench runs under torchrun (profiles/r5/26_*)."""
    out = _run("capture_vs_watchdog.py", 100, timeout=600)
    assert "100 collective + capture pairs done" in out

#!/usr/bin/env python3
"""Benchmark of the LADiff sampling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch of synthetic prompts: 128 prompts per GPU, 196 frames,
50-step DDIM with classifier-free guidance 7.5 (`_diffusion_reverse`), LA-VAE decode to [128,196,263] and, for
N > 1, the final RCCL all-gather of the frames.  Inputs (random-init weights, random "CLIP" embeddings, seeded
noise: ladiff_amd/synthetic.py) are resident in HBM before the timed region.  Metric: motions/s, whole job.

Extra objects on the JSON line:
  roofline     the DOMINANT KERNEL.  bf16x3 mode: the persistent pipeline kernel that runs all 50 guided steps in one launch
               (csrc/systolic.hip, ~85 % of the device time): achieved = algorithmic FLOPs of one launch (reference-equivalent
               denoiser arithmetic, SURVEY.md §8d: 358.27 MFLOP per motion and step) / its duration measured live with HIP
               events recorded around the launch on the bench stream (ladiff_sampler_loop_ms).  fp32 mode: ffn.linear1's GEMM,
               timed live over back-to-back launches.  peak = dense MFMA peak of the timed mode's dtype.  traffic, mfma_util_pmc
               and share_of_pass are READ from profiles/r2/summary.json (rocprofv3 runs of scripts/profile_pass.py, stamped
               with the commit they were taken at) - null when that file has no entry for the kernel.
               roofline.whole_pass: the same for the whole pass - reference-equivalent FLOPs (21.757 GFLOP
               per motion at F=196, C=263, 50 steps) / the pass's device time; executed_tflops counts the FLOPs the
               kernels really execute after hoisting so the two cannot be conflated.
  cpu_baseline the CPU oracle (oracle/ladiff_oracle.py, a port of the reference's op sequence, fp32 PyTorch) timed on
               the host cores of this box on a bounded sample of the same workload.  Baseline only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ladiff_amd import LADIFF, DDIMScheduler, LADiffDenoiser, LADiffVae, distributed as D, synthetic as syn  # noqa: E402

FRAMES, NFEATS, STEPS_DDIM, BATCH = 196, 263, 50, 128
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: bf16 MFMA, dense (not the 2:1 sparsity figure)


def ref_flops_per_motion(F=FRAMES, C=NFEATS, n_steps=STEPS_DDIM):
    """Reference-equivalent FLOPs (flop-counted on the reference modules; CFG x2 included)."""
    return n_steps * 358.27e6 + (17609728 + 512 * C) * F + 9216 * F * F + 11796480


def executed_flops_per_motion(F=FRAMES, C=NFEATS, n_steps=STEPS_DDIM, T=5):
    """MFMA FLOPs the kernels execute per motion: per latent row and layer 2*256*(768+256+2*1024+256
    +2*1024+256) = 2.753 MFLOP (+ skip 512x256 on 4 of 9 layers), 2 branches x T rows; decoder as the reference."""
    per_row_layer = 2 * 256 * (768 + 256 + 2048 + 256 + 2048 + 256)
    den = n_steps * 2 * T * (9 * per_row_layer + 4 * 2 * 512 * 256)
    dec_row_layer = 2 * 256 * (768 + 256 + 256 + 256 + 2048)
    attn = 2 * 2 * 224 * 224 * 64 * 4 / F          # QK^T + PV on 32-padded tiles, per frame row
    dec = F * (9 * (dec_row_layer + attn) + 4 * 2 * 512 * 256 + 2 * 256 * C)
    return den + dec


def build_pipe(dev, batch):
    from test_abi import ABL, DEN_KW, VAE_KW
    den = LADiffDenoiser(ABL, **DEN_KW)
    den.load_state_dict(syn.denoiser_weights())
    vae = LADiffVae(ABL, **VAE_KW)
    vae.load_state_dict(syn.vae_weights(NFEATS))
    sch = DDIMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                        clip_sample=False, set_alpha_to_one=False, steps_offset=1)
    return LADIFF(denoiser=den.to(dev).eval(), vae=vae.to(dev).eval(), scheduler=sch, guidance_scale=7.5,
                  num_inference_timesteps=STEPS_DDIM, eta=0.0)


def dominant_kernel_roofline(dev, stream, precision, summary, launches=400):
    """Live HIP-event timing of the kernel that dominates the pass (profiles/r1: gemm_kp_kernel<80,64,2,2> / gemm_kr_kernel<80,64,...>, ~36-40 % of the
    device time): the denoiser's 256->1024 linear (ffn.linear1, GELU) at M = 2*128*5 rows, launched back to back on the
    bench stream, in the arithmetic of the timed mode."""
    from ladiff_amd import _lib
    L = _lib.lib()
    M, N, K = 2 * BATCH * 5, 1024, 256
    split = 1 if precision == "bf16x3" else 0
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / 16; b = torch.randn(N, device=dev)
    Y = torch.empty(M, N, device=dev)
    with torch.cuda.stream(stream):
        if split:
            As, Ws = torch.empty_like(A), torch.empty_like(W)
            _lib.check(L.ladiff_split_rows(A.data_ptr(), As.data_ptr(), M, K, stream.cuda_stream))
            _lib.check(L.ladiff_split_rows(W.data_ptr(), Ws.data_ptr(), N, K, stream.cuda_stream))
            A, W = As, Ws
        args = (A.data_ptr(), K, None, 0, K, W.data_ptr(), K, b.data_ptr(), None, 0, None if split else Y.data_ptr(), N, M, N, K,
                2, split, Y.data_ptr() if split else None, stream.cuda_stream)
        for _ in range(20):
            _lib.check(L.ladiff_gemm_resident(*args))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(launches):
            L.ladiff_gemm_resident(*args)
        e1.record(stream)
        torch.cuda.synchronize(dev)
    us = e0.elapsed_time(e1) * 1e3 / launches
    flops = 2.0 * M * N * K
    peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
    mfma_flops = flops * (3 if split else 1)
    return {"name": ("gemm_kp_kernel<80,64,2,2>" if split else "gemm_kr_kernel<80,64,1,4,16>") + " (ffn.linear1: M=1280, N=1024, K=256, GELU)",
            "flops_per_launch": flops, "mfma_flops_per_launch": mfma_flops, "us_per_launch": round(us, 2),
            "achieved": round(flops / us / 1e6, 2), "unit": "TFLOP/s", "peak": peak,
            "frac": round(flops / us / 1e6 / peak, 4), "mfma_frac": round(mfma_flops / us / 1e6 / peak, 4),
            **dict(zip(("traffic", "mfma_util_pmc", "share_of_pass", "traffic_source"),
                       profiled(summary, "gemm_kp_kernel<80, 64" if split else "gemm_kr_kernel<80, 64")))}


def profile_summary():
    """profiles/rN/summary.json of the newest round that has one (written by scripts/pmc_summary.py)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "summary.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            d["_path"] = os.path.relpath(path, ROOT)
            return d
        except Exception:
            continue
    return None


def profiled(summary, kernel_key):
    """(traffic bytes per launch, mfma_util_pmc, share_of_pass, source) of the kernel whose profiled name contains `kernel_key`."""
    if summary is None:
        return None, None, None, "no profiles/r*/summary.json in this tree"
    for name, e in summary.get("kernels", {}).items():
        if kernel_key in name:
            tr = None
            if "fetch_bytes_per_launch" in e and "write_bytes_per_launch" in e:
                tr = e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"]
            return tr, e.get("mfma_util_pmc"), e.get("share_of_pass"), f"{summary['_path']} (commit {summary.get('git_sha', '?')}): {name}"
    return None, None, None, f"{summary['_path']} has no kernel matching {kernel_key!r}"


def pipeline_kernel_roofline(pipe, loop_ms_samples, summary):
    """The persistent pipeline kernel: one launch = 50 guided steps on the rank's 128 prompts."""
    ms = sorted(loop_ms_samples)[len(loop_ms_samples) // 2]
    flops = BATCH * STEPS_DDIM * 358.27e6                      # reference-equivalent (guidance x2 included), SURVEY.md §8d
    per_row_layer = 2 * 256 * (768 + 256 + 2048 + 256 + 2048 + 256)
    mfma_flops = 3.0 * BATCH * STEPS_DDIM * 2 * 5 * (9 * per_row_layer + 4 * 2 * 512 * 256)     # executed, 3 bf16 MFMAs per product
    traffic, util, share, src = profiled(summary, "systolic_loop_kernel")
    return {"bound": "mfma", "achieved": round(flops / ms / 1e9, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(flops / ms / 1e9 / PEAK_BF16_MFMA_TFLOPS, 4), "traffic": traffic,
            "kernel": "systolic_loop_kernel (all 50 guided DDIM steps of 128 prompts in one persistent launch; the profiled instantiation is named in traffic_source)",
            "us_per_launch": round(ms * 1e3, 1), "flops_per_launch": flops, "mfma_flops_per_launch": mfma_flops,
            "mfma_frac": round(mfma_flops / ms / 1e9 / PEAK_BF16_MFMA_TFLOPS, 4), "mfma_util_pmc": util, "share_of_pass": share,
            "traffic_source": src}


def cpu_baseline(sample_b):
    """Oracle on the host cores, bounded sample: `sample_b` motions of the same shape (196 frames, 50 steps)."""
    from oracle import ladiff_oracle as orc
    lens = [FRAMES] * sample_b
    # the reference path is a chain of small fp32 ops: it stops scaling (and then slows down) beyond ~16 threads
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    text, noise = syn.text_embeddings(sample_b), syn.init_noise(lens)
    den_sd, vae_sd = syn.denoiser_weights(), syn.vae_weights(NFEATS)
    with torch.no_grad():
        warm = torch.cat([text[:2], text[sample_b:sample_b + 2]])
        orc.sample_motions(den_sd, vae_sd, warm, lens[:2], noise[:2], 2, "ddim")   # warm-up (threads, allocator)
        t0 = time.perf_counter()
        orc.sample_motions(den_sd, vae_sd, text, lens, noise, STEPS_DDIM, "ddim")
        dt = time.perf_counter() - t0
    return {"value": sample_b / dt, "unit": "motions/s", "cores": torch.get_num_threads(), "kind": "port",
            "host_cpus": os.cpu_count(),
            "sample": f"{sample_b} motions, 196 frames, 50-step DDIM + decode, fp32 PyTorch CPU oracle, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="prompts per GPU")
    ap.add_argument("--cpu-sample", type=int, default=32, help="motions in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--precision", default="bf16x3", choices=["fp32", "bf16x3"],
                    help="matrix-product arithmetic of the timed mode (DESIGN.md §1); the other mode is timed after it. "
                         "BASELINE.json's 'bf16' (config c2) and 'fp16' (c5) labels are served by bf16x3: plain bf16 / fp16 operands "
                         "miss the 1e-3 decoded-frame gate by 50x / 7x on this network (DESIGN.md §1), three bf16 MFMAs per product do not")
    ap.add_argument("--loop", default="pipeline", choices=["pipeline", "launches"],
                    help="bf16x3 mode: the 50 steps as one persistent pipeline kernel (default) or as hipGraph replays of one launch per stage")
    args = ap.parse_args()

    # the JSON line must be the ONLY thing on stdout: RCCL prints a version banner to fd 1 when the process group comes
    # up, so everything but the final line goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank, world, local = D.init_from_env()
    use_dist = torch.distributed.is_available() and torch.distributed.is_initialized()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU implementation")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    B = args.batch
    total = B * world
    lens = [FRAMES] * B
    glens = [FRAMES] * total          # every rank knows the global lengths: the gather needs no metadata exchange
    # global inputs, sliced per rank: results do not depend on the sharding
    lo, hi = D.shard_range(total, rank, world)
    gtext = syn.text_embeddings(total)
    text = torch.cat([gtext[:total][lo:hi], gtext[total:][lo:hi]]).to(dev)
    noise = syn.init_noise(lens, offset=lo, total=total).to(dev)
    pipe = build_pipe(dev, B)
    pipe.loop = args.loop
    gather_buf = torch.empty(total, FRAMES, NFEATS, device=dev) if use_dist else None

    stream = torch.cuda.Stream(device=dev)        # hipGraph capture needs a non-null stream; events go on it too

    def one_pass():
        z, feats = pipe.sample(text, lens, init_noise=noise)
        if use_dist:      # final gather of the decoded frames (RCCL over xGMI); also exercised at world size 1 under torchrun
            feats = D.gather_feats(feats, total, world, out=gather_buf, lengths=glens)
        return feats

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    def timed(precision, steps, warmup):
        """W warm-up passes, then exactly K passes between fences; returns (max-over-ranks wall s, device ms, frames)."""
        pipe.precision = precision
        with torch.cuda.stream(stream), torch.no_grad():
            for _ in range(warmup):
                one_pass()
            fence()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            ev0.record(stream)
            for _ in range(steps):
                feats = one_pass()
            ev1.record(stream)
            fence()
            wall = time.perf_counter() - t0
        tmax = torch.tensor([wall], dtype=torch.float64, device=dev)
        if use_dist:
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        return float(tmax.item()), ev0.elapsed_time(ev1), feats.clone()

    wall, dev_ms, feats = timed(args.precision, args.steps, args.warmup)
    assert torch.isfinite(feats).all()
    # device time of the N-step loop alone (HIP events around it on the bench stream), sampled outside the timed region
    loop_ms = []
    with torch.cuda.stream(stream), torch.no_grad():
        for _ in range(5):
            one_pass()
            loop_ms.append(pipe.loop_ms())
    status = pipe.loop_status()
    if status[0] != 0:
        raise SystemExit(f"pipeline loop aborted: status {status}")
    other = "fp32" if args.precision == "bf16x3" else "bf16x3"
    o_wall, o_dev_ms, o_feats = timed(other, max(2, args.steps // 2), 1)        # second mode: shorter, reported beside
    mode_diff = (feats - o_feats).abs().max().item()

    if rank == 0:
        summary = profile_summary()
        motions_per_s = total * args.steps / wall
        dev_s_per_pass = dev_ms / 1e3 / args.steps
        ref_tf = B * ref_flops_per_motion() / dev_s_per_pass / 1e12
        exe_tf = B * executed_flops_per_motion() / dev_s_per_pass / 1e12
        peak = PEAK_F32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
        pipelined = args.precision == "bf16x3" and pipe.loop != "launches"
        whole_traffic = None
        if summary is not None and args.precision == "bf16x3":
            wp = summary.get("whole_pass", {})
            if "fetch_bytes_per_pass" in wp:
                whole_traffic = wp["fetch_bytes_per_pass"] + wp["write_bytes_per_pass"]
        whole = {"bound": "mfma", "achieved": round(ref_tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ref_tf / peak, 4),
                 "kernel": "whole pass (" + ("pipeline loop kernel" if pipelined else "hipGraph steps x50") + " + decode)",
                 "device_ms_per_pass": round(dev_ms / args.steps, 3),
                 "flops_per_motion_reference_equivalent": ref_flops_per_motion(),
                 "executed_tflops": round(exe_tf, 2),
                 "executed_frac": round(exe_tf * (1 if args.precision == "fp32" else 3) / peak, 4),
                 "traffic": whole_traffic,
                 "traffic_source": (summary["_path"] + f" (commit {summary.get('git_sha', '?')})") if whole_traffic is not None else None,
                 "mfma_util_pmc": summary.get("whole_pass", {}).get("mfma_util_pmc") if (summary and args.precision == "bf16x3") else None,
                 "peak_note": "fp32-input MFMA 157.3 TF/s" if args.precision == "fp32" else
                 "bf16 MFMA 2500 TF/s dense; every fp32-equivalent product costs 3 bf16 MFMAs (833 TF/s fp32-equivalent)"}
        line = {
            "metric": "motions/sec (196-frame, 50-step DDIM, bs128)", "value": round(motions_per_s, 2),
            "unit": "motions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "bf16x3+f32", "data": "synthetic",
            "config": {"workload": f"ddim50_cfg7.5_b{B}_f{FRAMES}_c{NFEATS}_humanml3d", "prompts_per_gpu": B,
                       "global_batch": total, "frames": FRAMES, "ddim_steps": STEPS_DDIM, "parallelism": f"dp{world}",
                       "loop": "persistent pipeline kernel (one launch for the 50 steps)" if pipelined else "hipGraph, 10 steps per replay"},
            "roofline": whole,
        }
        o_steps = max(2, args.steps // 2)
        o_tf = B * ref_flops_per_motion() / (o_dev_ms / 1e3 / o_steps) / 1e12
        o_peak = PEAK_F32_MFMA_TFLOPS if other == "fp32" else PEAK_BF16_MFMA_TFLOPS
        line["other_mode"] = {"precision": other, "value": round(total * o_steps / o_wall, 2), "unit": "motions/s",
                              "ms_per_step": round(o_wall / o_steps * 1e3, 3), "roofline_achieved_tflops": round(o_tf, 2),
                              "roofline_peak": o_peak, "roofline_frac": round(o_tf / o_peak, 4)}
        line["parity"] = {"max_abs_diff_frames_between_modes": mode_diff, "tolerance": 1e-3,
                          "note": "fp32 mode is within 1e-4 of the reference goldens, bf16x3 within 5e-4 (tests/test_gpu_path.py)"}
        if world == 1:
            # the contract's `roofline` describes the DOMINANT KERNEL (algorithmic FLOPs of one launch / its live HIP-event
            # duration); the whole-pass figures computed above move under roofline.whole_pass
            if pipelined:
                dk = pipeline_kernel_roofline(pipe, loop_ms, summary)
            else:
                d = dominant_kernel_roofline(dev, stream, args.precision, summary)
                dk = {"bound": "mfma", "achieved": d["achieved"], "peak": d["peak"], "unit": "TFLOP/s", "frac": d["frac"],
                      "traffic": d["traffic"], "kernel": d["name"], "us_per_launch": d["us_per_launch"],
                      "flops_per_launch": d["flops_per_launch"], "mfma_flops_per_launch": d["mfma_flops_per_launch"],
                      "mfma_frac": d["mfma_frac"], "mfma_util_pmc": d["mfma_util_pmc"], "share_of_pass": d["share_of_pass"],
                      "traffic_source": d["traffic_source"]}
            dk["peak_note"] = whole["peak_note"]
            dk["whole_pass"] = whole
            line["roofline"] = dk
        if world == 1 and args.cpu_sample > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample)
            line["cpu_baseline"]["gpu_over_cpu"] = round(motions_per_s / line["cpu_baseline"]["value"], 1)
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if use_dist:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

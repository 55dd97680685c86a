"""Per-kernel summary of rocprofv3 runs of scripts/profile_pass.py -> markdown table + profiles/rN/summary.json.

    python scripts/pmc_summary.py <run_dir> <passes> <out_md> <out_json> [git_sha]

<run_dir> holds one sub-directory per collection (each its own rocprofv3 run, as the guide prescribes):
    stats/                       --kernel-trace --stats               (un-countered durations)
    SQ_VALU_MFMA_BUSY_CYCLES/    --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace
    FETCH_SIZE/  WRITE_SIZE/     --pmc <counter> --kernel-trace
`passes` = sampling passes in each profiled run (profile_pass.py runs 1 warm-up + N).

MFMA utilisation.  Calibrated on bare MFMA loops of known count (scripts/ubench_mfma_calib.hip, profiles/r2/02_*):
SQ_VALU_MFMA_BUSY_CYCLES is summed over every SIMD of the chip and adds exactly 16 per v_mfma_f32_16x16x32_bf16 and 32
per v_mfma_f32_32x32x16_bf16, i.e. 1024 bf16 FLOP per counted cycle.  So
    util = BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz)
is the fraction of the 2.5 PFLOP/s dense bf16 peak the kernel's MFMAs fill, and it must equal the arithmetic figure
(MFMA FLOPs / duration / peak) at the same duration.  The r1 summaries divided by GRBM_GUI_ACTIVE / 8 instead; on the
calibration's 15 us dispatches that counter reads 1.26x the cycles the dispatch lasted (3.0 "GHz"), on 8 us ones more:
that, not the MFMA counter, made the r1 PMC figure read ~2x below the arithmetic one.
FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md §HBM); KiB units.
"""
import csv
import glob
import json
import sys
from collections import defaultdict

CLOCK_GHZ = 2.4          # the clock the 2.5 PFLOP/s peak is quoted at
N_SIMD = 1024


def short(name):
    return name.split("(")[0].replace("void ", "").replace("ladiff::", "")


MARKER = "init_latents_kernel"        # one per pass, in its prologue


def steady(rows, name_key):
    """(rows of the STEADY-STATE passes, their number): everything from the second pass's marker kernel to the last pass's - the
    warm-up pass (weight uploads, S-format splits, graph capture, the XCD probe) and the tail of the last pass are left out; the
    cut is a cyclic shift of whole passes, every kernel of a steady pass is counted exactly once per pass."""
    marks = sorted({int(r["Start_Timestamp"]) for r in rows if MARKER in r[name_key]})
    if len(marks) < 3:
        return rows, None
    lo, hi = marks[1], marks[-1]
    return [r for r in rows if lo <= int(r["Start_Timestamp"]) < hi], len(marks) - 2


def load_counter(root, name):
    files = glob.glob(f"{root}/{name}/**/*_counter_collection.csv", recursive=True)
    if not files:
        return None
    val = defaultdict(lambda: defaultdict(float)); n = defaultdict(int); dur = defaultdict(float)
    rows, _ = steady(list(csv.DictReader(open(files[0]))), "Kernel_Name")
    for r in rows:
        k = short(r["Kernel_Name"])
        val[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == name:
            n[k] += 1
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return val, n, dur


def load_trace(root):
    files = glob.glob(f"{root}/stats/**/*_kernel_trace.csv", recursive=True)
    n = defaultdict(int); dur = defaultdict(float)
    rows, passes = steady(list(csv.DictReader(open(files[0]))), "Kernel_Name")
    for r in rows:
        k = short(r["Kernel_Name"])
        n[k] += 1
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return n, dur, passes


def main():
    root, passes, out_md, out_json = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    sha = sys.argv[5] if len(sys.argv) > 5 else "unknown"
    tn, tdur, steady_passes = load_trace(root)
    runs = steady_passes if steady_passes else passes + 1          # steady-state passes only (fallback: every pass, warm-up included)
    mf = load_counter(root, "SQ_VALU_MFMA_BUSY_CYCLES")
    fe = load_counter(root, "FETCH_SIZE")
    wr = load_counter(root, "WRITE_SIZE")
    total_us = sum(tdur.values())
    kernels = {}
    for k in sorted(tn, key=lambda k: -tdur[k]):
        e = {"launches_per_pass": tn[k] / runs, "us_per_launch": tdur[k] / tn[k], "share_of_pass": tdur[k] / total_us}
        if mf and mf[1].get(k):
            v, n, dur = mf
            e["us_per_launch_pmc_run"] = dur[k] / n[k]
            e["mfma_busy_cycles_per_launch"] = v[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / n[k]
            e["mfma_util_pmc"] = v[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * dur[k] * 1e3 * CLOCK_GHZ)
            # the same busy cycles against the un-countered duration (counters serialise dispatches and stretch them)
            e["mfma_util_pmc_at_trace_duration"] = e["mfma_busy_cycles_per_launch"] / (N_SIMD * e["us_per_launch"] * 1e3 * CLOCK_GHZ)
        if fe and fe[1].get(k):
            e["fetch_bytes_per_launch"] = 2 * fe[0][k]["FETCH_SIZE"] * 1024 / fe[1][k]
        if wr and wr[1].get(k):
            e["write_bytes_per_launch"] = wr[0][k]["WRITE_SIZE"] * 1024 / wr[1][k]
        kernels[k] = e
    whole = {"us_per_pass": total_us / runs,
             "fetch_bytes_per_pass": sum(e.get("fetch_bytes_per_launch", 0) * e["launches_per_pass"] for e in kernels.values()),
             "write_bytes_per_pass": sum(e.get("write_bytes_per_launch", 0) * e["launches_per_pass"] for e in kernels.values()),
             "mfma_util_pmc": sum(e.get("mfma_busy_cycles_per_launch", 0) * e["launches_per_pass"] for e in kernels.values())
                              / (N_SIMD * total_us / runs * 1e3 * CLOCK_GHZ)}
    dominant = next(iter(kernels))
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    import bench
    summary = {"git_sha": sha, "csrc_hash": bench.csrc_hash(), "steady_state_passes": steady_passes, "command": "rocprofv3 ... -- python3 scripts/profile_pass.py f16x3 %d" % passes,
               "normalisation": "util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz); traffic = 2 x FETCH_SIZE + WRITE_SIZE",
               "dominant_kernel": dominant, "kernels": kernels, "whole_pass": whole}
    json.dump(summary, open(out_json, "w"), indent=1)
    with open(out_md, "w") as f:
        f.write(f"# PMC + kernel-trace summary (commit {sha}, csrc hash {summary['csrc_hash']}; {passes} passes + 1 warm-up per run; "
                f"counts and shares over the {steady_passes} steady-state passes between the second and the last prologue)\n\n")
        f.write("Separate rocprofv3 runs: `--kernel-trace --stats`, `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`, `--pmc FETCH_SIZE`, "
                "`--pmc WRITE_SIZE`, each `-- python3 scripts/profile_pass.py f16x3 N`.\n")
        f.write("MFMA util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz) - calibrated in 02_mfma_calibration.md; "
                "traffic = 2 x FETCH_SIZE + WRITE_SIZE at the L2 <-> fabric interface (Infinity-Cache hits included).\n\n")
        f.write("| kernel | launches/pass | us/launch (trace) | % of pass | us/launch (pmc run) | MFMA util (pmc run) | MFMA util (trace duration) | fetch MB/launch | write MB/launch |\n|---|---|---|---|---|---|---|---|---|\n")
        for k, e in kernels.items():
            if e["launches_per_pass"] < 1:
                continue
            f.write(f"| `{k[-70:]}` | {e['launches_per_pass']:.0f} | {e['us_per_launch']:.2f} | {100 * e['share_of_pass']:.1f} | "
                    f"{e.get('us_per_launch_pmc_run', 0):.2f} | {100 * e.get('mfma_util_pmc', 0):.1f} % | "
                    f"{100 * e.get('mfma_util_pmc_at_trace_duration', 0):.1f} % | {e.get('fetch_bytes_per_launch', 0) / 1e6:.2f} | "
                    f"{e.get('write_bytes_per_launch', 0) / 1e6:.2f} |\n")
        f.write(f"\nWhole pass: {whole['us_per_pass'] / 1e3:.2f} ms of kernels, fetch {whole['fetch_bytes_per_pass'] / 1e9:.2f} GB, "
                f"write {whole['write_bytes_per_pass'] / 1e9:.2f} GB, MFMA util {100 * whole['mfma_util_pmc']:.1f} % of the bf16 peak.\n")


if __name__ == "__main__":
    main()

"""Summarise rocprofv3 --pmc runs of bench.py (gpurun_out/pmc_r1/<counter>/...) per kernel: MFMA utilisation and
memory-side traffic.  FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md §HBM);
both counters are in KiB.  Usage: python scripts/pmc_summary.py gpurun_out/pmc_r1 > profiles/r1/03_pmc_summary.md"""
import csv, glob, sys
from collections import defaultdict
root = sys.argv[1]
PASSES = int(sys.argv[2]) if len(sys.argv) > 2 else 2   # passes in the profiled run
def load(name):
    f = glob.glob(f"{root}/{name}/*/*_counter_collection.csv")[0]
    d = defaultdict(lambda: defaultdict(float)); n = defaultdict(int); dur = defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        d[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == name:
            n[k] += 1; dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return d, n, dur
mf, n, dur = load("SQ_VALU_MFMA_BUSY_CYCLES")
fe, _, _ = load("FETCH_SIZE"); wr, _, _ = load("WRITE_SIZE")
print("# r1 / 03 - PMC counters of one benchmark pass (rocprofv3 --pmc, separate passes per counter group)\n")
print("`rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 1 --cpu-sample 0` (2 passes; numbers below are per pass).")
print("MFMA util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 4 SIMD x 256 CU) (the GRBM counter is summed over the 8 XCDs).")
print("Traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB at the L2 <-> fabric interface (Infinity-Cache hits included, so this is an upper bound on HBM bytes).\n")
print("| kernel | launches/pass | us/launch (profiled) | MFMA util | fetch MB/launch (x2 corrected) | write MB/launch |")
print("|---|---|---|---|---|---|")
tot_f = tot_w = 0.0
for k in sorted(n, key=lambda k: -dur[k]):
    if n[k] < 2: continue
    g = mf[k]["GRBM_GUI_ACTIVE"] / 8.0
    util = mf[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * 4 * 256) if g else 0.0
    f = 2 * fe[k]["FETCH_SIZE"] * 1024 / 1e6; w = wr[k]["WRITE_SIZE"] * 1024 / 1e6
    tot_f += f; tot_w += w
    print(f"| `{k[-60:]}` | {n[k] / PASSES:.0f} | {dur[k] / n[k]:.2f} | {100 * util:.1f} % | {f / n[k]:.2f} | {w / n[k]:.2f} |")
print(f"\nWhole pass: fetch {tot_f / PASSES / 1e3:.2f} GB (corrected), write {tot_w / PASSES / 1e3:.2f} GB; algorithmic minimum is ~0.12 GB of weights + ~1 GB of decoder activations,"
      " the rest is operand re-reads served by the Infinity Cache (every kernel boundary empties the L2s).")

"""Diagnostic (GPU box): timeline of one block through the persistent pipeline loop (s_memrealtime stamps, twin build;
the shipped library executes no stamp).  python scripts/stamps_pipeline.py [B] [steps]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
# the twin is built by `python -m ladiff_amd.build --stamps` (totals only) or with LADIFF_STAMPS_LEVEL=2 in the environment (per-block
# timeline too: each stamp costs ~0.1 us, so read INTERVALS from that build and busy / blocked TOTALS from the other)
subprocess.check_call([sys.executable, "-m", "ladiff_amd.build", "--stamps"], cwd=ROOT, stdout=subprocess.DEVNULL)
import torch
from ladiff_amd import _lib
from ladiff_amd import build as _build
_lib.LIB_PATH = _build.stamps_lib(os.environ.get("LADIFF_STAMPS_LEVEL", "1"))
import bench
from ladiff_amd import synthetic as syn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mode = sys.argv[3] if len(sys.argv) > 3 else "pipeline"
dev = torch.device("cuda", 0)
L = _lib.lib()
L.ladiff_debug_set_sys_stamps.argtypes = [ctypes.c_void_p]
L.ladiff_debug_set_probe(int(os.environ.get("PROBE", "0")))        # timing probes of the twin build (garbage results): scripts/ffn_probe.py
pipe = bench.build_pipe(dev, B)
pipe.precision = "f16x3"; pipe.num_inference_timesteps = steps; pipe.loop = mode
lens = [196] * B
text, noise = syn.text_embeddings(B).to(dev), syn.init_noise(lens).to(dev)
st = torch.zeros(256 * 4 * 4 * 8 + 256 * 4 + 256 * 4 * 8 + 256, dtype=torch.int64, device=dev)
s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s), torch.no_grad():
    pipe._diffusion_reverse(text, lens, init_noise=noise)
    torch.cuda.synchronize()
    L.ladiff_debug_set_sys_stamps(st.data_ptr())
    pipe._diffusion_reverse(text, lens, init_noise=noise)
    torch.cuda.synchronize()
print("status", pipe.loop_status())
stats = st[256 * 4 * 4 * 8:256 * 4 * 4 * 8 + 256 * 4].cpu().reshape(256, 4)
mid = st[256 * 4 * 4 * 8 + 256 * 4:256 * 4 * 4 * 8 + 256 * 4 + 256 * 4 * 8].cpu().reshape(256, 4, 8).double() * 0.01
ident = st[-256:].cpu().tolist()
t = st[:256 * 4 * 4 * 8].reshape(256, 4, 4, 8).cpu().double() * 0.01          # 100 MHz ticks -> us
names = []
for l in range(9):
    if l > 4: names += [f"L{l} SKIP{c}" for c in range(2)]
    names += [f"L{l} QKV{h}" for h in range(4)] + [f"L{l} OUT"] + [f"L{l} LIN{j}" for j in range(8)]
    # 16-row plan: RED2 x 2, STYL 2 groups (even / odd blocks) x 2 parts; 32-row plan: 3 + 3 (systolic.hip, red_plan)
    names += [f"L{l} RED2.{q}" for q in range(2 if mode == "pipeline16" else 3)] + [f"L{l} FFN{j}" for j in range(8)]
    names += ([f"L{l} STYL.{q}" for q in range(2)] + [f"L{l} STYLb.{q}" for q in range(2)]) if mode == "pipeline16" else [f"L{l} STYL.{q}" for q in range(3)]
names += [f"TAIL{k}" for k in range(4)]
# the stage table is permuted by the XCD placement: every workgroup wrote who it is; bring the rows back into chain order
def name_of(w):
    role, l, sl, b0 = (w >> 8) & 0xff, (w >> 16) & 0xff, (w >> 24) & 0xff, (w >> 32) & 0xff
    return {0: f"L{l} QKV{sl}", 1: f"L{l} OUT", 2: f"L{l} LIN{sl}", 3: f"L{l} RED2.{sl}", 4: f"L{l} FFN{sl}",
            5: f"L{l} STYL{'b' if b0 else ''}.{sl}", 6: f"L{l} SKIP{sl}", 7: f"TAIL{sl}"}[role]
where = {name_of(w): i for i, w in enumerate(ident) if w & 1}
assert all(n in where for n in names), [n for n in names if n not in where][:5]
order = [where[n] for n in names]
stats, mid, t = stats[order], mid[order], t[order]
xcd = [(ident[i] >> 48) & 0xff for i in order]
local = [(ident[i] >> 40) & 1 for i in order]
if any(x != 0xff for x in xcd):
    print("placement (XCD of each stage of layers 0-2, * = hands over through the XCD's L2): " +
          " ".join(f"{n.replace(' ', '.')}@{x}{'*' if lo else ''}" for n, x, lo in zip(names, xcd, local) if n[:2] in ("L0", "L1", "L2")))
    print(f"stages that store plainly: {sum(local)} of {len(local)}")
step, blk = min(1, steps - 1), 0
tail = len(names) - 4 + blk % 4
t0 = t[tail, step - 1, blk, 4] if step > 0 else t[:, step, blk, 1][t[:, step, blk, 1] > 0].min()
print(f"timeline of block {blk}, local step {step} (us after the TAIL stage finished the previous step's update of this block)")
print(f"{'stage':12s} {'flag seen':>10s} {'operands':>10s} {'mfma done':>10s} {'stored':>10s} {'published':>10s}   wait->publish")
for i, n in enumerate(names):
    r = t[i, step, blk]
    if r[1] == 0: continue
    if any(k in n for k in ("QKV0", "OUT", "LIN0", "RED2.0", "FFN0", "STYL.0", "SKIP0", "TAIL0")) or (len(sys.argv) > 4 and n.startswith(sys.argv[4])):
        f = lambda v: f"{v - t0:10.2f}" if v > 0 else f"{'-':>10s}"
        print(f"{n:12s} {f(r[1])} {f(r[2])} {f(r[3])} {f(r[4])} {f(r[5])}   {r[5] - r[1] if r[5] > 0 else r[4] - r[1]:6.2f}"
              "   since rows valid: " + " ".join(f"{j}:{r[j] - r[1]:5.2f}" for j in (2, 3, 6, 7, 4, 5) if r[j] > 0))
lap = t[tail, step, blk, 4] - t0
print(f"one step of one block: {lap:.1f} us over {59} hops = {lap / 59:.2f} us per hop")

# per-stage totals over the whole run: time blocked waiting for a producer, prefetch hit rate
print("stage-type totals (mean over the workgroups of a type): blocked us per step, prefetch hits / blocks")
import collections
agg = collections.defaultdict(list)
for i, n in enumerate(names):
    kind = n.split()[-1].rstrip("0123456789.")
    if stats[i, 2] > 0:
        agg[kind].append((float(stats[i, 0]) * 0.01 / steps, float(stats[i, 1]) / float(stats[i, 2])))
        if kind in ("QKV", "OUT"):          # wave-group loops: the other waves' time at the tile barrier (waiting for the loader waves)
            agg[kind + " compute waves at the tile barrier"].append((float(stats[i, 3]) * 0.01 / steps, 0.0))
for k, v in agg.items():
    print(f"  {k:6s} blocked {sum(a for a, _ in v) / len(v):8.1f} us/step (min {min(a for a, _ in v):8.1f})   prefetch hit rate {sum(h for _, h in v) / len(v):.2f}")

busy = sorted((float(stats[i, 0]) * 0.01 / steps, n, xcd[i], local[i]) for i, n in enumerate(names) if stats[i, 2] > 0)
print("least blocked workgroups (us/step blocked, stage, XCD, * = plain stores): " + "  ".join(f"{a:.1f} {n.replace(' ', '.')}@{x}{'*' if lo else ''}" for a, n, x, lo in busy[:16]))
print("steady state (mid-run, four consecutive blocks): us since the first block's loop top; stamps 0 top, 1 operands issued, 2 committed, 3 mfma, 6/7 (QKV: tile in LDS / scores; LIN, FFN: hidden slice written / past the mid barrier with the previous flag raised and the next loads issued), 4 stored, 5 published-or-deferred")
for i, n in enumerate(names):
    if n in ("L4 QKV0", "L4 OUT", "L4 LIN0", "L4 RED2.0", "L4 FFN0", "L4 STYL.0"):
        m = mid[i]
        if m[0, 0] == 0: continue
        t00 = m[0, 0]
        for k in range(4):
            order = [0, 1, 2, 3, 6, 7, 4, 5]
            print(f"  {n:10s} block+{k}: " + " ".join(f"{j}:{m[k, j] - t00:7.2f}" if m[k, j] > 0 else f"{j}:      -" for j in order))

# One block's way through three layers MID-RUN (step n/2, block NB/2), absolute times: how long it sat in front of each stage
# (rows stored by the producer -> this stage has them = transit + queueing) and how long the stage took
print("mid-run, ONE block through layers 3-5 (us since L3.QKV0 had its rows): rows valid | stored | in stage | since the producer stored")
seq = []
for l in (3, 4, 5):
    seq += ([f"L{l} SKIP0"] if l > 4 else []) + [f"L{l} QKV0", f"L{l} OUT", f"L{l} LIN0", f"L{l} RED2.0", f"L{l} FFN0", f"L{l} STYL.0", f"L{l} STYLb.0"]
idx = {n: i for i, n in enumerate(names)}
T0, prev_done = None, None
for n in seq:
    if n not in idx: continue
    m = mid[idx[n]][0]
    if m[1] == 0: continue                                     # (a STYL group that does not visit this block)
    valid = m[1]
    done = max(m[4], m[5])
    if T0 is None: T0 = valid
    print(f"  {n:10s} {valid - T0:8.2f} | {done - T0:8.2f} | {done - valid:6.2f} | " + (f"{valid - prev_done:6.2f}" if prev_done is not None else "     -"))
    prev_done = done

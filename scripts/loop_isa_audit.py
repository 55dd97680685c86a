"""Per-loop audit of the persistent pipeline kernel's ISA (VERDICT r5 next #1a): for one instantiation of systolic_loop_kernel, every loop
(a backward branch) of the gfx950 disassembly with what sits on its body's serial chain - MFMAs, `s_waitcnt vmcnt / lgkmcnt` by count,
barriers, `v_readlane` / `v_writelane` (SGPR spills to lanes), scratch loads / stores, `s_nop`, DPP, LDS and buffer operations, `s_sleep`
(poll loops).  The role a block loop belongs to is recognised by what it computes (see `guess`).
usage: loop_isa_audit.py [kernel-name substring, default the headline instantiation] [object stem, default systolic]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
key = sys.argv[1] if len(sys.argv) > 1 else "systolic_loop_kernelILi1ELi0ELi2ELi1ELb1E"
stem = sys.argv[2] if len(sys.argv) > 2 else "systolic"
builddir = os.environ.get("LADIFF_OBJDIR", os.path.join(ROOT, "ladiff_amd", "csrc", "build"))
with tempfile.TemporaryDirectory() as tmp:
    os.symlink(os.path.join(builddir, stem + ".o"), os.path.join(tmp, "o.o"))
    subprocess.run([OBJDUMP, "--offloading", "o.o"], cwd=tmp, check=True, capture_output=True)
    dev = [f for f in os.listdir(tmp) if "gfx950" in f][0]
    out = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, dev)], check=True, capture_output=True, text=True).stdout
ins, name = [], None
for line in out.splitlines():
    m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
    if m:
        name = m.group(2); continue
    if name is None or key not in name or not line.startswith("\t"):
        continue
    m = re.match(r"^\t(.*?)\s*//\s*([0-9A-F]+):", line)
    if m:
        ins.append((int(m.group(2), 16), m.group(1).strip()))
if not ins:
    sys.exit(f"no kernel matching {key}")
addr_to_idx = {a: i for i, (a, _) in enumerate(ins)}
print(f"{key}: {len(ins)} instructions")
loops = []
for i, (a, t) in enumerate(ins):
    m = re.match(r"s_cbranch_\w+ (\d+)", t) or re.match(r"s_branch (\d+)", t)
    if m:
        off = int(m.group(1)); off = off - 65536 if off >= 32768 else off
        tgt = a + 4 + 4 * off
        if off < 0 and tgt in addr_to_idx:
            loops.append((addr_to_idx[tgt], i))
loops.sort()
def stats(lo, hi):
    body = [t for _, t in ins[lo:hi + 1]]
    c = lambda pat: sum(1 for t in body if re.search(pat, t))
    vm = {}
    for t in body:
        if t.startswith("s_waitcnt"):
            for kind in ("vmcnt", "lgkmcnt"):
                m = re.search(kind + r"\((\d+)\)", t)
                if m: vm.setdefault(kind, {}).setdefault(int(m.group(1)), 0); vm[kind][int(m.group(1))] += 1
    return dict(n=len(body), mfma=c(r"^v_mfma"), barrier=c(r"^s_barrier"), readlane=c(r"^v_readlane"), writelane=c(r"^v_writelane"),
                scratch_ld=c(r"^scratch_load"), scratch_st=c(r"^scratch_store"), nop=c(r"^s_nop"), dpp=c(r"_dpp|row_shr|row_bcast|quad_perm"),
                ds_rd=c(r"^ds_read"), ds_wr=c(r"^ds_write"), buf_ld=c(r"^buffer_load"), buf_st=c(r"^buffer_store"), glob_ld=c(r"^global_load"),
                sleep=c(r"^s_sleep"), exp=c(r"^v_exp_f32"), rcp=c(r"^v_rcp_f32"), rsq=c(r"^v_rsq_f32"), cvt=c(r"^v_cvt_pk"), smem=c(r"^s_load|^s_buffer_load"),
                memtime=c(r"s_memrealtime|s_memtime"), wait=vm)
def guess(s):
    if s["mfma"] == 0: return "poll / row loop" if s["sleep"] else "-"
    if s["mfma"] == 72 or (s["exp"] and s["dpp"] > 40 and s["rsq"] == 0): return "QKV (wave-group loop: loader chain, projection, 7-key attention)"
    if s["rsq"] and s["exp"] == 0 and s["dpp"] >= 32: return "OUT (wave-group loop: loader chain, out-projection, residual + LayerNorm epilogue)"
    if s["exp"] and s["rsq"]: return "STYL (eight partial planes, LayerNorm, AdaLN, SiLU, 256x256 product)"
    if s["exp"] and s["rcp"]: return "FFN (two products, GELU)"
    if s["buf_ld"] >= 8 and s["buf_st"] <= 1: return "SKIP (K = 512 product of two row images)"
    return "LIN (two products, ReLU)"
outer = [(lo, hi) for lo, hi in loops if stats(lo, hi)["mfma"] > 0]
# keep the innermost MFMA-bearing loops that are not contained in a smaller MFMA loop (block loops); print nested poll loops beneath
for lo, hi in outer:
    s = stats(lo, hi)
    inner = [(l2, h2) for l2, h2 in loops if lo <= l2 and h2 <= hi and (l2, h2) != (lo, hi)]
    if any(stats(l2, h2)["mfma"] > 0 for l2, h2 in inner):
        print(f"\n[{lo}:{hi}] outer loop around MFMA loops ({s['n']} instructions) - step loop"); continue
    w = s["wait"]
    print(f"\n[{lo}:{hi}] block loop, {s['n']} instructions: {guess(s)}")
    print(f"    mfma {s['mfma']}  barriers {s['barrier']}  ds_read {s['ds_rd']} ds_write {s['ds_wr']}  buffer_load {s['buf_ld']} buffer_store {s['buf_st']} global_load {s['glob_ld']}  s_load {s['smem']}")
    print(f"    v_readlane {s['readlane']} v_writelane {s['writelane']}  scratch_load {s['scratch_ld']} scratch_store {s['scratch_st']}  s_nop {s['nop']}  dpp {s['dpp']}  exp {s['exp']} rcp {s['rcp']} rsq {s['rsq']} cvt_pk {s['cvt']}  s_memrealtime {s['memtime']}")
    print(f"    s_waitcnt vmcnt: {dict(sorted(w.get('vmcnt', {}).items()))}   lgkmcnt: {dict(sorted(w.get('lgkmcnt', {}).items()))}")
    for l2, h2 in inner:
        s2 = stats(l2, h2)
        print(f"      inner [{l2}:{h2}] {s2['n']} instructions: sleep {s2['sleep']} buffer_load {s2['buf_ld']} vmcnt {dict(sorted(s2['wait'].get('vmcnt', {}).items()))} readlane {s2['readlane']} scratch {s2['scratch_ld'] + s2['scratch_st']} memrealtime {s2['memtime']}")

# optional: LADIFF_AUDIT_DUMP=lo:hi prints the skeleton of that instruction range (runs of plain VALU / MFMA instructions are counted)
rng = os.environ.get("LADIFF_AUDIT_DUMP")
if rng:
    lo, hi = map(int, rng.split(":"))
    run = {}
    def flush():
        if run:
            print("        ... " + ", ".join(f"{v} x {k}" for k, v in run.items())); run.clear()
    for i in range(lo, hi + 1):
        t = ins[i][1]
        op = t.split()[0]
        keyop = re.match(r"^(s_waitcnt|s_barrier|s_cbranch|s_branch|s_sleep|buffer_|global_|scratch_|ds_|s_load|s_buffer|s_memrealtime|s_setprio|s_nop|v_readlane|v_readfirstlane|v_writelane|s_and_saveexec|s_or_b64 exec|s_mov_b64 exec|s_endpgm)", t)
        if keyop and not op.startswith("s_nop") and not op.startswith("v_readlane"):
            flush(); print(f"  {i:6d}  {t[:110]}")
        else:
            k = "v_mfma" if op.startswith("v_mfma") else ("v_readlane" if op.startswith("v_readlane") else ("s_nop" if op.startswith("s_nop") else ("valu" if op.startswith("v_") else "salu")))
            run[k] = run.get(k, 0) + 1
    flush()

"""Builds ladiff_amd/libladiff_hip.so (gfx950) in-tree with hipcc.  `python -m ladiff_amd.build`"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libladiff_hip.so")
SOURCES = ["gemm.hip", "gemm_big.hip", "gemm_kr.hip", "gemm_rowln.hip", "rowops.hip", "attention.hip", "qkv_attn.hip", "systolic.hip", "linear_ca.hip", "dec_cross.hip", "dec_mlp.hip", "dec_qkv_attn.hip", "feats2joints.hip", "denoiser.hip", "decoder.hip", "encoder.hip", "clip.hip", "evaluator.hip", "api.hip"]
# -fvisibility=hidden: the library exports the C ABI of include/ladiff_hip.h (+ ladiff_hip_debug.h) and nothing else (LADIFF_API)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def stamps_lib(level=1):
    """Path of the diagnostic twin of a stamps level (1: per-workgroup totals only, 2: + per-block timeline)."""
    return LIB.replace(".so", "_stamps.so" if int(level) == 1 else f"_stamps{int(level)}.so")


def diag_lib():
    """Path of the hand-off diagnostic build of the loop kernel (build_all): tags count 16 generations instead of 2 and every look a
    one-bit tag would have accepted from another generation is recorded (csrc/systolic.hip, LADIFF_SELFCHECK); never the product."""
    return LIB.replace(".so", "_diag.so")


def build(force=False, verbose=False, stamps=False, tag=None, defines=(), only=None):
    """stamps=True builds the diagnostic twin (in-kernel s_memrealtime stamps and timing probes; never timed, never shipped as the
    product): level 1 -> libladiff_hip_stamps.so, LADIFF_STAMPS_LEVEL=2 in the environment -> libladiff_hip_stamps2.so (each level has
    its own object directory AND its own library, so one never serves the other)."""
    obj, lib, flags = OBJ, LIB, list(FLAGS)
    if tag:                                                  # a whole-library variant for a same-box A/B (never the product): own objects, own .so
        obj, lib = OBJ + "_" + tag, LIB.replace(".so", f"_{tag}.so")
        flags += ["-D" + d for d in defines]
    if stamps:
        # level 1: per-workgroup totals only (blocked / busy time: undistorted); level 2 adds the per-block timeline stamps (~0.1 us
        # each: read intervals from it, not totals)
        lvl = os.environ.get("LADIFF_STAMPS_LEVEL", "1")
        obj, lib = OBJ + "_stamps" + (lvl if lvl != "1" else ""), stamps_lib(lvl)
        flags.append("-DLADIFF_STAMPS=" + lvl)
    os.makedirs(obj, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(os.path.dirname(HERE), "include", h) for h in ("ladiff_hip.h", "ladiff_hip_debug.h")]
    hipcc = _hipcc()
    jobs = []
    # only = the sources the variant's defines reach: every other object is the product's (built first by the caller)
    objdir = {src: (obj if only is None or src in only else OBJ) for src in SOURCES}
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir[src], src.replace(".hip", ".o"))
        if objdir[src] == obj and (force or _stale(o, [s] + headers)):
            jobs.append([hipcc, *flags, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for err in ex.map(run, jobs):
            if verbose and err.strip():
                print(err)
    objs = [os.path.join(objdir[s], s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(lib, objs + [os.path.join(CSRC, "exports.map")]):
        # the version script keeps what -fvisibility=hidden cannot reach (libstdc++ template instantiations, kernel handle objects) local
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"),
             "-o", lib, *objs])
    return lib


def build_all(force=False, verbose=False):
    """Both flavours of the product: libladiff_hip.so (split operands as fp16 pairs) and libladiff_hip_bf16.so (bf16 pairs,
    _lib.select_split_format("bf16")) - and the loop kernel's hand-off diagnostic build (diag_lib(): test infrastructure)."""
    lib = build(force=force, verbose=verbose)
    build(force=force, verbose=verbose, tag="bf16", defines=["LADIFF_SPLIT_BF16"])
    # the hand-off diagnostic build of the loop kernel (tests/test_gpu_pipeline.py runs it over the block geometries; diag_lib())
    build(force=force, verbose=verbose, tag="diag", defines=["LADIFF_TAG_BITS=4", "LADIFF_SELFCHECK"], only=("systolic.hip",))
    return lib


if __name__ == "__main__":
    _tag = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--tag=")), None)
    _defs = [a[2:] for a in sys.argv if a.startswith("-D")]
    if "--all" in sys.argv:
        print(build_all(force="--force" in sys.argv, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True, stamps="--stamps" in sys.argv, tag=_tag, defines=_defs))

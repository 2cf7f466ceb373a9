"""Builds libvfmseg_hip.so and its fp16 twin libvfmseg_hip_f16.so (gfx950) in-tree with hipcc.
`python -m vfmseg_amd.csrc.build [--force] [--only bf16|f16]`.  The twin is the same sources with -DVFM_HALF_F16 (common.h)."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libvfmseg_hip.so")
LIB_F16 = os.path.join(HERE, "libvfmseg_hip_f16.so")
OBJ = os.path.join(HERE, "build")
OBJ_F16 = os.path.join(HERE, "build_f16")
# bf16-only kernels that have no place in the fp16 twin (split-bf16 precision mode; the persistent GEMM experiment)
BF16_ONLY = set()
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-Wno-unused-result",
         "-I" + os.path.join(ROOT, "include")]


# per-file extras.  attention_fwd64.hip places every VALU instruction by hand between MFMAs: SLP-packed f32 adds (v_pk_add_f32)
# cost more issue cycles there than the two scalar adds they replace
EXTRA = {"attention_fwd64.hip": ["-fno-slp-vectorize"],
         # gemm_ps.hip runs its epilogue arithmetic between the MFMAs of the next sub-tile: packed f32 VALU is an anti-lever there
         "gemm_ps.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.exists(c) or c == "hipcc":
            return c


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# experiments that are not part of the shipped library: built only with VFMSEG_EXPERIMENTAL=1 (the library then exports the same C ABI;
# the dispatchers reach them through vfm_tune knobs).  attention_fwd64.hip: a 64-queries-per-wave hand-pipelined forward, 7-9 % faster
# back to back and no faster inside the train step.
# gemm_ps.hip (round 3: persistent, two accumulator sets per wave) and gemm_v5.hip (round 4: the vendor kernel's structure - 4 waves, 16x16x32
# MFMA, two whole K-tile stages, buffer-form LDS-DMA) are the vehicles of the GEMM experiments of DESIGN.md section 5.1: measured, no faster than
# the ring kernels of gemm_w4.hip, not part of the shipped library.
EXPERIMENTAL = {"attention_fwd64.hip": "VFM_EXPERIMENTAL_FWD64", "gemm_ps.hip": "VFM_EXPERIMENTAL_GEMM", "gemm_v5.hip": "VFM_EXPERIMENTAL_GEMM"}


def build(force=False, verbose=True, only=None):
    """Both libraries (or `only` = "bf16" / "f16"); returns the path of the bf16 one."""
    if only in (None, "bf16"):
        _build_one(LIB, OBJ, [], force, verbose, exp=os.environ.get("VFMSEG_EXPERIMENTAL", "0") == "1")
    if only in (None, "f16"):
        _build_one(LIB_F16, OBJ_F16, ["-DVFM_HALF_F16"], force, verbose, exp=False)
    return LIB


def _build_one(LIB, OBJ, variant_defs, force, verbose, exp):
    srcs = sorted(s for s in glob.glob(os.path.join(HERE, "*.hip")) if exp or os.path.basename(s) not in EXPERIMENTAL)
    defs = (["-D" + d for d in sorted(set(EXPERIMENTAL.values()))] if exp else []) + variant_defs
    hdrs = sorted(glob.glob(os.path.join(HERE, "*.h"))) + [os.path.join(ROOT, "include", "vfmseg_hip.h")]
    os.makedirs(OBJ, exist_ok=True)
    # the objects of a directory were compiled with ONE set of flags: a build with other flags (VFMSEG_EXPERIMENTAL on / off) starts over
    stamp, flags = os.path.join(OBJ, ".flags"), " ".join(FLAGS + defs)
    if not os.path.exists(stamp) or open(stamp).read() != flags:
        force = True
        for o in glob.glob(os.path.join(OBJ, "*.o")):
            os.remove(o)
        with open(stamp, "w") as f:
            f.write(flags)
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [_hipcc()] + FLAGS + defs + EXTRA.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (s, r.stderr[-6000:]))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr[-2000:])
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    if jobs or force or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    print(build(force="--force" in sys.argv, only=only))

"""Builds libdimo_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m dimo_amd.csrc.build [--force]

Per-file flags: preprocess.hip is compiled with -ffp-contract=off (bit-exact tile rects / depth
keys versus the CPU oracle); every other file keeps the default (fused multiply-add) contraction.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libdimo_hip.so")
ARCH = "gfx950"
SOURCES = {
    "api.hip": [],
    # (no SLP packing: the projection backward needs 90 VGPRs instead of 110 -- five waves per SIMD -- and runs 67.9
    # against 71.5 us per 8 renders alone: profiles/r05_kernel_stats_serial_8renders.txt vs gpurun call r5e)
    "preprocess.hip": ["-ffp-contract=off", "-fno-slp-vectorize"],
    "binning.hip": [],
    # the SLP vectoriser turns the per-pixel maths into v_pk_*_f32 + v_mov shuffles: a packed op issues in the time of
    # its two scalar halves on gfx950 (profiles/r02_valu_issue_rates.txt), so the packing only adds the moves
    # (150-187 -> 116 issue cycles per quadrant visit of the backward)
    "blend.hip": ["-fno-slp-vectorize"],
    "knn.hip": ["-ffp-contract=off"],
    "ssim.hip": ["-fno-slp-vectorize"],  # packed FMAs need register pairs: with the window in VGPRs they spilled
    "deform.hip": [],
    "image_loss.hip": [],
    "adam.hip": [],
    "timenet.hip": [],
    "fps.hip": ["-ffp-contract=off"],
    "executor.hip": [],
}
if os.environ.get("DIMO_BWD_TRACE") == "1":  # per-item trace of the blend backward (tools/bwd_trace.py)
    SOURCES["blend.hip"] = SOURCES["blend.hip"] + ["-DDIMO_BWD_TRACE"]
if os.environ.get("DIMO_BWD_WAVES"):  # waves per SIMD the blend backward is compiled for (experiments; default 4)
    SOURCES["blend.hip"] = SOURCES["blend.hip"] + ["-DDIMO_BWD_WAVES=%d" % int(os.environ["DIMO_BWD_WAVES"])]
if os.environ.get("DIMO_BIN_TRACE") == "1":  # per-workgroup phase trace of the binning kernels (tools/bin_trace.py)
    SOURCES["binning.hip"] = SOURCES["binning.hip"] + ["-DDIMO_BIN_TRACE"]
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-fast-math", "-Wall", "-Wno-unused-function"]
# A/B builds (tools/ab.sh "@name" modes): DIMO_BUILD_VARIANT=name [DIMO_BUILD_EXTRA="flags for every file"] builds
# csrc/variants/name.so from the same sources into its own object directory; the default library is untouched
VARIANT = os.environ.get("DIMO_BUILD_VARIANT")
if VARIANT:
    LIB = os.path.join(HERE, "variants", VARIANT + ".so")
    COMMON = COMMON + os.environ.get("DIMO_BUILD_EXTRA", "").split()


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    # every header of this directory is a dependency of every object (a glob, so that a new one cannot be forgotten:
    # loss_terms.hpp was missing from the hand-written list in round 5)
    headers = sorted(os.path.join(HERE, h) for h in os.listdir(HERE) if h.endswith(".hpp")) + \
              [os.path.join(HERE, "..", "..", "include", "dimo_hip.h"), __file__]
    objdir = os.path.join(HERE, "build" + ("_" + VARIANT if VARIANT else ""))
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    jobs = []
    for src, extra in SOURCES.items():
        path = os.path.join(HERE, src)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: listed in SOURCES but missing -- the library would link without it")
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        jobs.append((path, obj, extra))

    def compile_one(job):
        path, obj, extra = job
        if force or _stale(obj, [path] + headers):
            cmd = [hipcc, "-c", path, "-o", obj] + COMMON + extra
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            return True
        return False

    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        rebuilt = list(ex.map(compile_one, jobs))
    objs = [j[1] for j in jobs]
    if force or any(rebuilt) or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

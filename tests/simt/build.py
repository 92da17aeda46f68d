"""Host build of the binning kernels on the SIMT emulation shim (test infrastructure; see shim/hip/hip_runtime.h).

    python -m tests.simt.build

Reads dimo_amd/csrc/binning.hip + common.hpp AS THEY ARE, substitutes the gfx950 inline-asm statements (the LDS-only
barrier and v_writelane_b32) by their emulation calls, and compiles with g++ into tests/simt/_build/libbinning_emu.so.
The .hip sources carry no host / emulation switches of their own.
"""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
CSRC = os.path.join(ROOT, "dimo_amd", "csrc")
OUT = os.environ.get("SIMT_BUILD_DIR") or os.path.join(HERE, "_build")  # (the sanitizer builds go to /tmp: tools/emulated_asan.sh)

def _place_tile_asm():
    text = open(os.path.join(CSRC, "binning.hip")).read()
    i = text.index('asm volatile("v_bfe_u32 %[t], %[m], %[jb], 1')
    j = text.index('"memory");', i) + len('"memory");')
    return text[i:j]


PLACE_TILE_ASM = _place_tile_asm()
SUBST = {
    "common.hpp": [
        ('asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier" ::: "memory");', "__syncthreads();"),
    ],
    "binning.hip": [
        ('asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(sval), "n"(J));', "v = simt_writelane(v, sval, J);"),
        # place_tile: the whole statement, up to its closing parenthesis
        (PLACE_TILE_ASM, "(void)t, (void)save, (void)first; simt_place_tile(mhalf, JB, J, c, id, vals);"),
    ],
}


def _transformed(name, extra=()):
    text = open(os.path.join(CSRC, name)).read()
    # SIMT_MUTANT="<file>|<k>|<old>|<new>": the k-th occurrence of <old> in the RAW <file> becomes <new>
    # (tools/mutate_emulated.py: how many deliberately wrong kernels the emulated tests catch)
    mutant = os.environ.get("SIMT_MUTANT")
    if mutant:
        mfile, k, mold, mnew = mutant.split("|")
        if mfile == name:
            at = -1
            for _ in range(int(k) + 1):
                at = text.index(mold, at + 1)
            text = text[:at] + mnew + text[at + len(mold):]
    for old, new in list(SUBST[name]) + list(extra):
        if mutant and text.count(old) < 1:
            raise SystemExit("SIMT_MUTANT lands inside a substituted statement")
        assert text.count(old) >= 1, (name, old)
        text = text.replace(old, new)
    # kernel<<<grid, block, lds, stream>>>(args) -> hipLaunchKernelGGL((kernel), grid, block, lds, stream, args)
    text = re.sub(r"\b([A-Za-z_]\w*(?:<[^<>;]*>)?)<<<(.+?)>>>\(", r"hipLaunchKernelGGL((\1), \2, ", text)
    assert "asm volatile(" not in text and "asm(" not in text, name + ": an inline-asm statement without a substitution"
    return text


# emulated libraries: name -> (.hip sources of dimo_amd/csrc it compiles, its driver)
TARGETS = {
    "binning": (["binning.hip"], "binning_emu.cpp"),
    "points": (["knn.hip", "fps.hip"], "points_emu.cpp"),  # KNN / distCUDA2 / farthest point sampling
    "project": (["preprocess.hip"], "project_emu.cpp"),    # the projection kernel (forward and backward)
    "deform": (["deform.hip", "adam.hip"], "deform_emu.cpp"),  # skinning forward / backward, the flat Adam step
    # the rasterizer's whole C ABI: projection, binning, blend forward / backward, projection backward -- one
    # translation unit per source, as on the GPU (SEPARATE below)
    "raster": (["preprocess.hip", "binning.hip", "blend.hip"], "raster_emu.cpp"),
}
# the image losses: SSIM (forward, backward, fused), the other image terms, SSIM + every term in one tile pass
TARGETS["losses"] = (["ssim.hip", "image_loss.hip"], "losses_emu.cpp")
# the TimeNet: forward, dgrad chain, weight gradients (fp32 MFMA: the builtins are wave rendezvous in the shim)
TARGETS["timenet"] = (["timenet.hip"], "timenet_emu.cpp")
# the native step executor over the batched forms of the kernels (streams and events are no-ops: a launch has run
# when hipLaunchKernelGGL returns)
TARGETS["step"] = (["executor.hip", "deform.hip", "preprocess.hip", "binning.hip", "blend.hip"], "step_emu.cpp")
SEPARATE = {"raster", "losses", "timenet", "step"}
SUBST.setdefault("executor.hip", [])
# The step executor with ONE cross-stream dependency taken out, each a library of its own: what the deferred stream
# orders of the emulation (runtime.cpp) must catch -- tests/test_executor_emulated.py expects these to FAIL.
_STEP = TARGETS["step"]
BROKEN = {
    # the fold (caller's stream) no longer waits for the ranges' rasterizer / skinning backward on their streams
    "step_no_accumulate_wait": {"executor.hip": [
        ("if (wait_done(ex, main, si, ex->render_done[i], ex->render_val, i)) return DIMO_E_LAUNCH;", "(void)si;")]},
    # the joint backward (caller's stream) no longer waits for the ranges' forwards on their streams
    "step_no_joint_wait": {"executor.hip": [
        ("if (mark_done(ex, si, ex->fwd_done[i], ex->fwd_val, i) || wait_done(ex, main, si, ex->fwd_done[i], ex->fwd_val, i))",
         "if (mark_done(ex, si, ex->fwd_done[i], ex->fwd_val, i))")]},
    # a range's backward on its private stream no longer waits for the caller's stream (dimo_executor_backward_launch)
    "step_no_backward_fork": {"executor.hip": [
        ("    int rc = fork_one(ex, main, s);\n    if (!rc) rc = batched_backward_raster(c, d, first, count, s);",
         "    int rc = batched_backward_raster(c, d, first, count, s);")]},
}
for _name in BROKEN:
    TARGETS[_name] = _STEP
    SEPARATE.add(_name)
SUBST["timenet.hip"] = [
    # clang's vector extension -> GCC's
    ("typedef float f32x4 __attribute__((ext_vector_type(4)));", "typedef float f32x4 __attribute__((vector_size(16)));"),
    ("typedef float f32x16 __attribute__((ext_vector_type(16)));", "typedef float f32x16 __attribute__((vector_size(64)));"),
    ('asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier" ::: "memory");', "__syncthreads();"),
]
SUBST["ssim.hip"] = [
    # the window's taps moved from SGPRs into VGPRs
    ('asm volatile("v_mov_b32 %0, %1" : "=v"(v.w[k]) : "s"(win.w[k]));', "v.w[k] = win.w[k];"),
]
SUBST.setdefault("image_loss.hip", [])
SUBST.setdefault("loss_terms.hpp", [])
SUBST["blend.hip"] = [
    # (a register-allocation hint: an empty asm statement with a VGPR constraint)
    ('if (k < (NORMAL ? 13 : 10)) asm volatile("" : "+v"(v[k]));', ""),
    # the trace's hardware-id reads (trace builds only; the branch is dead here)
    ('asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));', "hw = 0;"),
    ('asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));', "xcc = 0;"),
]
SUBST.setdefault("knn.hip", [])
SUBST.setdefault("fps.hip", [])
SUBST.setdefault("preprocess.hip", [])
SUBST.setdefault("proj_math.hpp", [])
SUBST["deform.hip"] = [  # extern __shared__ arrays -> the workgroup's dynamic LDS of the emulation
    ("extern __shared__ __attribute__((aligned(16))) float s_cp[];", "HIP_DYNAMIC_SHARED(float, s_cp)"),
    ("extern __shared__ __attribute__((aligned(16))) float smem[];", "HIP_DYNAMIC_SHARED(float, smem)"),
]
SUBST.setdefault("adam.hip", [])
SUBST.setdefault("deform_body.hpp", [])
# wave_reduce16's asm block (one per value count) -> the shim's instruction-for-instruction spelling of it
SUBST["wave_ops.hpp"] = [
    ('asm volatile("s_nop 1\\n\\t" DIMO_RA(0, 8) DIMO_RA(1, 9) DIMO_RA(2, 10) DIMO_RA(3, 11) DIMO_RA(4, 12) DIMO_RA(5, 13)\n'
     '                 DIMO_RA(6, 14) DIMO_RA(7, 15) DIMO_RTAIL DIMO_ROPS);', "simt_wave_reduce16<16>(v);"),
    ('asm volatile("s_nop 1\\n\\t" DIMO_RA(0, 8) DIMO_RA(1, 9) DIMO_RA(2, 10) DIMO_RA(3, 11) DIMO_RA(4, 12) DIMO_RA1(5)\n'
     '                 DIMO_RA1(6) DIMO_RA1(7) DIMO_RTAIL DIMO_ROPS);', "simt_wave_reduce16<13>(v);"),
    ('asm volatile("s_nop 1\\n\\t" DIMO_RA(0, 8) DIMO_RA(1, 9) DIMO_RA1(2) DIMO_RA1(3) DIMO_RA1(4) DIMO_RA1(5) DIMO_RA1(6)\n'
     '                 DIMO_RA1(7) DIMO_RTAIL DIMO_ROPS);', "simt_wave_reduce16<10>(v);"),
]
HEADERS = ["common.hpp", "proj_math.hpp", "wave_ops.hpp", "deform_body.hpp", "loss_terms.hpp"]  # copied beside the sources (substitutions applied)


def build(force=False, target="binning"):
    hips, driver = TARGETS[target]
    # SIMT_ASAN=1: the same libraries under AddressSanitizer (run the tests with LD_PRELOAD=$(gcc -print-file-name=libasan.so)
    # ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0): numpy's buffers are malloc'ed, so a kernel's read or
    # write past the end of a workspace -- forgiven on a GPU while the page is mapped -- is reported with its source line
    asan = bool(os.environ.get("SIMT_ASAN"))
    # SIMT_UBSAN=1: ... under UndefinedBehaviorSanitizer (shifts by the operand's width or more, misaligned vector loads,
    # signed overflow, out-of-range float -> int conversions: each hardware-defined on the GPU, none of them meant)
    ubsan = bool(os.environ.get("SIMT_UBSAN"))
    lib = os.path.join(OUT, "lib%s_emu%s%s.so" % (target, "_asan" if asan else "", "_ubsan" if ubsan else ""))
    srcs = [os.path.join(CSRC, n) for n in HEADERS + hips] + [os.path.join(HERE, n) for n in
            ("runtime.cpp", driver, "build.py", os.path.join("shim", "hip", "hip_runtime.h"))] + \
           [os.path.join(ROOT, "include", "dimo_hip.h")]
    if not force and os.path.exists(lib) and all(os.path.getmtime(s) <= os.path.getmtime(lib) for s in srcs):
        return lib
    # the transformed sources keep their relative include paths: _build/src/dimo_amd/csrc/{common.hpp, <name>_src.inc}
    d = os.path.join(OUT, "src", "dimo_amd", "csrc")
    os.makedirs(d, exist_ok=True)
    os.makedirs(os.path.join(OUT, "src", "include"), exist_ok=True)
    extra = BROKEN.get(target, {})
    for h in HEADERS:
        open(os.path.join(d, h), "w").write(_transformed(h))
    for h in hips:
        open(os.path.join(d, h.replace(".hip", "_src.inc")), "w").write(_transformed(h, extra.get(h, ())))
    open(os.path.join(OUT, "src", "include", "dimo_hip.h"), "w").write(open(os.path.join(ROOT, "include", "dimo_hip.h")).read())
    # (-ffp-contract=off: knn.hip and fps.hip are built that way for the GPU too -- dimo_amd/csrc/build.py -- so that their
    # distances are bit for bit the oracle's)
    units = []
    if target in SEPARATE:
        for h in hips:
            units.append(os.path.join(d, h.replace(".hip", "_tu.cpp")))
            open(units[-1], "w").write('#include "%s"\n' % h.replace(".hip", "_src.inc"))
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", "-fno-omit-frame-pointer", "-ffp-contract=off",
           "-I", os.path.join(HERE, "shim"), "-I", d, "-Wno-unused-function", "-Wno-psabi",
           os.path.join(HERE, "runtime.cpp"), os.path.join(HERE, driver)] + units + ["-o", lib]
    if asan:
        cmd[1:1] = ["-fsanitize=address"]
    if ubsan:
        cmd[1:1] = ["-fsanitize=undefined,float-cast-overflow", "-fno-sanitize=vptr"]
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    for t in [t for t in TARGETS if t not in BROKEN]:
        print(build(force=True, target=t))

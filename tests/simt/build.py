"""Host build of the binning kernels on the SIMT emulation shim (test infrastructure; see shim/hip/hip_runtime.h).

    python -m tests.simt.build

Reads dimo_amd/csrc/binning.hip + common.hpp AS THEY ARE, substitutes the gfx950 inline-asm statements (the LDS-only
barrier and v_writelane_b32) by their emulation calls, and compiles with g++ into tests/simt/_build/libbinning_emu.so.
The .hip sources carry no host / emulation switches of their own.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
CSRC = os.path.join(ROOT, "dimo_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libbinning_emu.so")

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


def _transformed(name):
    text = open(os.path.join(CSRC, name)).read()
    for old, new in SUBST[name]:
        assert text.count(old) >= 1, (name, old)
        text = text.replace(old, new)
    assert "asm" not in text.replace("__builtin_amdgcn", ""), name + ": an inline-asm statement without a substitution"
    return text


def build(force=False):
    srcs = [os.path.join(CSRC, n) for n in SUBST] + [os.path.join(HERE, n) for n in
            ("runtime.cpp", "binning_emu.cpp", "build.py", os.path.join("shim", "hip", "hip_runtime.h"))] + \
           [os.path.join(ROOT, "include", "dimo_hip.h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(s) <= os.path.getmtime(LIB) for s in srcs):
        return LIB
    # the transformed sources keep their relative include paths: _build/src/dimo_amd/csrc/{common.hpp, binning_src.inc}
    d = os.path.join(OUT, "src", "dimo_amd", "csrc")
    os.makedirs(d, exist_ok=True)
    os.makedirs(os.path.join(OUT, "src", "include"), exist_ok=True)
    open(os.path.join(d, "common.hpp"), "w").write(_transformed("common.hpp"))
    open(os.path.join(d, "binning_src.inc"), "w").write(_transformed("binning.hip"))
    open(os.path.join(OUT, "src", "include", "dimo_hip.h"), "w").write(open(os.path.join(ROOT, "include", "dimo_hip.h")).read())
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", "-fno-omit-frame-pointer",
           "-I", os.path.join(HERE, "shim"), "-I", d, "-Wno-unused-function",
           os.path.join(HERE, "runtime.cpp"), os.path.join(HERE, "binning_emu.cpp"), "-o", LIB]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True))

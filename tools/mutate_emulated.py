"""How strong are the CPU-emulated kernel tests?  (no GPU)

    python tools/mutate_emulated.py [--per-file 6] [--seed 1] [--files binning.hip,blend.hip,...]

For every chosen source of dimo_amd/csrc a few MUTANTS are made -- one small textual change each: a comparison
loosened or tightened, an off-by-one, min <-> max, a numeric constant nudged, a barrier removed -- built for the
emulation (tests/simt/build.py, SIMT_MUTANT) into a scratch directory and run against the emulated tests of that source;
a barrier mutant that survives the forward fiber order is run again with the fibers shuffled.  A mutant the tests do not
fail on SURVIVES: either the change does not alter the results (an equivalent mutant: a tie that cannot occur, a slack
bound) or the tests have a hole -- the report lists them with their line for inspection.  Mutants that do not compile
are dropped.  The product library is never touched."""
import argparse
import os
import random
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
CSRC = os.path.join(ROOT, "dimo_amd", "csrc")
TESTS = {
    "binning.hip": ["tests/test_binning_emulated.py", "tests/test_raster_emulated.py"],
    "blend.hip": ["tests/test_raster_emulated.py"],
    "preprocess.hip": ["tests/test_projection_emulated.py", "tests/test_raster_emulated.py"],
    "proj_math.hpp": ["tests/test_projection_emulated.py", "tests/test_raster_emulated.py"],
    "deform.hip": ["tests/test_deform_emulated.py", "tests/test_executor_emulated.py"],
    "deform_body.hpp": ["tests/test_deform_emulated.py", "tests/test_executor_emulated.py"],
    "wave_ops.hpp": ["tests/test_deform_emulated.py", "tests/test_raster_emulated.py"],
    "adam.hip": ["tests/test_deform_emulated.py"],
    "ssim.hip": ["tests/test_losses_emulated.py"],
    "image_loss.hip": ["tests/test_losses_emulated.py"],
    "loss_terms.hpp": ["tests/test_losses_emulated.py"],
    "timenet.hip": ["tests/test_timenet_emulated.py"],
    "knn.hip": ["tests/test_points_emulated.py"],
    "fps.hip": ["tests/test_points_emulated.py"],
    "executor.hip": ["tests/test_executor_emulated.py"],
}
# (pattern, replacement): applied to ONE occurrence in code (not in comments)
OPERATORS = [
    (r" <= ", " < "), (r" < ", " <= "), (r" >= ", " > "), (r" > ", " >= "),
    (r" \+ 1\b", " + 0"), (r" - 1\b", " - 0"), (r"\bmin\(", "max("), (r"\bmax\(", "min("),
    (r"\bfminf\(", "fmaxf("), (r"\bfmaxf\(", "fminf("), (r" && ", " || "), (r" == ", " != "),
    (r"__syncthreads\(\);", ";"), (r"lds_barrier\(\);", ";"),
    (r"\b0\.5f\b", "0.51f"), (r"\b2\.0f\b", "2.02f"), (r"\b1\.0f\b", "0.99f"),
]


def code_spans(text):
    """Offsets that are code: not inside // or /* */ comments, string literals or preprocessor lines."""
    keep = [True] * len(text)
    for m in re.finditer(r"//[^\n]*|/\*.*?\*/|\"(?:\\.|[^\"\\])*\"|^[ \t]*#[^\n]*", text, re.S | re.M):  # (a macro's continuation lines ARE code)
        for i in range(m.start(), m.end()):
            keep[i] = False
    return keep


def candidates(name):
    text = open(os.path.join(CSRC, name)).read()
    keep = code_spans(text)
    out = []
    for pat, rep in OPERATORS:
        for m in re.finditer(pat, text):
            old = m.group(0)
            # build.py replaces the k-th LITERAL occurrence of `old` (counted over the raw text: its own substitutions,
            # applied first, do not contain these patterns)
            k = text.count(old, 0, m.start())
            if keep[m.start()] and "static_assert" not in text[text.rfind("\n", 0, m.start()):m.start()]:
                new = re.sub(pat, rep, old)
                out.append((name, k, old, new, text.count("\n", 0, m.start()) + 1))
    return out


def run(mutant, tests, order=None, timeout=900):
    scratch = tempfile.mkdtemp(dir=SCRATCH)  # (a directory of its own: build.py would take the previous mutant's library)
    env = dict(os.environ, SIMT_MUTANT="%s|%d|%s|%s" % mutant[:4], SIMT_BUILD_DIR=scratch, PYTHONPATH=ROOT)
    if order:
        env["SIMT_ORDER"] = order
    try:
        p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + tests, cwd=ROOT, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    except subprocess.TimeoutExpired:
        shutil.rmtree(scratch, ignore_errors=True)
        return "killed (timeout: a hang)"
    shutil.rmtree(scratch, ignore_errors=True)
    out = p.stdout.decode(errors="replace")
    if ("CalledProcessError" in out and "g++" in out) or "SIMT_MUTANT lands inside" in out:
        return "does not compile"
    if p.returncode == 0:
        return "SURVIVED"
    if p.returncode < 0 or "Fatal Python error" in out or "deadlock" in out:
        return "killed (crash / deadlock report)"
    return "killed"


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--per-file", type=int, default=6)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--files", default=",".join(TESTS))
    a = ap.parse_args()
    SCRATCH = tempfile.mkdtemp(prefix="simt_mutants_")
    rng = random.Random(a.seed)
    tally = {}
    try:
        for name in a.files.split(","):
            c = candidates(name)
            rng.shuffle(c)
            done = 0
            for mut in c:
                if done >= a.per_file:
                    break
                res = run(mut, TESTS[name])
                if res == "SURVIVED" and ("barrier" in mut[2] or "__syncthreads" in mut[2]):
                    res = run(mut, TESTS[name], order="random:1")
                    res = res if res != "SURVIVED" else "SURVIVED (also with the fibers shuffled)"
                    if res.startswith("killed"):
                        res += " -- only with the fibers shuffled"
                if res == "does not compile":
                    continue
                done += 1
                tally[res.split(" ")[0]] = tally.get(res.split(" ")[0], 0) + 1
                print("%-16s line %4d  %-18r -> %-12r %s" % (name, mut[4], mut[2], mut[3], res), flush=True)
        print("total:", ", ".join("%s %d" % kv for kv in sorted(tally.items())))
    finally:
        shutil.rmtree(SCRATCH, ignore_errors=True)

"""Mean SQ counters per kernel from rocprofv3 --pmc counter_collection.csv files.

    pmc_sq_summarise.py <csv> [pattern ...]                                   (prints one line per pattern)
    pmc_sq_summarise.py --json out.json --csv a.csv b.csv ... -- pattern ...  (merged passes + derived figures)

Derived (per launch, MI355X: 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE comes back SUMMED over the 8 XCDs, so the kernel's
cycle count is a eighth of it; a wave64 VALU instruction occupies its SIMD for 2 cycles if it is an fma / mul / add /
mov / and / or on VGPR operands, 4 for cmp / min / max / cndmask / shifts / cvt / any DPP or SGPR-operand form, 8 for
exp / rcp / sqrt / log -- profiles/r02_valu_issue_rates.txt; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are in
quad-cycles summed over waves -- MI355X_MICROARCH.md):
  valu_issue_frac   = 2 * SQ_INSTS_VALU / (1024 * cycles)   LOWER bound of the VALU pipe's busy share (all 2-cycle)
  valu_busy_quad    = 4 * SQ_ACTIVE_INST_VALU / (1024 * cycles)   the same with every instruction priced at 4 cycles
  wave_active_valu  = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES              share of a wave's life spent issuing VALU
  wave_wait_any     = SQ_WAIT_ANY / SQ_WAVE_CYCLES                      parked on s_waitcnt / barrier
  wave_wait_inst    = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                 ready but not issued (pipe / dependency stall)
  waves_per_simd    = 4 * SQ_WAVE_CYCLES / (1024 * cycles)              achieved occupancy
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import csv
import json
import sys
from collections import defaultdict


def collect(paths, pats):
    acc = defaultdict(lambda: defaultdict(list))
    for path in paths:
        for r in csv.DictReader(open(path)):
            for p in pats:
                if p in r["Kernel_Name"]:
                    acc[p][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    a = sys.argv[1:]
    if a and a[0] == "--json":
        out = a[1]
        sep = a.index("--")
        paths, pats = [p for p in a[3:sep] if p], a[sep + 1:]
    else:
        out, paths, pats = None, a[:1], a[1:] or ["blend_bwd", "blend_fwd"]
    acc = collect(paths, pats)
    res = {}
    for p, d in acc.items():
        m = {k: sum(v) / len(v) for k, v in d.items()}
        m["launches"] = len(next(iter(d.values())))
        g, wc = m.get("GRBM_GUI_ACTIVE"), m.get("SQ_WAVE_CYCLES")
        g = g / 8.0 if g else g  # summed over the 8 XCDs
        der = {}
        if g and "SQ_INSTS_VALU" in m:
            der["valu_issue_frac"] = 2.0 * m["SQ_INSTS_VALU"] / (1024.0 * g)
            der["kernel_cycles"] = g
        if g and "SQ_ACTIVE_INST_VALU" in m:
            der["valu_busy_quad"] = 4.0 * m["SQ_ACTIVE_INST_VALU"] / (1024.0 * g)
        if wc:
            for name, key in (("wave_active_valu", "SQ_ACTIVE_INST_VALU"), ("wave_active_any", "SQ_ACTIVE_INST_ANY"),
                              ("wave_wait_any", "SQ_WAIT_ANY"), ("wave_wait_inst", "SQ_WAIT_INST_ANY"),
                              ("wave_active_lds", "SQ_ACTIVE_INST_LDS"), ("wave_active_scalar", "SQ_ACTIVE_INST_SCA")):
                if key in m:
                    der[name] = m[key] / wc
            if g:
                der["waves_per_simd"] = 4.0 * wc / (1024.0 * g)
        if m.get("SQ_LDS_IDX_ACTIVE"):
            der["lds_conflict_frac"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
        if m.get("SQ_WAVES"):
            der["valu_per_wave"] = m.get("SQ_INSTS_VALU", 0.0) / m["SQ_WAVES"]
        res[p] = {"counters": {k: round(v, 1) for k, v in m.items()}, "derived": {k: round(v, 4) for k, v in der.items()}}
        print(p, res[p])
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()

"""Prints the per-wave instruction mix and occupancy figures of a pmc_sq.sh summary: python tools/pmc_show.py <json>"""
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    c, der = v["counters"], v["derived"]
    w = c["SQ_WAVES"]
    print(k)
    print("   cycles %.0f (%.1f us @2.4GHz) waves %d | per wave: valu %.0f salu %.0f lds %.0f vmem_rd %.1f vmem_wr %.1f smem %.1f branch %.0f" % (
        der.get("kernel_cycles", 0), der.get("kernel_cycles", 0) / 2400, w, c["SQ_INSTS_VALU"] / w, c["SQ_INSTS_SALU"] / w,
        c["SQ_INSTS_LDS"] / w, c["SQ_INSTS_VMEM_RD"] / w, c["SQ_INSTS_VMEM_WR"] / w, c["SQ_INSTS_SMEM"] / w, c["SQ_INSTS_BRANCH"] / w))
    print("   ", {k2: der[k2] for k2 in der if k2 != "kernel_cycles"})

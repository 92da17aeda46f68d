"""Summarises rocprofv3 --pmc counter CSVs (one pass per counter) into profiles/pmc_blend_bwd.json.

usage: pmc_summarise.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [renders per launch]
FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3).  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports
half of the bytes of a wide coalesced streaming read; the 1 GiB calibration copy in the same run measures the actual
factor for this environment, which is then applied to the kernel of interest."""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    f, w, out = sys.argv[1:4]
    renders_per_launch = float(sys.argv[4]) if len(sys.argv) > 4 else 4.0  # pmc_probe: one motion (4 renders) per batch
    fetch, write = per_kernel(f, "FETCH_SIZE"), per_kernel(w, "WRITE_SIZE")
    find = lambda d, pat: next((v for k, v in d.items() if pat in k), [])
    mean = lambda xs: sum(xs) / len(xs) if xs else None
    GiB_KiB = float(1 << 20)
    # the 1 GiB clone runs as __amd_rocclr_copyBuffer: take its largest dispatches (other copies are tiny)
    big = lambda d: [v for v in find(d, "__amd_rocclr_copyBuffer") if v > 100000.0]
    cal_f, cal_w = mean(big(fetch)), mean(big(write))
    res = {"units": "bytes per launch", "calibration": {"kernel": "torch clone of 1 GiB (1 GiB read + 1 GiB write)",
                                                         "FETCH_SIZE_KiB": cal_f, "WRITE_SIZE_KiB": cal_w}}
    kf = (GiB_KiB / cal_f) if cal_f else 2.0
    kw = (GiB_KiB / cal_w) if cal_w else 1.0
    res["calibration"]["fetch_factor"], res["calibration"]["write_factor"] = kf, kw
    for name, pat in (("blend_bwd", "blend_bwd_batched_kernel"), ("blend_fwd", "blend_fwd_batched_kernel"),
                      ("preprocess_bwd", "preprocess_bwd_batched_kernel"), ("level2_fill", "level2_fill_batched_kernel")):
        fk, wk = mean(find(fetch, pat)), mean(find(write, pat))
        if fk is None or wk is None:
            continue
        res[name] = {"FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk, "launches": len(find(fetch, pat)),
                     "hbm_bytes_per_launch": (fk * kf + wk * kw) * 1024.0}
    if "blend_bwd" in res:
        res["hbm_bytes_per_launch"] = res["blend_bwd"]["hbm_bytes_per_launch"]
        res["renders_per_launch"] = renders_per_launch
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

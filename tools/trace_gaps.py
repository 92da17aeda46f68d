"""GPU busy/idle analysis from a rocprofv3 kernel trace CSV: union of kernel intervals over the last steps."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# steady state = the last 5 steps, delimited by the Adam kernel that ends every step
adam = [x for x in iv if "flat_adam" in x[2]]
# (a step may take its optimizer update as two launches -- the per-Gaussian head early, the tail at the end: launches
# less than 0.4 ms apart belong to one step, whose end is the later one)
_ends = []
for a in adam:
    if _ends and a[0] - _ends[-1][1] < 400_000:
        _ends[-1] = a
    else:
        _ends.append(a)
adam = _ends
nsteps = min(5, len(adam) - 1)
t0, t1 = adam[-1 - nsteps][1], adam[-1][1]
iv = [x for x in iv if x[0] >= t0 and x[1] <= t1]
print("steps analysed:", nsteps, " ms/step:", (t1 - t0) / 1e6 / nsteps)
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e, _ in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = iv[-1][1] - iv[0][0]
print(f"span {span/1e6:.2f} ms  busy(union) {busy/1e6:.2f} ms = {100*busy/span:.1f}%  kernels {len(iv)}  sum-of-durations {sum(e-s for s,e,_ in iv)/1e6:.2f} ms")
gaps.sort(reverse=True)
print("largest gaps (us):", [round(g[0]/1e3,1) for g in gaps[:15]])
from collections import defaultdict
per = defaultdict(int)
for s_, e_, n_ in iv: per[n_.split("(")[0][-48:]] += e_ - s_
print("top kernels (ms/step):", [(k, round(v/1e6/nsteps, 3)) for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:14]])
print("gaps > 20us total", sum(g[0] for g in gaps if g[0] > 20000)/1e6, "ms ;  gaps < 20us total", sum(g[0] for g in gaps if g[0] <= 20000)/1e6, "ms, count", len(gaps))

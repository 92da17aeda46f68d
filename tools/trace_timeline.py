"""Compact per-queue timeline of the last full step in a rocprofv3 kernel trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r["Queue_Id"]) for r in rows)
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
t0, t1 = adam[-2][1], adam[-1][1]
step = [x for x in iv if x[0] >= t0 and x[1] <= t1]
print("step ms", (t1 - t0) / 1e6, "kernels", len(step))
queues = sorted({x[3] for x in step})
for q in queues:
    ks = [x for x in step if x[3] == q]
    busy = sum(e - s for s, e, _, _ in ks)
    print(f"queue {q}: {len(ks)} kernels, busy {busy/1e6:.2f} ms, first {((ks[0][0]-t0)/1e6):.2f} last {((ks[-1][1]-t0)/1e6):.2f}")
# coarse phases on each queue: print kernels longer than 30us with start offsets
for q in queues:
    print("--- queue", q)
    for s, e, n, _ in [x for x in step if x[3] == q]:
        if e - s > (0 if len(sys.argv) > 2 else 15000):
            print(f"  {(s-t0)/1e3:8.1f} us  +{(e-s)/1e3:6.1f}  {n}")

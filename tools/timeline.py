"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: kernels in start order with queue, start offset,
duration and the gap to the previous kernel's end on the same queue.  usage: timeline.py <kernel_trace.csv> [step_from_end]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r["Queue_Id"],
              r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", "")) for r in rows))
# a step starts at the first kernel after each adam kernel
adam = [i for i, k in enumerate(ks) if "adam" in k[2]]
ends = adam
lo, hi = ends[-back - 1] + 1, ends[-back] + 1
t0 = ks[lo][0]
last = {}
busy_end = t0
idle = 0
for s, e, n, q, g, w in ks[lo:hi]:
    gap = s - last.get(q, s)
    if s > busy_end:
        idle += s - busy_end
    busy_end = max(busy_end, e)
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  q{q} gap {gap / 1e3:6.1f}  {n} g{g}/{w}")
    last[q] = e
print("step span %.1f us, all-queue idle %.1f us" % ((busy_end - t0) / 1e3, idle / 1e3))

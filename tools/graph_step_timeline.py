"""One hipGraph-replayed training step from a rocprofv3 kernel trace of bench.py: kernels in start order with start offset, duration
and the gap to the previous kernel's end.  usage: python tools/graph_step_timeline.py <kernel_trace.csv>  (picks a step of typical span)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:70]) for r in rows)
adam = [i for i, k in enumerate(ks) if 'adam' in k[2]]
spans = [(ks[adam[i]][1] - ks[adam[i - 1]][1]) / 1e3 for i in range(1, len(adam))]
# the replayed steps: the most common dispatch count among the adam-to-adam intervals without bench.py's bracket-calibration kernels
from collections import Counter
plain = [i for i in range(1, len(adam)) if not any('spin' in k[2] for k in ks[adam[i - 1] + 1:adam[i] + 1])]
count = Counter(adam[i] - adam[i - 1] for i in plain).most_common(1)[0][0]
plain = [i for i in plain if adam[i] - adam[i - 1] == count]
med = sorted(spans[i - 1] for i in plain)[len(plain) // 2]
idx = [i for i in plain if abs(spans[i - 1] - med) < 0.02 * med]
i = idx[len(idx) // 2]
print(f"{len(plain)} intervals of {count} dispatches without calibration kernels; median span {med:.1f} us")
lo, hi = adam[i - 1] + 1, adam[i] + 1
t0, prev_end = ks[lo][0], ks[adam[i - 1]][1]
tot = gaps = 0.0
for s_, e, n in ks[lo:hi]:
    gap = (s_ - prev_end) / 1e3
    prev_end = e
    print(f"{(s_ - t0) / 1e3:8.1f} +{(e - s_) / 1e3:7.1f} gap {gap:5.1f} {n}")
    tot += (e - s_) / 1e3
    gaps += gap
print(f"dispatches {hi - lo}  kernel time {tot:.1f} us  gaps {gaps:.1f} us  (median step span of the trace {med:.1f} us)")

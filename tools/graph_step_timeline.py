"""One hipGraph-replayed training step from a rocprofv3 kernel trace of bench.py: kernels in start order with start offset, duration
and the gap to the previous kernel's end.  usage: python tools/graph_step_timeline.py <kernel_trace.csv>  (picks a step of typical span)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:70]) for r in rows)
adam = [i for i, k in enumerate(ks) if 'adam' in k[2]]
spans = [(ks[adam[i]][1] - ks[adam[i - 1]][1]) / 1e3 for i in range(1, len(adam))]
med = sorted(spans)[len(spans) // 2]
idx = [i for i, s_ in enumerate(spans) if abs(s_ - med) < 0.02 * med]
i = idx[len(idx) // 2] + 1
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

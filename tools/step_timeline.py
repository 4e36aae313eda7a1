"""Timeline of ONE training step from a rocprofv3 kernel trace: per kernel start offset, duration, queue; plus the
busy time per queue and the gaps.  usage: python tools/step_timeline.py <kernel_trace.csv> [step_index]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "increment_kernel" in r["Kernel_Name"]]
# one adam launch per step (flat buffer) -> step k = rows between adam k-1 and adam k
i0, i1 = adam[which - 1] + 1, adam[which] + 1
seg = rows[i0:i1]
t0 = int(seg[0]["Start_Timestamp"])
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)
    return n[:48]
busy = {}
last_end = {}
print(f"step {which}: {len(seg)} dispatches, span {(int(seg[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
for r in seg:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = r["Queue_Id"]
    gap = s - last_end.get(q, s)
    busy[q] = busy.get(q, 0) + (e - s)
    last_end[q] = e
    print(f"{s / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{q} gap{gap / 1e3:7.1f}  {short(r['Kernel_Name'])} [{r['Grid_Size_X']},{r['Grid_Size_Y']}]")
print({q: round(v / 1e3, 1) for q, v in busy.items()})

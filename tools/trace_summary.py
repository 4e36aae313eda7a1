#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid): python tools_trace_summary.py <kernel_trace.csv> [steps]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')[-44:]
    key = (name, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), r['Grid_Size_Y'], r['Grid_Size_Z'],
           r['LDS_Block_Size'], r['VGPR_Count'], r['Accum_VGPR_Count'])
    agg[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = sum(sum(v) for v in agg.values())
print(f"total kernel time {tot/1e6:.3f} ms over {len(rows)} dispatches; per step {tot/1e6/steps:.3f} ms")
print("  %    n/step   avg_us   kernel (wgs_x, grid_y, grid_z, lds, vgpr, agpr)")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:45]:
    print(f"{sum(v)/tot*100:5.1f} {len(v)/steps:7.1f} {sum(v)/len(v)/1e3:9.1f}  {k}")

#!/usr/bin/env python3
"""Prints the kernel timeline of ONE training step from a rocprofv3 kernel trace (the launches between the last two
optimiser launches): start offset, duration, kernel, grid.   python3 tools/step_timeline.py <dir or kernel_trace.csv>"""
import csv
import glob
import os
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adadelta" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["End_Timestamp"])
busy = 0.0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += (e - s) / 1e3
    print(f'{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {r["Kernel_Name"].replace("void ", "").replace("vgan::", "")[:64]:64s} grid={int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])}x{r["Workgroup_Size_X"]}')
print(f"step {(int(rows[b]['End_Timestamp']) - t0) / 1e3:.1f} us, kernels {busy:.1f} us")

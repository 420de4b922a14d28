#!/usr/bin/env python3
"""Prints the kernels of one mid-run cycle from a rocprofv3 --kernel-trace CSV of an engine run:
start and end in microseconds from the cycle's first kernel (the particles' kick and drift),
duration, hardware queue and name.  usage: tools/step_timeline.py <kernel_trace.csv> [cycle]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if "k_kick_drift" in r["Kernel_Name"]]
c = int(sys.argv[2]) if len(sys.argv) > 2 else len(first) // 2
a, b = first[c], first[c + 1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    e = (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:8.1f} {e:8.1f} {e - s:7.1f} q{r['Queue_Id']} {r['Kernel_Name'][:72]}")

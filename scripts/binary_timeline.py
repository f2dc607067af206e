#!/usr/bin/env python3
"""Dev helper: the launches of a few steady-state `binary` steps from a rocprofv3 kernel trace (start, end, duration in us; queue).
usage: python scripts/binary_timeline.py <kernel_trace.csv> [first FAST stage launch to show = 10] [rows = 24]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 10
count = int(sys.argv[3]) if len(sys.argv) > 3 else 24
idx = [i for i, r in enumerate(rows) if 'binary_stage_kernel<mh::BinFast' in r['Kernel_Name']]
t0 = int(rows[idx[first]]['Start_Timestamp'])
for r in rows[idx[first]:idx[first] + count]:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    print("%9.1f %9.1f %7.1f q=%s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r['Queue_Id'], r['Kernel_Name'][:64]))

#!/usr/bin/env python3
"""Dev helper: the last N launches of a rocprofv3 kernel trace (start, end, duration in us relative to the first shown; queue; name).
usage: python scripts/any_timeline.py <kernel_trace.csv> [N=40]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-n:]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    print("%9.1f %9.1f %7.1f q=%s grid=%s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r['Queue_Id'], r.get('Grid_Size_X', ''), r['Kernel_Name'][:70]))

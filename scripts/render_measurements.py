#!/usr/bin/env python3
"""Render DESIGN.md section 6's tables from a driver-style bench line (python bench.py --gpus 1 --steps 20 --warmup 5).
usage: python scripts/render_measurements.py profiles/r04/bench_driverstyle_end_of_round.json"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
L = d["legs"]
rows = [("**FAST + HLLC, blast - headline `value`** (planar kernel)", d, d["roofline"], d["roofline_step"]),
        ("FAST + HLLC, blast, general fused kernel (`--no-planar`)", L["fast_hllc_blast_general_kernel"], None, None),
        ("FAST + HLLE, blast (planar kernel; within 1e-12 of the reference)", L["fast_hlle_blast"], None, None),
        ("FAST + HLLC, blast, two launches (`--no-fuse`)", L["fast_hllc_blast_two_launches"], None, None),
        ("STRICT + HLLC, blast", L["strict_hllc_blast"], None, None),
        ("STRICT + HLLE, blast (bit-identical to the reference)", L["strict_hlle_blast"], None, None),
        ("FAST + HLLC, smooth periodic wave (planar kernel)", L["fast_hllc_smooth_wave"], None, None),
        ("STRICT + HLLE, smooth periodic wave", L["strict_hlle_smooth_wave"], None, None)]
print("| leg (4096², PLM θ = 1.5, RK2, 1 GPU) | launches / step | Mcells/s | ms / step | dominant kernel: ms per launch, frac of 8 TB/s at the contract's bytes | step frac (200 B) | bytes the launch moves per cell → frac of 8 TB/s |")
print("|---|---:|---:|---:|---|---:|---|")
for name, leg, r, rs in rows:
    r = r or leg["roofline"]; rs = rs or leg["roofline_step"]
    lps = leg.get("launches_per_step", d["config"].get("summary") and 1)
    if leg is d:
        lps = 1
    moved = "%d B → %.2f" % (r["bytes_actually_moved_per_cell"], r["frac_actual_traffic"]) if "frac_actual_traffic" in r else "stage 2: 120 B (algorithmic)"
    extra = ""
    if leg is d and "repeat_blocks" in d:
        extra = " (blocks 2-5: %.3f-%.3f)" % (d["repeat_blocks"]["min"], d["repeat_blocks"]["max"])
    print("| %s | %d | %s | %.3f%s | %.3f, %.3f | %.3f | %s |" % (name, lps, "**%d**" % round(leg["value"]) if leg is d else "%d" % round(leg["value"]), leg["ms_per_step"], extra,
          r["avg_launch_ms"], r["frac"], rs["frac"], moved))
print()
E = d.get("extra_configs", {})
print("| config (same line, `extra_configs`) | FAST Mzones/s | ms / step | launches / step | kernel: ms per launch, frac (contract bytes) | step frac | STRICT Mzones/s | CPU beside it |")
print("|---|---:|---:|---:|---|---:|---:|---|")
for cfg, label in (("c3", "C3 `binary` 2048² (168 B per zone-update; 184 B with the buffer-rate array)"), ("c4", "C4 `cloud` 4096² (compiled host, nozzle upload per step)"), ("c5", "C5 3-D 512³ (one rank's share of 1024³)")):
    e = E.get(cfg, {})
    if "value" not in e:
        print("| %s | error | | | | | | |" % label); continue
    r = e["roofline"]
    lps = r.get("launches_per_step", e.get("config", {}).get("launches_per_step", 2))
    stepf = (e.get("roofline_step") or {}).get("frac")
    if stepf is None:
        bytes_zu = 168 if cfg == "c3" else 200
        cells = {"c3": 2048 * 2048, "c5": 512 ** 3}.get(cfg, 4096 * 4096)
        stepf = cells * bytes_zu / (e["ms_per_step"] * 1e-3) / 1e9 / 8000.0
    cpu = e.get("cpu_baseline") or {}
    ref = e.get("cpu_reference") or {}
    cpus = "%s, %d threads: %.1f" % ("the reference's evaluator" if cpu.get("kind") == "reference" else "oracle port", cpu.get("cores", 0), cpu.get("value", 0.0)) if cpu else ""
    if ref:
        cpus += "; the reference's evaluator, %d threads: %.1f" % (ref["cores"], ref["value"])
    print("| %s | **%d** | %.3f | %s | %.3f, %.3f%s | %.3f | %d | %s |" % (label, round(e["value"]), e["ms_per_step"], lps, r["avg_launch_ms"], r["frac"],
          " (%.2f at 184 B)" % r["frac_at_184_bytes"] if "frac_at_184_bytes" in r else "", stepf, round(e["arith_strict"]["value"]), cpus))
print()
cb, cr = d.get("cpu_baseline"), d.get("cpu_reference")
if cb:
    print("CPU beside the headline, same box, same run: the oracle's C port, %d threads at the full 4096²: **%.1f Mcells/s**; " % (cb["cores"], cb["value"]), end="")
if cr:
    print("the reference's own lazy-array composition through ITS threaded evaluator `mara::evaluate_on<%d>()` at 1024²: **%.2f Mcells/s** (one thread: %.2f)." % (cr["cores"], cr["value"], cr["one_thread"]))
for k in d:
    if k.startswith("l1_fast_vs_strict"):
        print("`%s` = %.1e." % (k, d[k]))

#!/usr/bin/env python3
"""Collate the rocprofv3 passes of scripts/profile_r2.sh (gpurun_out/prof_r2) into the committed evidence:
   profiles/r02/kernels_<part>.md     per kernel: avg duration (kernel trace), HBM bytes per launch from PMC (2*FETCH_SIZE + WRITE_SIZE, KB;
                                      FETCH_SIZE tallies 64 B per 128-B request on gfx950, re-calibrated on mh::stream_copy_kernel in the same
                                      script), VALU instructions and utilisation, fp64 FLOP/s from the SQ_INSTS_VALU_*_F64 counters
   profiles/pmc_traffic.json          bytes per launch keyed as bench.py reads them, stamped with the hash of the kernel sources"""
import csv, glob as _glob, json, os, sys, collections


class glob:
    """one rocprofv3 run per directory is what the tables mean: gpurun MERGES a box's output into gpurun_out/, so a pass that is repeated leaves
    its files beside the earlier run's (another process id in the name) - only the newest run of a directory is read"""
    @staticmethod
    def glob(pattern, recursive=False):
        files = _glob.glob(pattern, recursive=recursive)
        newest = {}
        for f in files:
            d = os.path.dirname(f)
            if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
                newest[d] = f
        return sorted(newest.values()) if files and os.path.isfile(files[0]) else files

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_fingerprint
ROUND = os.environ.get("MH_PROF_ROUND", "r02")           # r03: scripts/profile_r3.sh -> gpurun_out/prof_r3 -> profiles/r03
OUT = os.path.join(ROOT, "gpurun_out", "prof_" + ROUND.replace("0", ""))
DST = os.path.join(ROOT, "profiles", ROUND)
os.makedirs(DST, exist_ok=True)


def counters(tag, kind):
    acc, dur = collections.defaultdict(list), collections.defaultdict(list)
    for f in glob.glob(os.path.join(OUT, "%s_%s" % (tag, kind), "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "mh::" not in k:
                continue
            short = k.split("(")[0].replace("void mh::", "").replace("mh::", "")
            acc[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
            dur[short].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: sum(v) / len(v) for k, v in dur.items()}


def trace(tag):
    out = {}
    for f in glob.glob(os.path.join(OUT, tag + "_trace", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Name"]
            if "mh::" in k:
                out[k.split("(")[0].replace("void mh::", "").replace("mh::", "")] = (float(r["AverageNs"]), int(r["Calls"]), float(r["MinNs"]))
    return out


def calibration():
    f, _ = counters("calib", "fetch")
    w, _ = counters("calib", "write")
    known = 5 * 4096 * 4096 * 8
    return known / (f[("stream_copy_kernel", "FETCH_SIZE")] * 1024), known / (w[("stream_copy_kernel", "WRITE_SIZE")] * 1024)


def table(tags, part):
    ffac, wfac = calibration()
    lines = ["# rocprofv3 summary, round %s (%s), one MI355X" % (ROUND[-1], part), "",
             ("Source: `scripts/profile_r" + ROUND[-1] + ".sh %s` (one rocprofv3 pass per counter group; program directly after `--`), collated by `scripts/pmc_collate.py`.") % part,
             "Calibration on `mh::stream_copy_kernel` (671 088 640 B read and written, 8 B per lane): bytes / (FETCH_SIZE KB x 1024) = %.3f, bytes / (WRITE_SIZE KB x 1024) = %.3f"
             % (ffac, wfac), "=> HBM bytes per launch = %.0f x FETCH_SIZE + %.0f x WRITE_SIZE (KB x 1024)." % (round(ffac), round(wfac)), "",
             "| run | kernel | avg us (kernel trace, ALL launches of the run: cold first ones included) | min us | calls | HBM MB / launch | VALU instr / launch | VALU busy | wait-on-instr | fp64 TFLOP/s | of 78.6 |",
             "|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    if part == "headline" and ROUND >= "r03":
        own = {}
        try:
            for l in open(os.path.join(OUT, "fast_hllc_trace.json")):
                if l.startswith("{"):
                    own = json.loads(l)
        except OSError:
            pass
        tr0 = ([v for k, v in trace("fast_hllc").items() if k.startswith("euler2d_fused_rk2_kernel<1")] or [(0.0, 0, 0.0)])[0]
        steps, launches = ("1000", "1100") if ROUND >= "r05" else ("200", "300")
        lines[3:3] = ["Kernel traces of the headline variants: `bench.py --steps %s` under `rocprofv3 --kernel-trace --stats` (about %s launches per kernel, so that the cold launches - the first ~25 after" % (steps, launches),
                      "an idle period run up to twice as long - do not carry the average; the counters come from 20-step runs). Since round 4 the tracked `kernel_stats_<run>.csv` beside this file ARE these traces",
                      "(tests/test_profiles_cpu.py holds the table to them). The same command's own bench line is",
                      "`bench_under_rocprof_trace_<run>.json`: for the fused kernel %.3f ms per launch (one pair of events around five launches, gaps included) and %.3f ms per timed step against"
                      % (own.get("roofline", {}).get("avg_launch_ms", 0.0), own.get("ms_per_step", 0.0)),
                      "this trace's %.3f ms average / %.3f ms minimum (same process, same box; boxes differ by 2 - 3 %%)." % (tr0[0] / 1e6, tr0[2] / 1e6), ""]
    traffic, extra = {}, {}
    for tag in tags:
        tr = trace(tag)
        if ROUND >= "r04":          # the trace this table quotes, tracked beside it
            for f in glob.glob(os.path.join(OUT, tag + "_trace", "**", "*kernel_stats.csv"), recursive=True):
                rows = [l for l in open(f) if l.startswith('"Name"') or "mh::" in l]
                open(os.path.join(DST, "kernel_stats_%s.csv" % tag), "w").write("".join(rows))
            own_line = os.path.join(OUT, tag + "_trace.json")
            if os.path.exists(own_line):
                keep = [l for l in open(own_line) if l.startswith("{")]
                if keep:
                    open(os.path.join(DST, "bench_under_rocprof_trace_%s.json" % tag), "w").write(keep[-1])
        fe, _ = counters(tag, "fetch")
        wr, _ = counters(tag, "write")
        sq, _ = counters(tag, "sq")
        fl, fdur = counters(tag, "flop")
        for k in sorted(tr):
            if not any(x in k for x in ("stage_kernel", "fused_rk2_kernel", "update_kernel", "flux_kernel", "gradient_kernel", "sink_kernel")):
                continue
            avg, calls, tmin = tr[k]
            hbm = None
            if (k, "FETCH_SIZE") in fe and (k, "WRITE_SIZE") in wr:
                hbm = (round(ffac) * fe[(k, "FETCH_SIZE")] + round(wfac) * wr[(k, "WRITE_SIZE")]) * 1024
                traffic[(tag, k)] = hbm
            valu = sq.get((k, "SQ_INSTS_VALU"))
            busy = wait = None
            if (k, "SQ_ACTIVE_INST_VALU") in sq and sq.get((k, "SQ_BUSY_CYCLES")):
                busy = sq[(k, "SQ_ACTIVE_INST_VALU")] * 4 / (sq[(k, "SQ_BUSY_CYCLES")] * 32)          # quad-cycles per SIMD over 32 SIMDs per shader engine
                wait = sq[(k, "SQ_WAIT_INST_ANY")] / sq[(k, "SQ_WAVE_CYCLES")]
            tf = None
            if (k, "SQ_INSTS_VALU_FMA_F64") in fl:
                flops = 64 * (fl[(k, "SQ_INSTS_VALU_ADD_F64")] + fl[(k, "SQ_INSTS_VALU_MUL_F64")] + 2 * fl[(k, "SQ_INSTS_VALU_FMA_F64")] + fl[(k, "SQ_INSTS_VALU_TRANS_F64")])
                tf = flops / (avg * 1e-9) / 1e12
                extra[(tag, k)] = {"fp64_flops_per_launch": flops, "valu_busy": busy, "valu_instructions_per_launch": valu}
            fmt = lambda x, f: "" if x is None else f % x
            lines.append("| %s | `%s` | %.1f | %.1f | %d | %s | %s | %s | %s | %s | %s |" % (tag, k, avg / 1e3, tmin / 1e3, calls, fmt(hbm and hbm / 1e6, "%.1f"), fmt(valu, "%.3g"),
                         fmt(busy, "%.2f"), fmt(wait, "%.2f"), fmt(tf, "%.1f"), fmt(tf and tf / 78.6, "%.2f")))
    lines += ["", "VALU busy = SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES x 32): the counter ticks in quad-cycles per SIMD, SQ_BUSY_CYCLES per shader engine of 32 SIMDs.",
              "wait-on-instr = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES. fp64 FLOP/s = 64 x (ADD + MUL + 2 FMA + TRANS of SQ_INSTS_VALU_*_F64) / avg duration."]
    open(os.path.join(DST, "kernels_%s.md" % part), "w").write("\n".join(lines) + "\n")
    return traffic, extra


if __name__ == "__main__":
    part = sys.argv[1] if len(sys.argv) > 1 else "headline"
    if part == "headline":
        tags = ["fast_hllc", "strict_hllc", "fast_hlle", "strict_hlle"] if ROUND == "r02" else ["fast_hllc", "fast_hllc_general", "fast_hllc_two", "fast_hlle", "strict_hlle", "strict_hllc"]
        tr, ex = table([t for t in tags if glob.glob(os.path.join(OUT, t + "_trace"))], "headline")
        path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        out = {"csrc_sha16": csrc_fingerprint(),
               "_comment": "HBM bytes per launch of the RK2 stage kernels at 4096^2 from rocprofv3 PMC (separate FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 20 "
                           "--warmup 3 --single-arith`, scripts/profile_r2.sh): 2 x FETCH_SIZE + WRITE_SIZE, calibrated on mh::stream_copy_kernel in the same script "
                           "(profiles/" + ROUND + "/kernels_headline.md). Keyed by the hash of the stage kernels' sources and device headers (bench.py: KERNEL_SOURCES): bench.py reports `traffic` only for the sources "
                           "these numbers were measured on."}
        for (tag, k), v in tr.items():
            arith, riemann = tag.split("_")[:2]
            stage = ("fused_planar" if k.endswith("true>") else "fused") if "fused" in k else ("stage2" if k.endswith("true>") else "stage1")
            out["%s_%s_%s_bytes_per_launch" % (stage, arith, riemann)] = v
            if (tag, k) in ex:
                out["%s_%s_%s_fp64" % (stage, arith, riemann)] = ex[(tag, k)]
        json.dump(out, open(path, "w"), indent=1)
        print(json.dumps(out, indent=1)[:1500])
    else:
        tr, ex = table(["c3", "c4", "c4two", "c4s", "c5"] if ROUND >= "r04" else ["c3", "c4", "c4s", "c5"], "configs")
        path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        t = json.load(open(path))
        for (tag, k), v in tr.items():
            t["%s:%s" % (tag, k)] = v
        for (tag, k), v in ex.items():
            t["%s:%s:fp64" % (tag, k)] = v
        json.dump(t, open(path, "w"), indent=1)
    print(open(os.path.join(DST, "kernels_%s.md" % part)).read())

"""What bench.py / bench_configs.py print: roofline objects that are HARDWARE fractions, and the compact final line.

Round 4's line had grown to 21.7 KB and the driver could not parse it (BENCH_r04.json: "parsed": null), and its `roofline.frac` was
SURVEY.md 8d's 200 B convention over a launch that moves 72 B per cell - a "fraction" that read 0.90 and, for a sibling leg, 1.14.
From round 5 on:

  * `roofline` of a ONE-launch RK2 step (the fused kernels) is bound "fp64": FLOP per launch (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 of the
    committed rocprofv3 PMC passes, per cell) over this run's average launch duration against the 78.6 TFLOP/s fp64 vector peak, with
    VALU-busy, the HBM bytes of the PMC passes as `traffic`, and `hbm_frac_measured` = those bytes / the launch time / 8 TB/s.
  * `roofline` of a stage kernel of a two-launch step stays bound "hbm": algorithmic bytes (80 B first stage, 120 B second) per cell.
  * every `roofline.frac` is in (0, 1). The 200 B x zone-updates figure lives ONLY in `roofline_step`, named for what it is:
    throughput in SURVEY 8(d) byte-equivalents - it can pass 8 TB/s because the fused launch never writes or re-reads the first-stage field.
  * the LAST stdout line is `final_line(details)`: at most 6000 characters; everything else (legs, configs in full, repeat blocks, notes)
    goes to bench_details.json next to bench.py and to stderr.
"""
import json
import os

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s
FP64_VECTOR_PEAK_TFLOPS = 78.6   # 256 CUs x 4 SIMDs x 16 lanes x 2 flop (FMA) x 2.4 GHz
BYTES_ZONE_UPDATE = 200          # RK2, five variables
TARGET_MCELLS = 16000.0          # 40 % of 8 TB/s at 200 B per zone-update
LINE_LIMIT = 6000

ROOT = os.path.dirname(os.path.abspath(__file__))
REF_CELLS = 4096 * 4096          # the launches profiles/pmc_traffic.json was recorded on (headline legs)


def sig(x, digits=6):
    return float("%.*g" % (digits, x)) if isinstance(x, float) else x


def round_floats(obj, digits=6):
    if isinstance(obj, dict):
        return {k: round_floats(v, digits) for k, v in obj.items()}
    if isinstance(obj, list):
        return [round_floats(v, digits) for v in obj]
    return sig(obj, digits)


class Counters:
    """profiles/pmc_traffic.json: HBM bytes, FLOP and VALU-busy per launch from rocprofv3 PMC passes (never measured inside a bench run: PMC
    collection needs its own passes). Stamped with the hash of the kernel sources it was taken on; `current` says whether that is this build.
    FLOP per cell is a property of the arithmetic and is used (flagged) across builds; bytes are reported as `traffic` only for the build
    they were measured on."""

    def __init__(self, fingerprint):
        try:
            self.table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        except Exception:
            self.table = {}
        self.sha = self.table.get("csrc_sha16")
        self.current = bool(self.sha) and self.sha == fingerprint

    def get(self, key):
        return self.table.get(key)

    def provenance(self):
        return ("rocprofv3 PMC passes of these kernel sources (profiles/pmc_traffic.json); durations are this run's" if self.current else
                "FLOP per cell from the PMC passes of an earlier build of the kernel (profiles/pmc_traffic.json, sources %s); no HBM bytes for this build" % self.sha)


def fp64_roofline(kernel, avg_ms, launches, cells, flops_per_cell, valu_busy, traffic_per_cell, nominal_bytes_per_cell, provenance, timing=None):
    """the one-launch RK2 step: fp64-issue-bound (VALU-busy 0.85 - 0.92), so the roofline is the fp64 vector peak"""
    tf = cells * flops_per_cell / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else None
    traffic = cells * traffic_per_cell if traffic_per_cell else None
    moved = traffic_per_cell if traffic_per_cell else nominal_bytes_per_cell
    r = {"bound": "fp64", "achieved": tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_VECTOR_PEAK_TFLOPS if tf else None,
         "traffic": traffic, "valu_busy": valu_busy,
         "hbm_frac_measured": (cells * moved / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if avg_ms > 0 else None,
         "bytes_moved_per_cell": moved, "bytes_moved_source": "PMC (FETCH_SIZE, WRITE_SIZE)" if traffic_per_cell else "nominal (reads + writes of the launch; no PMC record for this build)",
         "kernel": kernel, "avg_launch_ms": avg_ms, "launches": launches, "cells_per_launch": cells, "flops_per_launch": cells * flops_per_cell,
         "counters": provenance}
    if timing:
        r["timing"] = timing
    return r


def hbm_roofline(kernel, avg_ms, launches, cells, bytes_per_cell, traffic=None, timing=None, extra=None):
    """a stage kernel of a multi-launch step: algorithmic bytes per launch (SURVEY.md 8d per stage) over the launch duration against 8 TB/s"""
    nbytes = cells * bytes_per_cell
    ach = nbytes / (avg_ms * 1e-3) / 1e9 if avg_ms and avg_ms > 0 else None
    r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS if ach else None, "traffic": traffic,
         "kernel": kernel, "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": avg_ms, "launches": launches}
    if timing:
        r["timing"] = timing
    if extra:
        r.update(extra)
    return r


def step_equivalents(mcells_per_s_per_gpu, bytes_per_zone_update=BYTES_ZONE_UPDATE, target=TARGET_MCELLS):
    """NOT a roofline fraction: zone-updates/s x SURVEY 8(d)'s algorithmic bytes of a zone-update. BASELINE.md's target (40 % of 8 TB/s at
    200 B = 16 000 Mcells/s per GPU) is stated in this unit, so the ratio to that target is what this object is for."""
    gbs = mcells_per_s_per_gpu * 1e6 * bytes_per_zone_update / 1e9
    r = {"name": "throughput in SURVEY 8(d) byte-equivalents (zone-updates/s x %d B; not bytes moved - a one-launch step moves 72-80 B per cell)" % bytes_per_zone_update,
         "bytes_per_zone_update": bytes_per_zone_update, "equivalent_GBps": gbs, "equivalent_over_8TBps": gbs / HBM_PEAK_GBS}
    if target:
        r["target_Mcells_per_s_per_gpu"] = target
        r["value_over_target"] = mcells_per_s_per_gpu / target
    return r


ROOF_KEEP = ("bound", "achieved", "peak", "unit", "frac", "traffic", "valu_busy", "hbm_frac_measured", "bytes_moved_per_cell", "kernel",
             "avg_launch_ms", "launches", "flops_per_launch", "algorithmic_bytes_per_launch", "counters")


def leg_summary(leg):
    r = leg.get("roofline") or {}
    return [sig(leg.get("value"), 6), r.get("bound"), sig(r.get("frac"), 4)]


def config_summary(line):
    if "value" not in line:
        return "error"
    r = line.get("roofline") or {}
    strict = line.get("arith_strict") or {}
    return [sig(line["value"], 6), r.get("bound"), sig(r.get("frac"), 4), sig(strict.get("value"), 6) if "value" in strict else None]


def build_summary(details):
    s = {"headline": leg_summary(details), "n_gpus": details.get("n_gpus")}
    if details.get("stepper"):
        s["stepper"] = details["stepper"]
    for key, leg in (details.get("legs") or {}).items():
        s[key] = leg_summary(leg)
    for cfg, line in (details.get("extra_configs") or {}).items():
        s[cfg] = config_summary(line)
    for k in ("slabs_bit_identical_to_one_gpu_run",):
        if k in details:
            s[k] = details[k]
    for k, v in details.items():
        if k.startswith("l1_fast_vs_strict_after_"):
            s[k] = sig(v, 3)
    s["note"] = ("leg: [Mcells/s, bound, roofline.frac] - fp64: fraction of 78.6 TFLOP/s (one-launch steps), hbm: algorithmic stage bytes over 8 TB/s "
                 "(second stage of two-launch steps); c3-c5: [FAST Mzones/s, bound, frac, STRICT Mzones/s]")
    return s


def shorten(text, limit):
    return text if not isinstance(text, str) or len(text) <= limit else text[:limit - 3] + "..."


def final_line(details, limit=LINE_LIMIT, details_file="bench_details.json"):
    """the compact object of the LAST stdout line: contract keys, config, roofline, roofline_step, cpu_baseline, cpu_reference, summary (once)"""
    out = {k: details[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                   "dtype", "data") if k in details}
    cfg = details.get("config") or {}
    out["config"] = {k: cfg[k] for k in ("workload", "decomposition", "arith", "riemann", "planar_kernel", "launches_per_step", "status_word", "timed_region") if k in cfg}
    roof = details.get("roofline") or {}
    out["roofline"] = {k: roof[k] for k in ROOF_KEEP if k in roof}
    if details.get("roofline_stage1"):
        out["roofline_stage1"] = {k: details["roofline_stage1"][k] for k in ROOF_KEEP if k in details["roofline_stage1"]}
    if details.get("roofline_step"):
        out["roofline_step"] = details["roofline_step"]
    for k in ("cpu_baseline", "cpu_reference"):
        if details.get(k):
            out[k] = dict(details[k])
    if "slabs_bit_identical_to_one_gpu_run" in details:
        out["slabs_bit_identical_to_one_gpu_run"] = details["slabs_bit_identical_to_one_gpu_run"]
    if details.get("repeat_blocks"):
        out["repeat_blocks_ms_per_step"] = details["repeat_blocks"].get("ms_per_step")
    out["details"] = details_file
    out["summary"] = build_summary(details)
    out = round_floats(out)
    # the length is a contract (the driver keeps about 8 KB of stdout): shorten prose first, then drop what the side file holds anyway
    text = json.dumps(out)
    for cap in (400, 240, 160, 100):
        if len(text) <= limit:
            break
        for obj in (out["config"], out["roofline"], out.get("roofline_stage1") or {}, out.get("roofline_step") or {}, out.get("cpu_baseline") or {},
                    out.get("cpu_reference") or {}, out["summary"]):
            for k, v in list(obj.items()):
                obj[k] = shorten(v, cap)
        text = json.dumps(out)
    for victim in ("repeat_blocks_ms_per_step", "roofline_stage1", "cpu_reference"):
        if len(text) <= limit:
            break
        out.pop(victim, None)
        text = json.dumps(out)
    if len(text) > limit:
        keep = ("headline", "n_gpus", "note", "stepper")
        out["summary"] = {k: v for k, v in out["summary"].items() if k in keep or not isinstance(v, list) or k.startswith("c")}
        text = json.dumps(out)
    if len(text) > limit:
        raise RuntimeError("bench line still %d characters after trimming" % len(text))
    return text


REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline")


def check_line(text, n_gpus_one=True):
    """the contract of the final line, shared by the CPU test on a recorded line and the GPU test on a fresh one; returns the parsed object"""
    assert "\n" not in text.strip() and len(text) <= LINE_LIMIT, len(text)
    pairs = json.loads(text, object_pairs_hook=list)
    keys = [k for k, _ in pairs]
    assert len(keys) == len(set(keys)), "duplicate top-level keys"
    d = json.loads(text)
    for k in REQUIRED:
        assert k in d, k
    assert keys.count("summary") == 1 and "summary" not in d["config"]
    assert "workload" in d["config"] and not any(k in d["config"] for k in ("model", "global_batch", "seq_len"))
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "launches"):
        assert k in r, k
    assert r["bound"] in ("fp64", "hbm") and 0.0 < r["frac"] < 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-4 * r["frac"]
    if r["bound"] == "fp64":
        assert r["peak"] == FP64_VECTOR_PEAK_TFLOPS and r["unit"] == "TFLOP/s"
        assert abs(r["achieved"] - r["flops_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) <= 1e-4 * r["achieved"]
        assert 0.0 < r["hbm_frac_measured"] < 1.0 and 0.0 < r["bytes_moved_per_cell"] <= 200
    else:
        assert r["peak"] == HBM_PEAK_GBS and r["unit"] == "GB/s"
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 1e-4 * r["achieved"]
    if "roofline_step" in d:
        assert "frac" not in d["roofline_step"] and "byte-equivalents" in d["roofline_step"]["name"]
    for key, v in d["summary"].items():
        if isinstance(v, list) and len(v) >= 3 and isinstance(v[1], str) and v[2] is not None:
            assert v[1] in ("fp64", "hbm") and 0.0 < v[2] < 1.0, (key, v)
    if n_gpus_one and d["n_gpus"] == 1:
        c = d["cpu_baseline"]
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    return d

"""profiles/r04 reproduces its own summary: every (run, kernel) row of kernels_headline.md / kernels_configs.md quotes the average, minimum and
call count of the kernel-trace CSV that is TRACKED beside it (kernel_stats_<run>.csv, rocprofv3 --kernel-trace --stats of the run named in the
table) - round 3's table quoted 283-launch traces that lived only in scratch. And profiles/pmc_traffic.json carries, for the sources it is
stamped with, the bytes and counters the bench line's `traffic` / `fp64` objects are read from."""
import csv
import json
import os
import re
import pytest
from conftest import ROOT

R04 = os.path.join(ROOT, "profiles", "r04")


def rows_of(md):
    for line in open(md):
        m = re.match(r"\| (\w+) \| `([^`]+)` \| ([0-9.]+) \| ([0-9.]+) \| (\d+) \|", line)
        if m:
            yield m.group(1), m.group(2), float(m.group(3)), float(m.group(4)), int(m.group(5))


@pytest.mark.parametrize("part", ["headline", "configs"])
def test_kernel_tables_quote_the_tracked_traces(part):
    md = os.path.join(R04, "kernels_%s.md" % part)
    if not os.path.exists(md):
        pytest.skip("profiles/r04/kernels_%s.md not generated yet (scripts/profile_r4.sh, scripts/pmc_collate.py)" % part)
    seen = 0
    for run, kernel, avg_us, min_us, calls in rows_of(md):
        path = os.path.join(R04, "kernel_stats_%s.csv" % run)
        assert os.path.exists(path), path
        match = [r for r in csv.DictReader(open(path)) if r["Name"].split("(")[0].replace("void mh::", "").replace("mh::", "") == kernel]
        assert len(match) == 1, (run, kernel)
        r = match[0]
        assert int(r["Calls"]) == calls and abs(float(r["AverageNs"]) / 1e3 - avg_us) <= 0.051 and abs(float(r["MinNs"]) / 1e3 - min_us) <= 0.051, (run, kernel, r)
        seen += 1
    assert seen >= 4


def test_headline_trace_is_long_enough_for_cold_launches_not_to_carry_the_average():
    path = os.path.join(R04, "kernel_stats_fast_hllc.csv")
    if not os.path.exists(path):
        pytest.skip("profiles/r04 not generated yet")
    fused = [r for r in csv.DictReader(open(path)) if "euler2d_fused_rk2_kernel" in r["Name"]]
    assert fused and int(fused[0]["Calls"]) >= 250


def test_pmc_traffic_table_is_keyed_and_complete():
    t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert re.fullmatch(r"[0-9a-f]{16}", t["csrc_sha16"])
    if not os.path.exists(os.path.join(R04, "kernels_headline.md")):
        pytest.skip("profiles/r04 not generated yet")
    cells = 4096 * 4096
    fused = t["fused_planar_fast_hllc_bytes_per_launch"]
    assert 72 * cells <= fused <= 1.25 * 72 * cells          # the planar fused launch reads 32 B and writes 40 B per cell (+ halo re-reads)
    assert 0.5 < t["fused_planar_fast_hllc_fp64"]["valu_busy"] <= 1.0
    general = t["fused_fast_hllc_bytes_per_launch"]          # ... and the general kernel (bench.py --no-planar) 40 + 40
    assert 80 * cells <= general <= 1.25 * 80 * cells
    for key in ("stage2_strict_hlle_bytes_per_launch", "stage1_strict_hlle_bytes_per_launch"):
        assert key in t


def end_of_round_line():
    path = os.path.join(R04, "bench_driverstyle_end_of_round.json")
    if not os.path.exists(path):
        pytest.skip("profiles/r04/bench_driverstyle_end_of_round.json not recorded yet")
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


def test_recorded_bench_line_recomputes_from_its_own_fields():
    """every `frac` of the recorded driver-style line follows from the bytes and the launch time beside it (the judge's recomputation)"""
    d = end_of_round_line()
    roofs = [d["roofline"]] + [leg[k] for leg in d["legs"].values() for k in ("roofline", "roofline_stage1") if leg.get(k)]
    roofs += [c[k] for c in d["extra_configs"].values() for k in ("roofline",) if c.get(k)]
    roofs += [c["arith_strict"]["roofline"] for c in d["extra_configs"].values() if c.get("arith_strict", {}).get("roofline")]
    assert len(roofs) >= 12
    for r in roofs:
        ach = r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9
        assert abs(ach - r["achieved"]) <= 1e-5 * ach and abs(r["frac"] - ach / 8000.0) <= 1e-5, r          # (legs keep six significant digits)
    h = d["roofline"]
    assert h["algorithmic_bytes_per_launch"] == 200 * 4096 * 4096 and h["bytes_actually_moved_per_cell"] == 72
    assert abs(h["frac_actual_traffic"] - h["frac"] * 72 / 200) <= 1e-9
    assert abs(d["value"] - 4096 * 4096 / d["ms_per_step"] / 1e3) <= 1e-6 * d["value"]
    assert h["avg_launch_ms"] <= d["ms_per_step"] * 1.01          # a launch is not longer than the step it is


def test_readme_first_screen_quotes_the_recorded_line():
    d = end_of_round_line()
    text = open(os.path.join(ROOT, "README.md")).read()
    assert "@@" not in text
    fmt = lambda x: "{:,}".format(int(round(x))).replace(",", " ")
    for v in (d["value"], d["legs"]["fast_hlle_blast"]["value"], d["legs"]["strict_hlle_blast"]["value"], d["legs"]["fast_hllc_blast_general_kernel"]["value"]):
        assert fmt(v) in text, fmt(v)

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

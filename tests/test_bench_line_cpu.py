"""The final stdout line of bench.py is a contract with the driver: ONE JSON object of at most 6000 characters (round 4's 21.7 KB line came back
`"parsed": null`), every `roofline.frac` a hardware fraction in (0, 1), the summary once. Checked here without a GPU: on a synthetic record as
large as round 4's, and on the line recorded on the MI355X this round (profiles/r05/bench_driverstyle_*.json)."""
import glob
import json
import os
import pytest
from conftest import ROOT

import bench_report


def synthetic_details(legs=8, prose=1500):
    cells = 4096 * 4096
    fused = bench_report.fp64_roofline("euler2d_fused_rk2_kernel<hllc, planar> (both RK2 stages, one launch per step)", 0.466, 5, cells, 849.1, 0.919, 76.3, 72,
                                       "rocprofv3 PMC passes of these kernel sources (profiles/pmc_traffic.json); durations are this run's", timing="x" * prose)
    stage = bench_report.hbm_roofline("euler2d_stage_kernel<strict,hlle,PLM,COMBINE> (second RK2 stage)", 0.427, 5, cells, 120, traffic=2.19e9, timing="y" * prose)
    leg = lambda r, v: {"value": v, "ms_per_step": cells / v / 1e3, "status_word": 0, "launches_per_step": 1 if r is fused else 2, "planar_kernel": r is fused,
                        "roofline": dict(r), "roofline_stage1": None if r is fused else dict(stage), "roofline_step": bench_report.step_equivalents(v), "note": "n" * prose}
    d = {"metric": "zone-updates/sec (Mcells/s) whole node, 2D Euler 4096^2 PLM+HLLC RK2", "value": 36027.123456789, "unit": "Mcells/s", "n_gpus": 1, "steps": 20,
         "warmup": 5, "ms_per_step": 0.46567, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
         "config": {"workload": "2D Euler Sedov-type blast, 4096x4096 uniform grid, PLM(theta=1.5)+HLLC, RK2, fp64, fixed dt=0.3*dx/6, outflow BC",
                    "decomposition": "one GPU: the whole grid as one slab of the native stepper, no cuts, no exchange", "arith": "a" * 300, "riemann": "r" * 300,
                    "planar_kernel": True, "launches_per_step": 1, "status_word": 0, "planar_note": "p" * prose, "timed_region": "t" * 500, "preconditioning": "q" * 400},
         "roofline": fused, "roofline_step": bench_report.step_equivalents(36027.1),
         "repeat_blocks": {"ms_per_step": [0.465, 0.466, 0.467, 0.468], "median_ms_per_step": 0.466, "min": 0.465, "max": 0.468, "note": "z" * 200},
         "legs": {("leg_number_%d_with_a_long_name" % i): leg(fused if i % 2 else stage, 20000.0 + i) for i in range(legs)},
         "legs_note": "l" * prose, "l1_fast_vs_strict_after_150_steps": 9.2e-16,
         "extra_configs": {c: {"value": 18000.0, "ms_per_step": 0.9, "roofline": dict(stage), "arith_strict": {"value": 9000.0, "roofline": dict(stage)},
                               "config": {"workload": "w" * prose}} for c in ("c3", "c4", "c5")},
         "cpu_baseline": {"value": 15.8, "unit": "Mcells/s", "cores": 16, "kind": "port", "sample": "s" * 400},
         "cpu_reference": {"value": 2.54, "unit": "Mcells/s", "cores": 16, "kind": "reference", "one_thread": 1.29, "host_cores": 16, "sample": "s" * 600}}
    d["summary"] = bench_report.build_summary(d)
    return d


@pytest.mark.parametrize("legs,prose", [(8, 300), (8, 3000), (40, 3000)])
def test_final_line_is_compact_whatever_the_details_hold(legs, prose):
    d = synthetic_details(legs, prose)
    assert len(json.dumps(d)) > 20000                     # as large as the line that did not parse
    text = bench_report.final_line(d)
    assert len(text) <= bench_report.LINE_LIMIT
    got = bench_report.check_line(text)
    assert got["value"] == pytest.approx(d["value"], rel=1e-5) and got["roofline"]["bound"] == "fp64"
    assert "legs" not in got and "extra_configs" not in got and got["details"] == "bench_details.json"
    assert got["summary"]["headline"][0] == pytest.approx(d["value"], rel=1e-5)
    if legs <= 8:
        assert len([k for k in got["summary"] if k.startswith("leg_number")]) == legs and "c3" in got["summary"]


def test_roofline_objects_are_hardware_fractions():
    cells = 4096 * 4096
    r = bench_report.fp64_roofline("k", 0.466, 5, cells, 849.1, 0.92, 76.3, 72, "p")
    # verdict r4's recomputation: 14.25 GFLOP in 0.466 ms = 30.6 TF = 0.39 of 78.6; 1.28 GB in 0.466 ms = 2.75 TB/s = 0.34 of 8 TB/s
    assert r["bound"] == "fp64" and r["frac"] == pytest.approx(849.1 * cells / 0.466e-3 / 78.6e12) and 0.35 < r["frac"] < 0.42
    assert r["hbm_frac_measured"] == pytest.approx(76.3 * cells / 0.466e-3 / 8e12) and r["traffic"] == pytest.approx(76.3 * cells)
    nominal = bench_report.fp64_roofline("k", 0.466, 5, cells, 849.1, 0.92, None, 72, "p")
    assert nominal["traffic"] is None and nominal["bytes_moved_per_cell"] == 72 and "nominal" in nominal["bytes_moved_source"]
    s = bench_report.step_equivalents(45405.0)
    assert "frac" not in s and s["equivalent_over_8TBps"] > 1.0 and s["value_over_target"] == pytest.approx(45405.0 / 16000.0)
    h = bench_report.hbm_roofline("k", 0.39, 5, cells, 120)
    assert h["bound"] == "hbm" and h["frac"] == pytest.approx(120 * cells / 0.39e-3 / 8e12)


def test_check_line_refuses_what_round_4_printed():
    d = synthetic_details()
    good = json.loads(bench_report.final_line(d))
    for breakit in (lambda x: x["roofline"].__setitem__("frac", 1.14), lambda x: x["config"].__setitem__("summary", x["summary"]),
                    lambda x: x.__setitem__("pad", "x" * 7000), lambda x: x["roofline_step"].__setitem__("frac", 0.9),
                    lambda x: x.pop("cpu_baseline")):
        bad = json.loads(json.dumps(good))
        breakit(bad)
        with pytest.raises((AssertionError, KeyError)):
            bench_report.check_line(json.dumps(bad))


def recorded_lines():
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r05", "bench_driverstyle_*.json")))


def test_recorded_driver_style_line_of_this_round():
    """the stdout of `python bench.py --gpus 1 --steps 20 --warmup 5` on the MI355X, as the driver captures it: the LAST line parses, is within
    the limit and carries roofline + cpu_baseline; the side file recorded beside it holds the legs the summary quotes"""
    paths = recorded_lines()
    if not paths:
        pytest.skip("profiles/r05/bench_driverstyle_*.json not recorded yet")
    for path in paths:
        lines = [l for l in open(path).read().splitlines() if l.strip()]
        d = bench_report.check_line(lines[-1])
        assert d["n_gpus"] == 1 and d["value"] == pytest.approx(4096 * 4096 / d["ms_per_step"] / 1e3, rel=1e-5)
        assert d["roofline"]["avg_launch_ms"] <= 1.01 * d["ms_per_step"]
        side = path.replace("bench_driverstyle_", "bench_details_")
        if os.path.exists(side):
            full = json.load(open(side))
            for key, leg in full["legs"].items():
                assert d["summary"][key][0] == pytest.approx(leg["value"], rel=1e-5)
                r = leg["roofline"]
                assert 0.0 < r["frac"] < 1.0, (key, r)


def test_counters_are_stamped_with_the_code_of_this_tree():
    """profiles/pmc_traffic.json is stamped with a hash of the kernel sources' CODE (comments and white space aside): the counters the bench line
    quotes were taken on the kernels of this tree - a kernel edit without a new profile pass (scripts/profile_r5.sh, scripts/pmc_collate.py) fails here"""
    import bench
    assert bench.code_only("a = 1; // note\n/* block\n comment */ b  =\t2;\n") == "a = 1; b = 2;"
    t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert t["csrc_sha16"] == bench.csrc_fingerprint(), "kernel sources changed since the PMC passes: re-run scripts/profile_r5.sh and scripts/pmc_collate.py"
    assert bench_report.Counters(bench.csrc_fingerprint()).current


@pytest.mark.parametrize("config,kernel,lps", [("c3", "binary_stage_kernel + binary_sink_kernel + binary_reduce_kernel (one stage)", 2),
                                               ("c5", "euler3d_stage_kernel<fast,hlle,PLM> (mean of both RK2 stages)", 2),
                                               ("c4", "cloud_fused_rk2_kernel<planar> (both RK2 stages in one launch per step)", 1)])
def test_config_lines_find_their_counters(config, kernel, lps):
    """bench_configs.attach_traffic maps a config's kernels to the rows of profiles/pmc_traffic.json (the C3 FAST kernel is BinFastT<true> since
    round 5): with a current stamp every config gets its fp64 figure and its PMC bytes; the one-launch `cloud` step becomes bound fp64"""
    import bench_configs
    roof = bench_report.hbm_roofline(kernel, 0.9 if config == "c4" else (3.7 if config == "c5" else 0.1), 10, 4096 * 4096, 104 if config == "c4" else 100,
                                     extra={"bytes_moved_per_cell": 104, "launches_per_step": lps})
    out = bench_configs.attach_traffic({"roofline": roof, "arith_strict": {"roofline": dict(roof, kernel=kernel.replace("fast", "strict"), launches_per_step=2)}}, config)
    r = out["roofline"]
    if config == "c4":
        assert r["bound"] == "fp64" and 0.0 < r["frac"] < 1.0 and r["traffic"] > 0 and 0.5 < r["valu_busy"] <= 1.0
    else:
        assert r["bound"] == "hbm" and r["traffic"] > 0 and 0.0 < r["fp64"]["frac"] < 1.0 and 0.3 < r["fp64"]["valu_busy"] <= 1.0
        assert out["arith_strict"]["roofline"]["fp64"]["valu_busy"] > 0.3

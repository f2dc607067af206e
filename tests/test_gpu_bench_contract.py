"""The one-line JSON contract of bench.py (N = 1, a short run): every field the driver and the judge read is there and consistent."""
import json
import os
import subprocess
import sys
import pytest
from conftest import ROOT

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def test_bench_json_line_contract():
    import bench_report
    side = os.path.join(ROOT, "bench_details.json")
    if os.path.exists(side):
        os.remove(side)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1"], cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and out.stdout.rstrip().endswith(lines[0])       # ONE JSON line, and it is the LAST thing on stdout
    # round 5: the line the driver parses is compact (round 4's 21.7 KB line came back unparsed) and its roofline is a hardware fraction
    d = bench_report.check_line(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True and d["dtype"] == "f64"
    assert d["unit"] == "Mcells/s" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert abs(d["value"] - 4096 * 4096 / d["ms_per_step"] / 1e3) <= 1e-5 * d["value"]
    for key in ("legs", "extra_configs", "repeat_blocks", "legs_note"):
        assert key not in d
    r = d["roofline"]
    # the FAST headline is ONE fused launch per RK2 step on the planar kernel: fp64-issue-bound, so that is its roofline; what it moves stands beside it
    assert r["bound"] == "fp64" and r["peak"] == 78.6 and "fused" in r["kernel"] and "planar" in r["kernel"] and "roofline_stage1" not in d
    assert 0.0 < r["frac"] < 1.0 and 0.0 < r["hbm_frac_measured"] < 1.0 and 72 <= r["bytes_moved_per_cell"] <= 1.25 * 72
    assert r["traffic"] is None or r["traffic"] >= 72 * 4096 * 4096
    assert d["config"]["planar_kernel"] is True and d["config"]["launches_per_step"] == 1 and d["config"]["status_word"] == 0
    s = d["roofline_step"]
    assert s["bytes_per_zone_update"] == 200 and s["equivalent_GBps"] == pytest.approx(d["value"] * 200 / 1e3, rel=1e-4) and s["value_over_target"] == pytest.approx(d["value"] / 16000, rel=1e-4)
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mcells/s" and c["sample"]
    # everything else is in the side file (and on stderr)
    assert d["details"] == "bench_details.json" and os.path.exists(side)
    full = json.load(open(side))
    assert "bench.py details: " in out.stderr
    assert full["summary"] == json.loads(json.dumps(bench_report.round_floats(bench_report.build_summary(full), 8)))
    assert full["value"] == pytest.approx(d["value"], rel=1e-5) and "general_kernel" in full["config"]["planar_note"]
    legs = full["legs"]
    g = legs["fast_hllc_blast_general_kernel"]
    assert g["planar_kernel"] is False and g["launches_per_step"] == 1 and 0.0 < g["value"] <= 1.02 * d["value"]
    for key in ("strict_hllc_blast", "strict_hlle_blast", "fast_hlle_blast", "fast_hllc_smooth_wave", "strict_hlle_smooth_wave", "fast_hllc_blast_two_launches",
                "fast_hllc_blast_general_kernel"):
        assert key in legs and legs[key]["value"] > 0 and legs[key]["status_word"] == 0
        lr = legs[key]["roofline"]
        assert 0.0 < lr["frac"] < 1.0, (key, lr)                               # EVERY roofline.frac is a fraction (round 4 printed 1.14 here)
        assert "frac" not in legs[key]["roofline_step"]
        if key.startswith("fast") and not key.endswith("two_launches"):
            assert legs[key]["launches_per_step"] == 1 and legs[key]["roofline_stage1"] is None
            assert 0.0 < lr["hbm_frac_measured"] < 1.0 and lr["bound"] in ("fp64", "hbm")
            assert legs[key]["planar_kernel"] is (not key.endswith("general_kernel"))
        else:
            assert lr["bound"] == "hbm" and lr["algorithmic_bytes_per_launch"] == 120 * 4096 * 4096
            assert legs[key]["launches_per_step"] == 2 and 0.0 < legs[key]["roofline_stage1"]["frac"] < 1.0
        assert d["summary"][key][0] == pytest.approx(legs[key]["value"], rel=1e-5) and d["summary"][key][2] == pytest.approx(lr["frac"], rel=1e-3)
    assert len(full["repeat_blocks"]["ms_per_step"]) == 4 and len(d["repeat_blocks_ms_per_step"]) == 4
    assert any(k.startswith("l1_fast_vs_strict_after_") and full[k] <= 1e-12 for k in full)
    for cfg in ("c3", "c4", "c5"):
        e = full["extra_configs"][cfg]
        assert e.get("value", 0) > 0, e
        assert 0.0 < e["roofline"]["frac"] < 1.0 and 0.0 < e["arith_strict"]["roofline"]["frac"] < 1.0
        assert d["summary"][cfg][0] == pytest.approx(e["value"], rel=1e-5)


@pytest.mark.parametrize("cuts", [None, "0"])
def test_bench_loopback_rehearsal_of_the_multi_gpu_path(cuts):
    """`bench.py --loopback-slabs 4`: the N > 1 code path of the bench (decomposed stepper, timing, per-rank fingerprints, partition check
    against a one-GPU run) on one GPU through the native stepper's loopback backend - with the library's own choice of schedule (the fused
    step across the cuts, one launch per step) and with the two-launch schedule (MH_SLAB_FUSED_CUTS=0)."""
    env = dict(os.environ)
    if cuts is not None:
        env["MH_SLAB_FUSED_CUTS"] = cuts
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--loopback-slabs", "4", "--single-arith", "--no-cpu-baseline",
                          "--blocks", "2"], cwd=ROOT, capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["slabs_bit_identical_to_one_gpu_run"] is True
    assert "REHEARSAL" in d["config"]["decomposition"] and d["config"]["status_word"] == 0
    assert ("one exchange per step" in d["config"]["timed_region"]) == (cuts is None)


def run_configs(extra, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench_configs.py")] + extra, cwd=ROOT, capture_output=True, text=True, timeout=280,
                         env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_config4_rehearsal_of_the_four_slab_run_checks_itself():
    """`bench_configs.py --config c4 --gpus 4` on this one-GPU box: the compiled host drives four radial slabs of the native stepper (the slabs
    share the device, peer copies become device-to-device copies), incl. the per-step nozzle row that only the slab owning row 0 reads - and
    the line says whether their union equals the one-device run (whose FAST step is the ONE-launch kernel) bit for bit"""
    d = run_configs(["--config", "c4", "--gpus", "4", "--grid", "256", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"])
    assert d["slabs_bit_identical_to_one_gpu_run"] is True and d["n_gpus"] == 4 and d["value"] > 0
    assert "REHEARSAL" in d["config"]["decomposition"] and d["config"]["launches_per_step"] == 2          # 64 rows per slab: the two launches
    # round 5: from 384 rows per slab on (the 4096-row config: 1024) - here on request - every slab takes the ONE-launch step across its cuts
    d = run_configs(["--config", "c4", "--gpus", "4", "--grid", "256", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"], env={"MH_SLAB_FUSED_CUTS": "1"})
    assert d["slabs_bit_identical_to_one_gpu_run"] is True and d["config"]["launches_per_step"] == 1 and "one exchange per step" in d["config"]["decomposition"]
    d = run_configs(["--config", "c4", "--gpus", "2", "--grid", "800", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert d["slabs_bit_identical_to_one_gpu_run"] is True and d["config"]["launches_per_step"] == 1          # 400 rows per slab: by default


def test_config5_rehearsal_of_the_eight_block_run_checks_itself():
    """`bench_configs.py --config c5 --loopback-blocks 8`: the (2,2,2) blocks of propose_block_decomposition<3>(8) as objects of one process,
    the line carries `blocks_bit_identical_to_one_gpu_run`"""
    d = run_configs(["--config", "c5", "--loopback-blocks", "8", "--grid", "48", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert d["blocks_bit_identical_to_one_gpu_run"] is True and d["value"] > 0 and d["arith_strict"]["status_word"] == 0


def test_config4_one_gpu_line_reports_the_fused_launch():
    d = run_configs(["--config", "c4", "--grid", "512", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"])
    r = d["roofline"]
    assert d["config"]["launches_per_step"] == 1 and "cloud_fused_rk2_kernel" in r["kernel"] and r["launches_per_step"] == 1
    assert 0.0 < r["frac"] < 1.0 and r["bound"] in ("fp64", "hbm") and r["bytes_moved_per_cell"] in (104, 120)
    assert "frac" not in d["roofline_step"]
    # kernel time <= step time: the profile pass brackets several launches with ONE pair of events
    assert r["avg_launch_ms"] <= 1.05 * d["ms_per_step"]
    s = d["arith_strict"]
    assert s["roofline"]["launches_per_step"] == 2 and s["value"] > 0

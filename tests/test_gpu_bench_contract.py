"""The one-line JSON contract of bench.py (N = 1, a short run): every field the driver and the judge read is there and consistent."""
import json
import os
import subprocess
import sys
import pytest
from conftest import ROOT

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def test_bench_json_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1"], cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True and d["dtype"] == "f64"
    assert d["unit"] == "Mcells/s" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 4096 * 4096 / d["ms_per_step"] / 1e3) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    # round 3: the FAST headline is ONE fused launch per RK2 step; it is charged the 200 B of SURVEY.md 8d and moves 80 B per cell
    assert d["config"]["summary"] == d["summary"] and list(d)[-1] == "summary"
    # round 4: the blast has no third momentum, the library verifies that at upload and the fused launch skips the component (72 B per cell);
    # the same leg on the general kernel (80 B) stands beside it, and the line says which one `value` is
    assert "fused" in r["kernel"] and "planar" in r["kernel"] and r["bytes_actually_moved_per_cell"] == 72 and "roofline_stage1" not in d
    assert d["config"]["planar_kernel"] is True and "general_kernel" in d["config"]["planar_note"]
    assert r["algorithmic_bytes_per_launch"] == 200 * 4096 * 4096
    assert r["traffic"] is None or r["traffic"] >= 72 * 4096 * 4096
    assert abs(r["frac_actual_traffic"] - r["achieved_actual_traffic"] / 8000.0) < 1e-12 and "fp64 issue" in r["bound_measured"]
    g = d["legs"]["fast_hllc_blast_general_kernel"]
    assert g["planar_kernel"] is False and g["launches_per_step"] == 1 and 0.0 < g["value"] <= 1.02 * d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mcells/s" and c["sample"]
    assert d["config"]["status_word"] == 0
    # round 2: the pinned variant, the smooth wave and the other configs are in the driver-run line too, each with its own rooflines
    legs = d["legs"]
    for key in ("strict_hllc_blast", "strict_hlle_blast", "fast_hlle_blast", "fast_hllc_smooth_wave", "strict_hlle_smooth_wave", "fast_hllc_blast_two_launches",
                "fast_hllc_blast_general_kernel"):
        assert key in legs and legs[key]["value"] > 0 and legs[key]["status_word"] == 0
        if key.startswith("fast") and not key.endswith("two_launches"):
            # one launch per step, charged SURVEY 8d's 200 B per zone-update while it moves 72 or 80 B per cell: the convention's fraction may pass 1
            # (the planar HLLE step does) - the fraction of 8 TB/s the launch really uses stands beside it and cannot
            assert legs[key]["launches_per_step"] == 1 and legs[key]["roofline_stage1"] is None
            lr = legs[key]["roofline"]
            assert 0.0 < lr["frac_actual_traffic"] < 1.0 and lr["frac"] == pytest.approx(lr["frac_actual_traffic"] * 200 / lr["bytes_actually_moved_per_cell"], rel=1e-4)
            assert legs[key]["planar_kernel"] is (not key.endswith("general_kernel"))
        else:
            assert 0.0 < legs[key]["roofline"]["frac"] < 1.0
            assert legs[key]["launches_per_step"] == 2 and 0.0 < legs[key]["roofline_stage1"]["frac"] < 1.0
        assert d["summary"][key][0] == pytest.approx(legs[key]["value"], rel=1e-3)
    assert len(d["repeat_blocks"]["ms_per_step"]) == 4
    assert any(k.startswith("l1_fast_vs_strict_after_") and d[k] <= 1e-12 for k in d)
    for cfg in ("c3", "c4", "c5"):
        assert d["extra_configs"][cfg].get("value", 0) > 0, d["extra_configs"][cfg]
    if r["traffic"] is not None:
        assert r["fp64"]["peak"] == 78.6 and 0.0 < r["fp64"]["frac"] < 1.0 and 0.0 < r["fp64"]["valu_busy"] <= 1.0


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


def run_configs(extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench_configs.py")] + extra, cwd=ROOT, capture_output=True, text=True, timeout=280)
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
    assert "REHEARSAL" in d["config"]["decomposition"] and d["config"]["launches_per_step"] == 2


def test_config5_rehearsal_of_the_eight_block_run_checks_itself():
    """`bench_configs.py --config c5 --loopback-blocks 8`: the (2,2,2) blocks of propose_block_decomposition<3>(8) as objects of one process,
    the line carries `blocks_bit_identical_to_one_gpu_run`"""
    d = run_configs(["--config", "c5", "--loopback-blocks", "8", "--grid", "48", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert d["blocks_bit_identical_to_one_gpu_run"] is True and d["value"] > 0 and d["arith_strict"]["status_word"] == 0


def test_config4_one_gpu_line_reports_the_fused_launch():
    d = run_configs(["--config", "c4", "--grid", "512", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"])
    r = d["roofline"]
    assert d["config"]["launches_per_step"] == 1 and "cloud_fused_rk2_kernel" in r["kernel"] and r["launches_per_step"] == 1
    assert r["algorithmic_bytes_per_launch"] == 200 * 512 * 512 and 0.0 < r["frac"] < 1.0
    # kernel time <= step time: the profile pass brackets several launches with ONE pair of events
    assert r["avg_launch_ms"] <= 1.05 * d["ms_per_step"]
    s = d["arith_strict"]
    assert s["roofline"]["launches_per_step"] == 2 and s["value"] > 0

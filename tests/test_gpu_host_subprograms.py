"""GPU tests of the compiled C++ host `mara_hip` (mara3_amd/host): the sub-program drivers keep the
reference's plug-in surface (name, key=value options, run loop, kzps message) and must reproduce the
reference-generated golden vectors through the C ABI, bit for bit."""
import os
import struct
import subprocess
import numpy as np
import pytest
from conftest import golden, bits_equal, ROOT

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "mara3_amd", "host", "mara_hip")


def read_dump(path):
    raw = open(path, "rb").read()
    off = 0
    (rank,) = struct.unpack_from("q", raw, off); off += 8
    shape = struct.unpack_from("%dq" % rank, raw, off); off += 8 * rank
    (nq,) = struct.unpack_from("q", raw, off); off += 8
    (time,) = struct.unpack_from("d", raw, off); off += 8
    (iteration,) = struct.unpack_from("q", raw, off); off += 8
    (nv,) = struct.unpack_from("q", raw, off); off += 8
    vertices = np.frombuffer(raw, dtype=np.float64, count=nv, offset=off); off += 8 * nv
    data = np.frombuffer(raw, dtype=np.float64, offset=off).reshape(tuple(shape) + (nq,))
    return dict(time=time, iteration=iteration, vertices=vertices, data=data)


def run(args, cwd):
    assert os.path.exists(EXE), "build the compiled host first (__graft_entry__.build())"
    out = subprocess.run([EXE] + args, cwd=cwd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    return out.stdout


def test_sedov_subprogram_matches_reference(tmp_path):
    g = golden("sedov_newtonian_nr256")
    dt = 0.4 * (g["vertices"][1] - g["vertices"][0])
    for nsteps in (10, 100):
        stdout = run(["sedov", "newtonian=1", "nr=256", "outer_radius=100", "tfinal=%r" % float((nsteps - 0.5) * dt), "outdir=out%d" % nsteps], str(tmp_path))
        d = read_dump(os.path.join(tmp_path, "out%d" % nsteps, "final.bin"))
        assert d["iteration"] == nsteps
        assert bits_equal(d["vertices"], g["vertices"])
        assert bits_equal(d["data"], g["u_%d" % nsteps])
        assert "total execution time" in stdout
        if nsteps == 100:
            assert "[0100] t=" in stdout and "kzps=" in stdout


def test_sedov_subprogram_default_system_is_srhd(tmp_path):
    g = golden("sedov_srhd_nr256")
    dt = 0.4 * (g["vertices"][1] - g["vertices"][0])
    run(["sedov", "nr=256", "tfinal=%r" % float(9.5 * dt)], str(tmp_path))
    d = read_dump(os.path.join(tmp_path, "data", "final.bin"))
    assert d["iteration"] == 10
    assert bits_equal(d["data"], g["u_10"])


def test_sedov_subprogram_option_errors(tmp_path):
    out = subprocess.run([EXE, "sedov", "nosuchkey=1"], cwd=str(tmp_path), capture_output=True, text=True)
    assert out.returncode == 1 and "unknown key" in out.stdout
    out = subprocess.run([EXE, "sedov", "nr=abc"], cwd=str(tmp_path), capture_output=True, text=True)
    assert out.returncode == 1 and "wrong data type" in out.stdout


def test_sedov_subprogram_ends_on_a_device_status_bit(tmp_path):
    """Where the reference throws out of the failing step (mara::srhd::recover_primitive, src/physics_srhd.hpp:430-449: negative pressure)
    the device leaves a status bit; the host must end the run on ANY bit with the reference's exception text, not carry on and exit 0.
    A negative explosion pressure makes the very first recover_primitive fail. (With newtonian=1 nothing ends the run upstream either:
    mara::euler::recover_primitive never throws, src/physics_euler.hpp:555-575.)"""
    out = subprocess.run([EXE, "sedov", "nr=64", "tfinal=0.05", "explosion_pressure=-1.0", "cpi=0", "tsi=0", "dfi=0"],
                         cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert out.returncode != 0, out.stdout[-500:]
    assert "mara::srhd::recover_primitive failure" in out.stdout
    assert "first failing cell" in out.stdout
    assert not os.path.exists(os.path.join(tmp_path, "data", "final.bin"))


def test_euler2d_subprogram_matches_reference(tmp_path):
    g = golden("euler2d_blast64_plm15_rk2")
    dt = float(g["dt"])
    run(["euler2d", "n=64", "tfinal=%r" % float(9.5 * dt), "riemann=hlle", "plm_theta=1.5", "rk_order=2", "steps_per_call=3"], str(tmp_path))
    d = read_dump(os.path.join(tmp_path, "data", "final.bin"))
    assert d["iteration"] == 10
    assert bits_equal(d["data"], g["u_10"])


@pytest.mark.parametrize("case,args", [
    ("cloud_nr32_plm_rk2", ["nr=32", "num_decades=1", "rk_order=2", "max_steps=3"]),
    ("cloud_nr24_pcm_rk1", ["nr=24", "num_decades=1", "rk_order=1", "reconstruct_method=1", "max_steps=4"]),
    ("cloud_nr20x2dec_plm_rk2", ["nr=20", "num_decades=2", "rk_order=2", "plm_theta=1.5", "max_steps=2"]),
])
def test_cloud_subprogram_matches_reference(tmp_path, case, args):
    """`mara_hip cloud`: grid, units, envelope-model initial condition, nozzle row and the device steps must
    reproduce the reference-composed run (oracle/ref_drivers/cloud_ref.cpp) bit for bit."""
    g = golden(case)
    stdout = run(["cloud", "tfinal=1e9", "write_inflow=1"] + args, str(tmp_path))
    d = read_dump(os.path.join(tmp_path, "data", "final.bin"))
    assert d["iteration"] == int(g["nsteps"])
    assert bits_equal(d["vertices"], np.concatenate([g["rv"], g["qv"]]))
    inflow = read_dump(os.path.join(tmp_path, "data", "inflow0.bin"))["data"]
    assert bits_equal(inflow, g["inflow"][0])
    assert bits_equal(d["data"], g["un"]), np.abs(d["data"] - g["un"]).max()
    assert "kzps=" in stdout


@pytest.mark.parametrize("gpus", [2, 4])
def test_cloud_subprogram_over_radial_slabs_matches_reference(tmp_path, gpus):
    """BASELINE config 4's decomposition through the sub-program itself: `mara_hip cloud gpus=N` steps N radial slabs (one process, slab r on
    device r; on a one-GPU box the slabs share the device) with the two-row halo per stage, and must land on the reference-composed run bit
    for bit, like gpus=1 - including a diagnostics task, which gathers the slabs."""
    g = golden("cloud_nr70_plm_rk2")
    stdout = run(["cloud", "tfinal=1e9", "write_inflow=1", "nr=70", "num_decades=1", "rk_order=2", "max_steps=2", "gpus=%d" % gpus, "dfi=1e-9"], str(tmp_path))
    d = read_dump(os.path.join(tmp_path, "data", "final.bin"))
    assert d["iteration"] == int(g["nsteps"])
    assert bits_equal(d["data"], g["un"]), np.abs(d["data"] - g["un"]).max()
    assert "share devices round-robin" in stdout and "write diagnostics" in stdout


def test_euler2d_subprogram_over_slabs_matches_reference(tmp_path):
    g = golden("euler2d_blast64_plm15_rk2")
    dt = float(g["dt"])
    run(["euler2d", "n=64", "tfinal=%r" % float(9.5 * dt), "riemann=hlle", "plm_theta=1.5", "rk_order=2", "steps_per_call=3", "gpus=3"], str(tmp_path))
    d = read_dump(os.path.join(tmp_path, "data", "final.bin"))
    assert d["iteration"] == 10
    assert bits_equal(d["data"], g["u_10"])


@pytest.mark.parametrize("case", ["binary_d2_b16", "binary_d1_b24_nu", "binary_d3_b8_axisym", "binary_d2_b16_q", "binary_d2_b16_live"])
def test_binary_subprogram_matches_reference(tmp_path, case):
    """The whole host path of `mara_hip binary`: set-up with the host libm (bit-exact vertices), dt choice, RK steps,
    totals and the run-loop message. Field tolerance as in tests/test_gpu_binary.py (device libm: 1e-12 of the field scale)."""
    import json
    g = golden(case)
    over = json.loads(str(g["config"]))
    nsteps = int(over.pop("nsteps"))
    args = ["%s=%r" % (k, int(v) if float(v).is_integer() and k in ("depth", "block_size", "fixed_dt", "rk_order", "axisymmetric_cs2", "counter_rotate", "no_accretion_force", "conserve_linear_p") else float(v)) for k, v in over.items()]
    args.append("focus_factor=1e9")          # the uniform-depth tree these vectors were made on (the default, 2.0, grades the tree)
    stdout = run(["binary"] + args + ["max_iterations=%d" % nsteps, "tfinal=100.0"], str(tmp_path))
    d = read_dump(os.path.join(tmp_path, "data", "final.bin"))
    n = g["u_final"].shape[0]
    assert d["iteration"] == nsteps
    assert bits_equal(d["vertices"][:n + 1], g["xv"])
    sc = g["scalars"]
    assert abs(d["time"] - sc[0]) <= 1e-13 * sc[0]
    scale = np.abs(g["u_final"]).reshape(-1, 3).max(axis=0)
    err = np.abs(d["data"] - g["u_final"]).reshape(-1, 3).max(axis=0)
    assert np.all(err <= 1e-12 * scale), err / scale
    acc = d["vertices"][n + 1:]
    assert np.allclose(acc, sc[2:12], rtol=1e-9, atol=1e-11 * np.abs(sc[2:12]).max())
    assert "[%04d] orbits=" % nsteps in stdout and "kzps=" in stdout


def test_binary_subprogram_rejects_what_is_not_built(tmp_path):
    out = subprocess.run([EXE, "binary", "reconstruct_method=weno"], cwd=str(tmp_path), capture_output=True, text=True)
    assert out.returncode == 1 and "must be plm or pcm" in out.stdout


H5DUMP = "/opt/conda/bin/h5dump"


def _h5_dataset(path, name, dtype=np.float64):
    """A dataset of an HDF5 file as a flat array, parsed from h5dump's text output with round-trip float formatting (there is
    no h5py in this image, and h5dump's binary export does not handle array-typed elements)."""
    import re
    out = subprocess.run([H5DUMP, "-d", name, "-m", "%.17g", "-w", "65535", path], check=True, capture_output=True, text=True).stdout
    body = out[out.index("DATA {") + 6:out.rindex("}")]
    body = re.sub(r"\([0-9,]+\):", " ", body)
    tokens = [t for t in re.split(r"[\s,\[\]{}]+", body) if t]
    return np.array([float(t) for t in tokens]).astype(dtype)


@pytest.mark.skipif(not os.path.exists(H5DUMP), reason="needs the HDF5 tools of the image")
def test_sedov_checkpoint_and_restart(tmp_path):
    """Checkpoints in the reference's layout (src/subprog_sedov.cpp:329-335, :486-495): the file written at iteration 10 holds
    the reference-generated state bit for bit, and a run restarted from it ends exactly where the uninterrupted run does."""
    g = golden("sedov_newtonian_nr256")
    dt = 0.4 * (g["vertices"][1] - g["vertices"][0])
    common = ["sedov", "newtonian=1", "nr=256", "outer_radius=100", "tfinal=%r" % float(99.5 * dt), "cpi=%r" % float(8.5 * dt)]      # due once a step STARTS at t >= 8.5 dt: after step 10
    stdout = run(common + ["outdir=a"], str(tmp_path))
    assert "write checkpoint: a/chkpt.0000.h5" in stdout and "write checkpoint: a/chkpt.0001.h5" in stdout
    chk = os.path.join(tmp_path, "a", "chkpt.0001.h5")
    assert bits_equal(_h5_dataset(chk, "/solution/conserved").reshape(-1, 5), g["u_10"])
    assert bits_equal(_h5_dataset(chk, "/solution/vertices"), g["vertices"])
    assert list(_h5_dataset(chk, "/solution/iteration", np.int32)) == [10, 1]
    assert _h5_dataset(chk, "/schedule/write_checkpoint/num_times_performed", np.int32)[0] == 1
    header = subprocess.run([H5DUMP, "-H", chk], check=True, capture_output=True, text=True).stdout
    assert "H5T_ARRAY { [5] H5T_IEEE_F64LE }" in header and "H5T_ARRAY { [2] H5T_STD_I32LE }" in header
    assert 'GROUP "config"' in header and 'DATASET "newtonian"' in header and 'GROUP "write_checkpoint"' in header
    a = read_dump(os.path.join(tmp_path, "a", "final.bin"))
    assert a["iteration"] == 100 and bits_equal(a["data"], g["u_100"])
    # restart: the stored run configuration comes back (newtonian, nr, tfinal, cpi); only outdir is overridden
    stdout = run(["sedov", "restart=a/chkpt.0001.h5", "outdir=b"], str(tmp_path))
    b = read_dump(os.path.join(tmp_path, "b", "final.bin"))
    assert b["iteration"] == 100 and b["time"] == a["time"]
    assert bits_equal(b["data"], a["data"])
    # upstream's sedov / cloud store the schedule BEFORE marking the task completed (subprog_sedov.cpp:486-495, :565-568), so a
    # run restarted from checkpoint 1 numbers its next checkpoint 1 again
    assert "write checkpoint: b/chkpt.0001.h5" in stdout and "b/chkpt.0000.h5" not in stdout


@pytest.mark.skipif(not os.path.exists(H5DUMP), reason="needs the HDF5 tools of the image")
def test_cloud_checkpoint_and_restart(tmp_path):
    g = golden("cloud_nr32_plm_rk2")
    args = ["cloud", "nr=32", "num_decades=1", "rk_order=2", "max_steps=3"]
    run(args + ["outdir=a", "cpi=%r" % float(0.5 * float(g["dt"])), "dfi=0"], str(tmp_path))        # due once a step starts at t >= 0.5 dt
    a = read_dump(os.path.join(tmp_path, "a", "final.bin"))
    assert bits_equal(a["data"], g["un"])
    chk = os.path.join(tmp_path, "a", "chkpt.0001.h5")          # written after the second step
    assert list(_h5_dataset(chk, "/solution/iteration", np.int32)) == [2, 1]
    assert bits_equal(_h5_dataset(chk, "/solution/radial_vertices"), g["rv"]) and bits_equal(_h5_dataset(chk, "/solution/polar_vertices"), g["qv"])
    header = subprocess.run([H5DUMP, "-H", "-d", "/solution/conserved", chk], check=True, capture_output=True, text=True).stdout
    assert "H5T_ARRAY { [5] H5T_IEEE_F64LE }" in header and "( 32, 32 )" in header
    run(["cloud", "restart=a/chkpt.0001.h5", "outdir=b", "max_steps=3"], str(tmp_path))
    b = read_dump(os.path.join(tmp_path, "b", "final.bin"))
    assert b["iteration"] == 3 and bits_equal(b["data"], g["un"])


@pytest.mark.parametrize("case", ["binary_tree_d3_b8", "binary_tree_d3_b12_nu", "binary_tree_d3_b8_q"])
def test_binary_subprogram_on_graded_trees_matches_reference(tmp_path, case):
    """`mara_hip binary` with a refinement predicate that grades the tree: tree construction, block vertices and solver data on
    the host, block kernels on the device; against vectors made with the reference's own tree machinery."""
    import json
    g = golden(case)
    over = json.loads(str(g["config"]))
    nsteps = int(over.pop("nsteps"))
    ints = ("depth", "block_size", "fixed_dt", "rk_order", "conserve_linear_p")
    args = ["%s=%r" % (k, int(v) if k in ints else float(v)) for k, v in over.items()]
    stdout = run(["binary"] + args + ["max_iterations=%d" % nsteps, "tfinal=100.0"], str(tmp_path))
    d = read_dump(os.path.join(tmp_path, "data", "final.bin"))
    nb = len(g["blocks"])
    assert "block tree: %d blocks" % nb in stdout
    assert np.array_equal(d["vertices"][:3 * nb].reshape(nb, 3), g["blocks"])
    assert d["iteration"] == nsteps and abs(d["time"] - g["scalars"][0]) <= 1e-13 * g["scalars"][0]
    scale = np.abs(g["u_final"]).reshape(-1, 3).max(axis=0)
    err = np.abs(d["data"] - g["u_final"]).reshape(-1, 3).max(axis=0)
    assert np.all(err <= 1e-12 * scale), err / scale


def test_binary_subprogram_default_configuration_runs(tmp_path):
    # depth=4 block_size=24 focus_factor=2: the graded default; with the three tasks switched off the iterations are batched
    stdout = run(["binary", "max_iterations=20", "steps_per_call=10", "cpi=0", "dfi=0", "tsi=0"], str(tmp_path))
    assert "block tree: 64 blocks of 24 x 24 zones" in stdout and "[0020] orbits=" in stdout
    d = read_dump(os.path.join(tmp_path, "data", "final.bin"))
    assert np.isfinite(d["data"]).all() and (d["data"][..., 0] > 0).all()


def _binary_args(over):
    ints = ("depth", "block_size", "fixed_dt", "rk_order", "axisymmetric_cs2", "counter_rotate", "no_accretion_force", "conserve_linear_p")
    return ["%s=%r" % (k, int(v) if k in ints else float(v)) for k, v in over.items()]


@pytest.mark.skipif(not os.path.exists(H5DUMP), reason="needs the HDF5 tools of the image")
@pytest.mark.parametrize("case", ["binary_tree_d2_b16_uniform", "binary_tree_d3_b8", "binary_tree_d3_b8_q"])
def test_binary_tasks_checkpoint_diagnostics_time_series_and_restart(tmp_path, case):
    """`mara_hip binary` with its three tasks (src/subprog_binary.cpp:296-385): file layout as subprog_binary_io.cpp writes it (one
    dataset per tree block named level:ii-jj, compound orbital elements and time-series samples), contents against the
    reference-generated vectors, and restart == uninterrupted run. With tiny intervals a task falls due from the second
    iteration on (the schedule looks at the time the step started from), so files number 0000 (initial), 0001 (after step 2)..."""
    import json
    g = golden(case)
    over = json.loads(str(g["config"]))
    nsteps = int(over.pop("nsteps"))
    assert nsteps >= 2
    qform = int(over.get("conserve_linear_p", 1)) == 0
    args = ["binary"] + _binary_args(over) + ["max_iterations=%d" % nsteps, "tfinal=100.0", "cpi=1e-9", "dfi=1e-9", "tsi=1e-9"]
    stdout = run(args + ["outdir=a"], str(tmp_path))
    last = nsteps - 1
    for prefix in ("chkpt", "diagnostics"):
        for k in range(last + 1):
            assert "write %s: a/%s.%04d.h5" % ("checkpoint" if prefix == "chkpt" else "diagnostics", prefix, k) in stdout
    chk = os.path.join(tmp_path, "a", "chkpt.%04d.h5" % last)
    header = subprocess.run([H5DUMP, "-H", chk], check=True, capture_output=True, text=True).stdout
    for name in ('GROUP "solution"', 'GROUP "conserved_u"', 'GROUP "conserved_q"', 'GROUP "schedule"', 'GROUP "run_config"', 'DATASET "time_series"',
                 'DATASET "orbital_elements_grav"', '"cm_velocity_y"', '"eccentricity"', '"disk_angular_momentum"', '"position_of_mass2"',
                 'GROUP "record_time_series"', "H5T_ARRAY { [3] H5T_IEEE_F64LE }", "H5T_ARRAY { [2] H5T_IEEE_F64LE }"):
        assert name in header, name
    assert list(_h5_dataset(chk, "/solution/iteration", np.int32)) == [nsteps, 1]
    assert abs(_h5_dataset(chk, "/solution/time")[0] - g["scalars"][0]) <= 1e-13 * g["scalars"][0]
    bs = int(over["block_size"])
    scale = np.abs(g["u_final"]).reshape(-1, 3).max(axis=0)
    form = "conserved_q" if qform else "conserved_u"
    for k, (level, i, j) in enumerate(g["blocks"]):
        width = 1 + int(np.log10(1 << level))
        name = "%d:%0*d-%0*d" % (level, width, i, width, j)
        cells = _h5_dataset(chk, "/solution/%s/%s" % (form, name)).reshape(bs, bs, 3)[..., ::-1]       # std::tuple storage order of libstdc++
        assert np.all(np.abs(cells - g["u_final"][k]).reshape(-1, 3).max(axis=0) <= 1e-12 * scale), name
    # the time series: samples at t = 0 and after every step from the second on; the last one belongs to the final state
    series = _h5_dataset(chk, "/time_series").reshape(-1, 47)
    assert len(series) == nsteps and series[0, 0] == 0.0
    assert abs(series[-1, 0] - g["scalars"][0]) <= 1e-13 * g["scalars"][0]
    assert abs(series[-1, 1] - g["diag_scalars"][0]) <= 1e-12 * g["diag_scalars"][0]          # disk_mass
    assert abs(series[-1, 2] - g["diag_scalars"][1]) <= 1e-11 * abs(g["diag_scalars"][1])     # disk_angular_momentum
    assert np.allclose(series[-1, 43:47], g["diag_scalars"][2:6], rtol=0, atol=1e-13)         # positions of the two masses
    assert np.allclose(series[-1, 33:43], g["scalars"][32:42], rtol=1e-13, atol=1e-15)        # orbital_elements
    sched = _h5_dataset(chk, "/schedule/write_checkpoint/num_times_performed", np.int32)[0]
    assert sched == nsteps                                                                     # completed before the state is written (:333-335)
    # diagnostics file of the final state
    dg = os.path.join(tmp_path, "a", "diagnostics.%04d.h5" % last)
    vscale = np.abs(g["diag_fields"][:, 1:]).max()
    for k, (level, i, j) in enumerate(g["blocks"][:6]):
        width = 1 + int(np.log10(1 << level))
        name = "%d:%0*d-%0*d" % (level, width, i, width, j)
        sig = _h5_dataset(dg, "/sigma/" + name).reshape(bs, bs)
        assert np.all(np.abs(sig - g["diag_fields"][k, 0]) <= 1e-12 * scale[0])
        vp = _h5_dataset(dg, "/phi_velocity/" + name).reshape(bs, bs)
        assert np.all(np.abs(vp - g["diag_fields"][k, 2]) <= 1e-10 * vscale)
        verts = _h5_dataset(dg, "/vertices/" + name).reshape(bs + 1, bs + 1, 2)
        assert bits_equal(verts[:, 0, 0], g["xv"][k, 0]) and bits_equal(verts[0, :, 1], g["xv"][k, 1])
    assert np.allclose(_h5_dataset(dg, "/position_of_mass1"), g["diag_scalars"][2:4], rtol=0, atol=1e-13)
    # restart from the checkpoint written after step 2 and compare with the uninterrupted run
    a = read_dump(os.path.join(tmp_path, "a", "final.bin"))
    if nsteps > 2:
        stdout = run(["binary", "restart=a/chkpt.0001.h5", "outdir=b"], str(tmp_path))
        b = read_dump(os.path.join(tmp_path, "b", "final.bin"))
        assert b["iteration"] == nsteps and b["time"] == a["time"]
        assert bits_equal(b["data"], a["data"]) and bits_equal(b["vertices"], a["vertices"])
        assert "write checkpoint: b/chkpt.0002.h5" in stdout and "b/chkpt.0000.h5" not in stdout
        series_b = _h5_dataset(os.path.join(tmp_path, "b", "chkpt.0002.h5"), "/time_series").reshape(-1, 47)
        assert bits_equal(series_b, series)
    else:
        stdout = run(["binary", "restart=a/chkpt.0001.h5", "outdir=b", "max_iterations=%d" % (nsteps + 1)], str(tmp_path))
        b = read_dump(os.path.join(tmp_path, "b", "final.bin"))
        assert b["iteration"] == nsteps + 1
        stdout = run(args[:-3] + ["outdir=c", "cpi=0", "dfi=0", "tsi=0", "max_iterations=%d" % (nsteps + 1)], str(tmp_path))
        c = read_dump(os.path.join(tmp_path, "c", "final.bin"))
        assert bits_equal(b["data"], c["data"]) and b["time"] == c["time"]


def test_closing_step_runs_the_tasks_once_more(tmp_path):
    """Upstream's run loops end with one more `run_tasks(next(state))` (subprog_binary.cpp:437, subprog_sedov.cpp:644, subprog_cloud.cpp:935):
    the state it produces is dropped, but a task that falls due on it still writes its file - numbered after the loop's last one and
    holding iteration N + 1. final.bin is the state the loop ended with."""
    stdout = run(["binary", "depth=2", "block_size=8", "focus_factor=1e9", "domain_radius=4.0", "tfinal=0.02", "cpi=1e-9", "dfi=0", "tsi=0"], str(tmp_path))
    d = read_dump(os.path.join(tmp_path, "data", "final.bin"))
    lines = [l for l in stdout.splitlines() if l.startswith("[")]
    n = len(lines)
    assert d["iteration"] == n and d["time"] / (2 * np.pi) >= 0.02
    last = os.path.join(tmp_path, "data", "chkpt.%04d.h5" % n)                 # 0000 initial, then one per step from the second on, then the closing one
    assert os.path.exists(last) and not os.path.exists(os.path.join(tmp_path, "data", "chkpt.%04d.h5" % (n + 1)))
    if os.path.exists(H5DUMP):
        assert list(_h5_dataset(last, "/solution/iteration", np.int32)) == [n + 1, 1]


@pytest.mark.skipif(not os.path.exists(H5DUMP), reason="needs the HDF5 tools of the image")
@pytest.mark.parametrize("name,extra", [("sedovdiag_newtonian_nr256", ["newtonian=1"]), ("sedovdiag_srhd_nr256", [])])
def test_sedov_diagnostics_and_time_series_files(tmp_path, name, extra):
    """`mara_hip sedov` write_diagnostics and write_time_series (src/subprog_sedov.cpp:497-529): diagnostics.NNNN.h5 with the reference's
    dataset names and time_series.h5 with one row per sample in extendible datasets; the sample taken after step 100 against the
    reference-composed vectors (entropy to a few ulp, everything else bit for bit)."""
    g = golden(name)
    dt = 0.4 * (g["vertices"][1] - g["vertices"][0])
    # tasks fall due when a step STARTS at t >= interval: interval 98.5 dt -> sampled after step 100
    stdout = run(["sedov", "nr=256", "outer_radius=100", "tfinal=%r" % float(99.5 * dt), "cpi=0", "dfi=%r" % float(98.5 * dt), "tsi=%r" % float(98.5 * dt)] + extra, str(tmp_path))
    assert "write diagnostics: data/diagnostics.0000.h5" in stdout and "write diagnostics: data/diagnostics.0001.h5" in stdout
    d = read_dump(os.path.join(tmp_path, "data", "final.bin"))
    assert d["iteration"] == 100 and bits_equal(d["data"], g["u_100"])
    f = os.path.join(tmp_path, "data", "diagnostics.0001.h5")
    ref = g["fields_100"]
    for k, key in ((1, "gas_pressure"), (2, "mass_density"), (3, "radial_gamma_beta")):
        assert bits_equal(_h5_dataset(f, "/" + key), ref[k]), key
    ent = _h5_dataset(f, "/specific_entropy")
    assert np.all(np.abs(ent - ref[0]) <= 32 * np.spacing(np.maximum(np.abs(ref[0]), 1.0)))
    assert bits_equal(_h5_dataset(f, "/radial_coordinates"), (g["vertices"][:-1] + g["vertices"][1:]) * 0.5)
    cols = ["time", "shock_radius", "shock_radius_upstream", "shock_radius_downstream", "shock_radius_interpolated", "shock_velocity"]
    series = g["series_100"]
    ts = os.path.join(tmp_path, "data", "time_series.h5")
    for k, key in enumerate(cols):
        assert abs(_h5_dataset(f, "/" + key)[0] - series[k]) <= 1e-13 * abs(series[k]), key      # time: 100 additions of dt against n * dt
        column = _h5_dataset(ts, "/" + key)
        assert len(column) == 2 and abs(column[1] - series[k]) <= 1e-13 * abs(series[k]), key
    header = subprocess.run([H5DUMP, "-H", "-p", ts], check=True, capture_output=True, text=True).stdout
    assert "CHUNKED" in header and "( 1000 )" in header and "H5S_UNLIMITED" in header and 'GROUP "run_config"' in header


@pytest.mark.timeout(600)
def test_cloud_full_size_equatorial_symmetry(tmp_path):
    """BASELINE config 4 at full size (`mara_hip cloud nr=4096 num_decades=1 rk_order=2 plm_theta=1.2`, strict arithmetic): the nozzle model
    is symmetric about the equator (model_jet_nozzle: gamma_beta(q) + gamma_beta(pi - q)), so after three steps the state mirrors about
    theta = pi / 2 - the polar momentum changes sign - to the rounding of the polar grid (linspace(0, pi) is not exactly mirror-symmetric),
    and every cell keeps a positive density and energy."""
    run(["cloud", "nr=4096", "num_decades=1", "rk_order=2", "plm_theta=1.2", "max_steps=3", "tfinal=100.0", "cpi=0", "dfi=0", "tsi=0"], str(tmp_path))
    d = read_dump(os.path.join(tmp_path, "data", "final.bin"))
    u = d["data"]
    assert d["iteration"] == 3 and u.shape == (4096, 4096, 5)
    assert (u[..., 0] > 0).all() and (u[..., 4] > 0).all()
    sign = np.array([1.0, 1.0, -1.0, 1.0, 1.0])
    mirror = u[:, ::-1, :] * sign
    scale = np.abs(u).reshape(-1, 5).max(axis=0)
    scale[2] = scale[1]                                        # the polar momentum is measured against the radial one
    err = np.abs(u - mirror).reshape(-1, 5).max(axis=0)
    assert np.all(err <= 1e-9 * scale), err / scale


@pytest.mark.skipif(not os.path.exists(H5DUMP), reason="needs the HDF5 tools of the image")
def test_output_files_hold_what_the_reference_tools_read(tmp_path):
    """Format pin from the CONSUMER side: no reference-written file can be produced here, but the reference ships the
    readers of its files - tools/plot_cloud.py and tools/plot_binary.py. Every dataset, group and compound member those scripts access must
    exist, with that nesting, in the files `mara_hip` writes:
      tools/plot_cloud.py:18-25, :77-102, :134-136     diagnostics.NNNN.h5 of `cloud`: top-level datasets
      tools/plot_binary.py:99-106, :143-149             diagnostics.NNNN.h5 of `binary`: groups vertices | sigma | radial_velocity | phi_velocity / <block>
      tools/plot_binary.py:222-256, :331-350, :386-438  chkpt.NNNN.h5 of `binary`: compound /time_series (members by name) and /run_config items."""
    out = run(["cloud", "nr=32", "num_decades=1", "rk_order=2", "max_steps=3", "cpi=0", "dfi=1e-9", "outdir=c"], str(tmp_path))
    assert "diagnostics" in out
    hdr = subprocess.run([H5DUMP, "-H", os.path.join(tmp_path, "c", "diagnostics.0001.h5")], check=True, capture_output=True, text=True).stdout
    top_level = hdr.split("\n")
    for name in ("time", "radial_vertices", "polar_vertices", "mass_density", "gas_pressure", "radial_gamma_beta", "radial_energy_flow",
                 "solid_angle_at_theta", "shock_midpoint_radius", "shock_pressure_radius", "shock_luminosity_radius"):
        assert any(line.startswith('   DATASET "%s"' % name) for line in top_level), name          # depth 1: h5f['<name>']
    out = run(["binary", "depth=2", "block_size=16", "focus_factor=1e9", "max_iterations=3", "tfinal=100.0", "cpi=1e-9", "dfi=1e-9", "tsi=1e-9", "outdir=b"], str(tmp_path))
    hdr = subprocess.run([H5DUMP, "-H", os.path.join(tmp_path, "b", "diagnostics.0001.h5")], check=True, capture_output=True, text=True).stdout
    for group in ("vertices", "sigma", "radial_velocity", "phi_velocity"):
        assert '   GROUP "%s"' % group in hdr, group
        body = hdr[hdr.index('   GROUP "%s"' % group):]
        assert '      DATASET "' in body[:body.index("\n   }")], group                               # one dataset per block inside the group
    hdr = subprocess.run([H5DUMP, "-H", os.path.join(tmp_path, "b", "chkpt.0001.h5")], check=True, capture_output=True, text=True).stdout
    ts = hdr[hdr.index('DATASET "time_series"'):]
    ts = ts[:ts.index("DATASPACE")]
    for member in ("time", "disk_mass", "mass_ejected", "mass_accreted_on", "disk_angular_momentum", "angular_momentum_ejected", "integrated_torque_on",
                   "angular_momentum_accreted_on", "work_done_on", "orbital_elements_acc", "orbital_elements_grav", "orbital_elements",
                   "position_of_mass1", "position_of_mass2", "elements", "separation", "eccentricity", "total_mass", "pomega", "tau", "cm_position_x"):
        assert '"%s"' % member in ts, member
    rc = hdr[hdr.index('GROUP "run_config"'):]
    for item in ("mass_ratio", "eccentricity", "disk_mass", "begin_live_binary"):
        assert 'DATASET "%s"' % item in rc, item


H5_REF = os.path.join(ROOT, "oracle", "_ref", "h5_ref")
H5_REF_ENV = dict(os.environ, LD_LIBRARY_PATH="/usr/lib/x86_64-linux-gnu")


def _h5dump_all(path):
    out = subprocess.run([H5DUMP, "-m", "%.17g", path], check=True, capture_output=True, text=True).stdout
    return out.split("\n", 1)[1]


@pytest.mark.skipif(not (os.path.exists(H5DUMP) and os.path.exists(H5_REF)), reason="needs h5dump and the prebuilt oracle/_ref/h5_ref")
@pytest.mark.parametrize("prog", ["sedov", "cloud"])
def test_checkpoints_pass_through_the_reference_readers_and_writers_and_restart_from_its_file(tmp_path, prog):
    """The file a real run of `mara_hip` writes is READ by the reference's own read_solution / read_schedule / read_config
    (src/subprog_sedov.cpp:338-346, src/subprog_cloud.cpp:599-609, src/app_serialize.hpp:72-113) and WRITTEN again by its
    write_solution / write_schedule / write_config (oracle/_ref/h5_ref, the reference's headers): the reference's file is the same file
    (h5dump: every type, dataspace and value), and `mara_hip ... restart=<the reference-written file>` ends bit for bit where the
    uninterrupted run ends."""
    if prog == "sedov":
        g = golden("sedov_newtonian_nr256")
        dt = 0.4 * (g["vertices"][1] - g["vertices"][0])
        args = ["sedov", "newtonian=1", "nr=256", "outer_radius=100", "tfinal=%r" % float(99.5 * dt), "cpi=%r" % float(8.5 * dt)]
        more = []
    else:
        g = golden("cloud_nr32_plm_rk2")
        args = ["cloud", "nr=32", "num_decades=1", "rk_order=2", "max_steps=3", "cpi=%r" % float(0.5 * float(g["dt"])), "dfi=0"]
        more = ["max_steps=3"]
    run(args + ["outdir=a"], str(tmp_path))
    a = read_dump(os.path.join(tmp_path, "a", "final.bin"))
    ours = os.path.join(tmp_path, "a", "chkpt.0001.h5")
    spec, theirs = os.path.join(tmp_path, "spec.txt"), os.path.join(tmp_path, "reference_written.h5")
    subprocess.run([H5_REF, "read", prog, ours, spec], check=True, env=H5_REF_ENV)
    subprocess.run([H5_REF, "write", spec, theirs], check=True, env=H5_REF_ENV)
    assert _h5dump_all(theirs) == _h5dump_all(ours)
    run([prog, "restart=reference_written.h5", "outdir=b"] + more, str(tmp_path))
    b = read_dump(os.path.join(tmp_path, "b", "final.bin"))
    assert b["iteration"] == a["iteration"] and b["time"] == a["time"] and bits_equal(b["data"], a["data"])

"""CPU tests of the compiled host's logic that needs no GPU: sub-program registry, typed key=value configuration
(unknown key / wrong type -> error, as the reference's mara::config_t, src/app_config.hpp:103-136), and the loud
failure when no HIP device is present (no CPU fallback)."""
import os
import subprocess
import pytest
from conftest import ROOT

EXE = os.path.join(ROOT, "mara3_amd", "host", "mara_hip")


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__
    __graft_entry__.build()
    assert os.path.exists(EXE)


def run(args):
    return subprocess.run([EXE] + args, capture_output=True, text=True, timeout=60)


def test_registry_lists_subprograms():
    out = run([])
    assert out.returncode == 0
    for name in ("sedov", "cloud", "euler2d"):
        assert "mara_hip " + name in out.stdout
    out = run(["nosuchprogram"])
    assert "invalid sub-program 'nosuchprogram'" in out.stdout


@pytest.mark.parametrize("prog", ["sedov", "cloud", "euler2d"])
def test_config_errors(prog):
    out = run([prog, "nosuchkey=1"])
    assert out.returncode == 1 and "config got unknown key: nosuchkey" in out.stdout
    out = run([prog, "outdir"])
    assert out.returncode == 1 and "key=val" in out.stdout
    key = "n" if prog == "euler2d" else "nr"
    out = run([prog, key + "=12x"])
    assert out.returncode == 1 and "wrong data type" in out.stdout


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    out = run(["sedov", "newtonian=1", "tfinal=0.001"])
    assert out.returncode == 1
    assert "mh_create" in out.stdout and "config" in out.stdout      # the configuration was parsed and printed first


def test_hdf5_checkpoint_layer_round_trip_and_layout(tmp_path):
    """The checkpoint layer of the compiled hosts (mara3_amd/host/h5_checkpoint.hpp) without a GPU: write, read back, and
    check the on-disk layout the reference's writers produce (types as h5dump names them)."""
    import subprocess
    exe = os.path.join(ROOT, "mara3_amd", "host", "h5_selftest")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "mara3_amd", "host"), "h5_selftest"])
    path = os.path.join(tmp_path, "t.h5")
    out = subprocess.run([exe, path], capture_output=True, text=True)
    if out.returncode == 77:
        pytest.skip("libhdf5 is not available in this environment")
    assert out.returncode == 0 and "round trip ok" in out.stdout
    h5dump = "/opt/conda/bin/h5dump"
    if os.path.exists(h5dump):
        header = subprocess.run([h5dump, "-H", path], check=True, capture_output=True, text=True).stdout
        for needle in ('DATASET "conserved"', "H5T_ARRAY { [5] H5T_IEEE_F64LE }", "H5T_ARRAY { [2] H5T_STD_I32LE }", 'GROUP "schedule"',
                       'DATASET "num_times_performed"', 'DATASET "last_performed"', 'GROUP "config"', "STRSIZE 4;"):
            assert needle in header, needle

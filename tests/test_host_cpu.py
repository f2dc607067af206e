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

"""The row-range guard (mara3_amd/csrc/row_check.hpp, libmara_hip_check.so = the row-marching kernels rebuilt with -DMH_CHECK_ROWS).

Round 3's GPU memory fault (a fused 2-D step across slab cuts whose producer asked for rows two beyond the four its caller allocates;
whether that read faulted depended on where the allocation ended) is closed here as a CLASS: every row (plane) index a kernel of these
families forms an address from is recorded, and over the chunk / tail / segment combinations of the suite the recorded range must stay
inside the rows that exist: [-2, n0 + 1], and [-4, n0 + 3] on the cut sides of the fused 2-D step. A check build holds every access to
those rows itself, so the test runs safely even on a kernel that would leave them - which is what the second library is:
libmara_hip_check_noclamp.so has the missing clamp of round 3 back (-DMH_PROBE_FUSED_NO_EXTERNAL_CLAMP), and the guard must flag it.
Each family runs in a child process (tests/row_range_child.py), one after the other: the library is chosen at load time."""
import json
import os
import subprocess
import sys
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "mara3_amd", "libmara_hip_check.so")
NOCLAMP = os.path.join(ROOT, "mara3_amd", "libmara_hip_check_noclamp.so")


def run_child(family, library):
    assert os.path.exists(library), "build the check libraries: make -C mara3_amd/csrc check (__graft_entry__.build() does)"
    env = dict(os.environ)
    env["MARA_HIP_LIBRARY"] = library
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "row_range_child.py"), family], env=env, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    rows = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert rows, p.stdout[-500:]
    return rows


@pytest.mark.parametrize("family", ["euler2d", "euler2d_fused", "euler2d_fused_cuts", "cloud", "cloud_fused", "cloud_fused_cuts", "euler3d", "binary"])
def test_no_row_request_leaves_the_stored_rows(family):
    for r in run_child(family, CHECK):
        ghost = 4 if r["cut"] else 2
        assert r["status"] == 0, r
        # the guard is recording: the kernels did ask for the field's first and last rows ...
        assert r["lo"] <= 0 and r["hi"] >= r["n0"] - 1, r
        # ... and for nothing beyond the rows that exist
        assert -ghost <= r["lo"] and r["hi"] <= r["n0"] - 1 + ghost, r


def test_the_guard_flags_the_round_3_fault():
    """the fused step across cuts WITHOUT the clamp of euler2d_fused.hip's row_of: the producer's look-ahead asks for rows beyond the four
    a cut side holds. (The check build holds the access itself, so nothing is read outside the allocation here.)"""
    rows = run_child("euler2d_fused_cuts", NOCLAMP)
    assert any(r["lo"] < -4 or r["hi"] > r["n0"] + 3 for r in rows), rows
    # and the same library on physical sides stays inside (its clamp there is untouched)
    for r in run_child("euler2d_fused", NOCLAMP):
        assert -2 <= r["lo"] and r["hi"] <= r["n0"] + 1, r


def test_product_library_carries_no_guard():
    import ctypes as C
    import mara3_amd
    lib = mara3_amd.load_library()
    out = (C.c_int32 * 2)()
    assert lib.mh_debug_row_range(0, out, 0) == -4          # MH_E_STATE
    assert b"MH_CHECK_ROWS" in lib.mh_last_error(None)

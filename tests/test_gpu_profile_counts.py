"""The profile switches of the steppers (what bench.py's `roofline.avg_launch_ms` is read from): since round 3 a profiled call brackets
its stage launches with ONE pair of events - events around every launch put two more markers between consecutive kernels and read 3 - 7 %
long on sub-millisecond stages - and reports the number of launches between them. Counts are exact; the average is a duration per launch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_context_stepper_counts_stage_launches_per_call():
    from mara3_amd import setups
    from mara3_amd.engine import EulerCartSolver
    shape, gamma = (128, 200), 1.4
    dl = (1.0 / shape[0], 1.0 / shape[1])
    u0 = setups.wave_ic(shape, gamma, seed=3)
    for arith, fuse, per_step in (("fast", None, 1), ("fast", False, 2), ("strict", None, 2)):
        s = EulerCartSolver(shape, dl, gamma, 1.5, "hllc", 2, "periodic", arith=arith, fuse=fuse)
        s.upload(u0)
        s.step(1e-3, 2)
        s.profile(True)
        s.step(1e-3, 3); s.step(1e-3, 1)
        ms, n = s.profile_read()
        assert n == 4 * per_step and 0.0 < ms < 50.0, (arith, fuse, ms, n)
        s.profile(False)
        s.step(1e-3, 2)
        assert s.profile_read()[1] == 0
        s.close()


def test_binary_solver_counts_stages_per_call():
    from mara3_amd import binary
    uniform = binary.config(depth=2, block_size=16, fixed_dt=1)          # (eager first stages inside a call)
    graded = binary.config(depth=3, block_size=8)
    for s in (binary.BinarySolver(uniform, arith="fast"), binary.BinarySolver(binary.config(depth=2, block_size=16)), binary.BinaryTreeSolver(graded)):
        s.next(2)
        s.profile(True)
        safe = s.next(3) + s.next(2)
        ms, n = s.profile(False)
        # (a safe-mode retry issues its stages again, and may leave an eager stage unused: exact only without one)
        assert (n == 10 if safe == 0 else n >= 10 + 2 * safe) and 0.0 < ms < 50.0, (ms, n, safe)
        assert s.profile(False) == (0.0, 0)
        s.close()


def test_native_slab_stepper_counts_launches():
    from mara3_amd import setups
    from mara3_amd.slab import NativeSlabStepper
    shape, gamma = (128, 200), 1.4
    dl = (1.0 / shape[0], 1.0 / shape[1])
    u0 = setups.wave_ic(shape, gamma, seed=4)
    for arith, want in (("fast", (0, 5)), ("strict", (5, 5))):
        st = NativeSlabStepper(shape, dl, gamma, 1.5, "hllc", 2, "periodic", arith=arith)
        st.load_slab(u0)
        st.step(1e-3, 2)
        st.profile(True)
        st.step(1e-3, 3); st.step(1e-3, 2); st.synchronize()
        (ms1, ms2), (n1, n2), rows = st.profile_read()
        assert (n1, n2) == want and ms2 > 0.0 and rows == shape[0], (arith, n1, n2, rows)
        st.profile(False)
        st.close()

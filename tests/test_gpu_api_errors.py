"""Error behaviour of the C ABI on a live device (include/mara_hip.h: every entry point returns 0 or a negative mh_error, nothing throws
across the boundary, mh_last_error carries the message): misuse is refused with a message and leaves the context usable."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import mara3_amd
    from mara3_amd import engine, _lib
    assert mara3_amd.load_library().mh_device_count() >= 1
    return mara3_amd, engine, _lib


def test_step_and_download_before_upload_are_refused(mods):
    mara3_amd, engine, L = mods
    s = engine.EulerCartSolver((32, 32), (1 / 32, 1 / 32), 1.4)
    with pytest.raises(mara3_amd.MaraHipError, match="upload"):
        s.step(1e-3, 1)
    u = np.zeros((32, 32, 5))
    u[..., 0] = 1.0
    u[..., 4] = 1.0
    s.upload(u)                                     # the context is still usable
    s.step(1e-3, 1)
    assert s.status() == 0
    s.close()


def test_wrong_sizes_and_descriptors_are_refused(mods):
    mara3_amd, engine, L = mods
    lib = L.load_library()
    s = engine.EulerCartSolver((32, 32), (1 / 32, 1 / 32), 1.4)
    u = np.zeros((32, 32, 5))
    assert lib.mh_upload(s.ctx, u.ctypes.data_as(C.c_void_p), 31 * 32) != 0          # cell count of another grid
    assert lib.mh_last_error(s.ctx)
    assert lib.mh_upload(s.ctx, None, 32 * 32) != 0                                   # null host pointer
    s.close()
    for kwargs in (dict(gamma=1.0), dict(gamma=float("nan")), dict(rk_order=3)):
        args = dict(shape=(32, 32), dl=(1 / 32, 1 / 32), gamma=1.4)
        args.update(kwargs)
        with pytest.raises(mara3_amd.MaraHipError):
            engine.EulerCartSolver(**args)
    with pytest.raises(mara3_amd.MaraHipError):
        engine.EulerCartSolver((32, 0), (1 / 32, 1 / 32), 1.4)
    ctx = C.c_void_p()
    assert lib.mh_create(C.byref(ctx), 9999) != 0                                      # no such device
    assert lib.mh_status_word(None, None) != 0


def test_cloud_and_sedov_contexts_refuse_foreign_calls(mods):
    mara3_amd, engine, L = mods
    lib = L.load_library()
    s = engine.EulerCartSolver((32, 32), (1 / 32, 1 / 32), 1.4)
    units = (C.c_double * 3)(1.0, 1.0, 1.0)
    assert lib.mh_cloud_diagnostics(s.ctx, units, None, None) != 0                     # not a cloud context
    assert lib.mh_sedov_diagnostics(s.ctx, None, None) != 0
    assert lib.mh_cloud_set_inflow(s.ctx, None) != 0
    s.close()

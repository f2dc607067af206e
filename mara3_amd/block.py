"""Host mirror of the native BLOCK stepper (libmara_hip.so: mh_block_*): a 3-D uniform-cartesian Euler run under the reference's
3-axis block decomposition (mara::propose_block_decomposition<3>, create_access_pattern_array, src/app_parallel.hpp:119-179) with a
ghost exchange on every cut side. This module owns no arithmetic."""
import ctypes as C
import numpy as np
from . import _lib as L
from .slab import euler_cart_desc


def block_layout(global_shape, world, rank):
    """(blocks per axis, coordinates, start, count) of block `rank` - host-side integer work of the C ABI (mh_block_layout)"""
    lib = L.load_library()
    n = (C.c_int * 3)(*[int(x) for x in global_shape])
    B, c, s, k = (C.c_int * 3)(), (C.c_int * 3)(), (C.c_int * 3)(), (C.c_int * 3)()
    L.check(lib.mh_block_layout(n, world, rank, B, c, s, k))
    return tuple(B), tuple(c), tuple(s), tuple(k)


class NativeBlock:
    """One rank's block (RCCL backend; `comm_id` as for NativeSlabStepper), or a wrapped member of a NativeBlockGroup."""

    def __init__(self, global_shape, dl, gamma, plm_theta=1.5, riemann="hlle", rk_order=2, bc="outflow", rank=0, world=1, comm_id=None,
                 device=0, chunk_rows=0, arith="strict", handle=None, self_exchange=False):
        self.lib = L.load_library()
        self.global_shape = tuple(int(n) for n in global_shape)
        if handle is not None:
            self.handle = handle
        else:
            d = euler_cart_desc(self.global_shape, dl, gamma, plm_theta, riemann, bc, chunk_rows, arith)
            self.handle = C.c_void_p()
            idbuf = C.create_string_buffer(bytes(comm_id), 128) if comm_id is not None else None
            L.check(self.lib.mh_block_create(C.byref(self.handle), C.byref(d), rk_order, rank, world, idbuf, 1 if self_exchange else 0, device))
        B, c, s, k = (C.c_int * 3)(), (C.c_int * 3)(), (C.c_int * 3)(), (C.c_int * 3)()
        L.check(self.lib.mh_block_extent(self.handle, B, c, s, k))
        self.blocks, self.coords, self.start, self.count = tuple(B), tuple(c), tuple(s), tuple(k)
        nb, msg = (C.c_int * 6)(), (C.c_size_t * 3)()
        L.check(self.lib.mh_block_neighbours(self.handle, nb, msg))
        self.neighbours, self.message_doubles = tuple(nb), tuple(msg)

    def connect(self, comm_id):
        L.check(self.lib.mh_block_connect(self.handle, C.create_string_buffer(bytes(comm_id), 128)))

    def use_comm(self, comm):
        """lend the block the process's communicator (mara3_amd.slab.NativeComm); not collective"""
        L.check(self.lib.mh_block_use_comm(self.handle, comm.handle))
        self._comm = comm

    def slices(self):
        return tuple(slice(s, s + k) for s, k in zip(self.start, self.count))

    def upload(self, u_block):
        u = np.ascontiguousarray(u_block, dtype=np.float64)
        assert u.shape == self.count + (5,), (u.shape, self.count)
        L.check(self.lib.mh_block_upload(self.handle, u.ctypes.data_as(C.c_void_p)))

    def download(self):
        u = np.empty(self.count + (5,))
        L.check(self.lib.mh_block_download(self.handle, u.ctypes.data_as(C.c_void_p)))
        return u

    def step(self, dt, nsteps=1):
        L.check(self.lib.mh_block_step(self.handle, dt, nsteps))

    def synchronize(self):
        L.check(self.lib.mh_block_synchronize(self.handle))

    def status(self):
        """(bits, flat index of the first failing cell in the GLOBAL host array or None); clears"""
        r = L.StepResult()
        L.check(self.lib.mh_block_status(self.handle, C.byref(r), (C.c_int * 3)(*self.global_shape)))
        return r.status, (None if r.status == 0 else int(r.first_bad_index))

    def profile(self, enable):
        """-> ((avg ms of the interior launch of stage 1, stage 2), (launches), interior cells) since the last call; then sets the switch"""
        ms, n, cells = (C.c_double * 2)(), (C.c_int * 2)(), C.c_long()
        L.check(self.lib.mh_block_profile(self.handle, 1 if enable else 0, ms, n, C.byref(cells)))
        return (ms[0], ms[1]), (n[0], n[1]), cells.value

    def close(self):
        if self.handle:
            self.lib.mh_block_destroy(self.handle)
            self.handle = None


class NativeBlockGroup:
    """All `world` blocks of the decomposition as objects of ONE process on one GPU (LOOPBACK backend, include/mara_hip.h)."""

    def __init__(self, global_shape, dl, gamma, plm_theta=1.5, riemann="hlle", rk_order=2, bc="outflow", world=8, device=0, chunk_rows=0,
                 arith="strict"):
        self.lib = L.load_library()
        self.world = world
        self.global_shape = tuple(int(n) for n in global_shape)
        self.handles = (C.c_void_p * world)()
        d = euler_cart_desc(self.global_shape, dl, gamma, plm_theta, riemann, bc, chunk_rows, arith)
        L.check(self.lib.mh_block_group_create(self.handles, C.byref(d), rk_order, world, device))
        self.members = [NativeBlock(self.global_shape, dl, gamma, handle=C.c_void_p(self.handles[r])) for r in range(world)]

    def upload(self, u_global):
        u = np.ascontiguousarray(u_global, dtype=np.float64)
        assert u.shape == self.global_shape + (5,), (u.shape, self.global_shape)
        L.check(self.lib.mh_block_group_upload(self.handles, self.world, u.ctypes.data_as(C.c_void_p)))

    def download(self):
        u = np.empty(self.global_shape + (5,))
        L.check(self.lib.mh_block_group_download(self.handles, self.world, u.ctypes.data_as(C.c_void_p)))
        return u

    def step(self, dt, nsteps=1):
        L.check(self.lib.mh_block_group_step(self.handles, self.world, dt, nsteps))

    def synchronize(self):
        for m in self.members:
            m.synchronize()

    def status(self):
        bits, first = 0, None
        for m in self.members:
            b, f = m.status()
            bits |= b
            if b:
                first = f if first is None else min(first, f)
        return bits, first

    def close(self):
        for m in self.members:
            m.close()
        self.members = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""Slab decomposition of the structured grid over the GPUs of one node, one process per GPU.

The cut is exactly the reference's thread-slab formula (nd::partition_shape, src/core_ndarray.hpp:820-836,
used by mara::evaluate_on<N>, src/app_parallel.hpp:75-103): rank n of N owns axis-0 rows
[n*Ni/N, (n+1)*Ni/N). The reference never exchanges anything (its slabs share one address space); here each
Runge-Kutta stage is followed by a two-row ghost exchange with the axis-0 neighbours as point-to-point
send/recv (RCCL over xGMI under torch.distributed's "nccl" backend), issued as ONE group per stage and
overlapped with the interior update, which runs on the main stream while the edge rows and the exchange run
on a second, high-priority stream.

This module owns no arithmetic. The stage kernel is libmara_hip.so's mh_euler_cart_stage (C ABI); tests may
inject another `stage_fn` to exercise the exchange logic on CPU with the gloo backend.
"""
import ctypes as C
import torch
import torch.distributed as dist
from . import _lib as L

HALO = 2
NQ = 5


def slab_fingerprint(t):
    """Order-independent fingerprint of a slab's bits: [wrapping sum, xor] of its doubles viewed as int64 (bench.py compares the
    ranks' slabs with a one-GPU run through it)."""
    v = t.contiguous().view(torch.int64).reshape(-1)
    x = v.clone()
    while x.numel() > 1:                      # xor reduction by halving
        h = x.numel() // 2
        x = torch.cat([x[:h] ^ x[h:2 * h], x[2 * h:]])
    return [int(v.sum().item()), int(x[0].item()) if x.numel() else 0]


def partition_rows(count, nparts, part):
    """Rows [start, final) of slab `part` (calls the C ABI's host-side mh_partition_rows)."""
    lib = L.load_library()
    a, b = C.c_size_t(), C.c_size_t()
    lib.mh_partition_rows(count, nparts, part, C.byref(a), C.byref(b))
    return a.value, b.value


class HipStage:
    """Default stage function: launches the HIP kernel on the current torch stream through the C ABI."""

    def __init__(self, desc):
        self.lib = L.load_library()
        self.desc = desc
        self.status = torch.zeros(2, dtype=torch.int32, device="cuda")

    def __call__(self, u_in, u_base, u_out, dt, weight, row_ranges):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        base = C.c_void_p(u_base.data_ptr()) if u_base is not None else None
        for (a, b) in row_ranges:
            if b > a:
                L.check(self.lib.mh_euler_cart_stage(C.byref(self.desc), C.c_void_p(u_in.data_ptr()), base,
                                                      C.c_void_p(u_out.data_ptr()), dt, weight, a, b,
                                                      C.c_void_p(self.status.data_ptr()), stream))


def slab_plan(nrows_global, world, rank, periodic, self_exchange=False, rk_order=2, fused_cut=False):
    """mh_slab_plan_make (csrc/slab_plan.hpp, host code - callable without a GPU): the ONE place that decides a rank's rows, its neighbours,
    the ghost rows that travel and the messages of an exchange in issue order - for the native stepper and for this module alike."""
    p = L.SlabPlan()
    L.check(L.load_library().mh_slab_plan_make(int(nrows_global), int(world), int(rank), 1 if periodic else 0, 1 if self_exchange else 0,
                                              int(rk_order), 1 if fused_cut else 0, C.byref(p)))
    return p


class TorchDistExchange:
    """Ghost-row exchange with the axis-0 neighbours over torch.distributed (nccl = RCCL on GPUs, gloo on CPU): the messages of the native
    plan, in its order - the same pairing the library's exchange_rccl issues (csrc/slab.hip)."""

    def __init__(self, rank, world, periodic, group=None, self_exchange=False):
        """self_exchange: with a periodic domain on ONE rank, wrap around by sending to oneself instead of letting the
        kernel copy the rows locally - exercises the whole send/recv + overlap machinery on a single GPU."""
        self.rank, self.world, self.group = rank, world, group
        self.periodic, self.self_exchange = bool(periodic), bool(self_exchange)
        self.bind(4 * world, 2)          # neighbours are known at once; the rows of the messages once the stepper has bound the grid

    def bind(self, nrows_global, rk_order):
        """the plan of THIS grid: rows, neighbours and the messages of an exchange in issue order, as csrc/slab_plan.hpp decides them"""
        self.plan = slab_plan(nrows_global, self.world, self.rank, self.periodic, self.self_exchange, rk_order, False)
        self.lo = self.plan.lo if self.plan.lo >= 0 else None
        self.hi = self.plan.hi if self.plan.hi >= 0 else None
        return self.plan

    def start(self, field, n0):
        """Post the plan's sends of the edge row-blocks and receives into the ghost row-blocks (rows of the plan are local: row r of the slab is
        field[r + HALO]; ghost rows are negative / >= n0)."""
        ops = []
        # field: [n0 + 4, NQ, pitch]; row-blocks are contiguous: one message per neighbour and direction. Tags pair the two messages that
        # one pair of ranks may exchange in a step (lo == hi on a periodic axis of two ranks): a block of LOW rows always lands in HIGH ghosts.
        for k in range(self.plan.nmsg):
            m = self.plan.msg[k]
            block = field[m.first_row + HALO:m.first_row + HALO + m.rows]
            low = m.first_row == 0 if m.send else m.first_row >= n0
            ops.append(dist.P2POp(dist.isend if m.send else dist.irecv, block, m.peer, self.group, tag=1 if low else 2))
        return dist.batch_isend_irecv(ops) if ops else []

    @staticmethod
    def finish(reqs):
        for r in reqs:
            r.wait()


class SlabEulerStepper:
    """One rank's share of a uniform-cartesian 2-D Euler run: fields, stages, ghost exchange.

    global_shape = (N0, N1); this rank owns rows partition_rows(N0, world, rank). `bc` is the physical boundary
    condition ("outflow" | "periodic") of the global domain. Fields live on `device` in the library's device
    layout [n0 + 4, 5, n1] (include/mara_hip.h)."""

    def __init__(self, global_shape, dl, gamma, plm_theta=1.5, riemann="hllc", rk_order=2, bc="outflow",
                 rank=0, world=1, device="cuda", stage_fn=None, exchange=None, overlap=True, chunk_rows=0,
                 edge_chunk_rows=8, arith="strict"):
        self.global_shape = tuple(global_shape)
        self.rank, self.world = rank, world
        periodic = bc == "periodic"
        # rows, neighbours, message order: the native plan (two launches per stage here: two ghost rows after every stage)
        if exchange is None:
            exchange = TorchDistExchange(rank, world, periodic)
        self.plan = exchange.bind(global_shape[0], rk_order) if hasattr(exchange, "bind") else slab_plan(global_shape[0], world, rank, periodic, False, rk_order, False)
        assert self.plan.ghost_rows == HALO
        self.row0, self.row1 = self.plan.row0, self.plan.row1
        self.n0, self.n1 = self.row1 - self.row0, global_shape[1]
        if self.n0 < 2 * HALO and world > 1:
            raise ValueError("slab of %d rows is thinner than two ghost layers" % self.n0)
        self.rk_order = rk_order
        self.device = torch.device(device)
        self.exchange = exchange
        has_lo = getattr(self.exchange, "lo", None) is not None
        has_hi = getattr(self.exchange, "hi", None) is not None

        d = L.EulerCartDesc()
        d.rank = 2
        d.n[0], d.n[1], d.n[2] = self.n0, self.n1, 1
        d.dl[0], d.dl[1], d.dl[2] = dl[0], dl[1], 1.0
        d.gamma, d.plm_theta = gamma, plm_theta
        d.riemann = {"hlle": L.RIEMANN_HLLE, "hllc": L.RIEMANN_HLLC}[riemann]
        phys = L.BC_PERIODIC if periodic else L.BC_OUTFLOW
        d.bc_transverse = phys
        d.bc_lo0 = L.BC_EXTERNAL if has_lo else phys
        d.bc_hi0 = L.BC_EXTERNAL if has_hi else phys
        d.arith = {"strict": L.ARITH_STRICT, "fast": L.ARITH_FAST}[arith]
        d.chunk_rows = chunk_rows
        self.desc = d
        self.edge_desc = L.EulerCartDesc.from_buffer_copy(d)
        self.edge_desc.chunk_rows = edge_chunk_rows
        self.has_neighbours = has_lo or has_hi
        self.edge_rows = edge_chunk_rows if self.has_neighbours else 0
        if 2 * self.edge_rows > self.n0:
            self.edge_rows = max(HALO, self.n0 // 2) if self.has_neighbours else 0

        if stage_fn is None:
            if self.device.type != "cuda":
                raise L.MaraHipError("SlabEulerStepper has no CPU path: pass device='cuda' (tests inject stage_fn)")
            self.stage = HipStage(self.desc)
            self.edge_stage = HipStage(self.edge_desc)
            self.edge_stage.status = self.stage.status
        else:
            self.stage = self.edge_stage = stage_fn
        self.overlap = overlap and self.has_neighbours and self.device.type == "cuda"
        self.side = torch.cuda.Stream(device=self.device, priority=-1) if self.overlap else None

        self.timers = None   # list of (stage_weight, start_event, end_event) when enabled
        self.graph, self.graph_dt = None, None
        shape = (self.n0 + 2 * HALO, NQ, self.n1)
        self.u = torch.zeros(shape, dtype=torch.float64, device=self.device)
        self.scratch = torch.zeros(shape, dtype=torch.float64, device=self.device)

    # ---- data movement ---------------------------------------------------
    def load_slab(self, u_aos_slab):
        """u_aos_slab: host/any tensor or ndarray [n0][n1][5] of this rank's rows (reference host order)."""
        t = torch.as_tensor(u_aos_slab, dtype=torch.float64).to(self.device)
        assert tuple(t.shape) == (self.n0, self.n1, NQ), (tuple(t.shape), (self.n0, self.n1, NQ))
        self.u[HALO:HALO + self.n0] = t.permute(0, 2, 1)
        self.fill_ghosts(self.u)

    def slab(self):
        """This rank's rows as a host-order AoS tensor [n0][n1][5] on the compute device."""
        return self.u[HALO:HALO + self.n0].permute(0, 2, 1).contiguous()

    def fill_ghosts(self, f):
        """Physical ghost rows from the interior; neighbour ghost rows by exchange (initial condition only)."""
        self.fill_ghosts_physical_only(f)
        if self.has_neighbours:
            self.exchange.finish(self.exchange.start(f, self.n0))

    def fill_ghosts_physical_only(self, f):
        n0 = self.n0
        d = self.desc
        if d.bc_lo0 == L.BC_OUTFLOW:
            f[0:HALO] = f[HALO:HALO + 1]
        if d.bc_hi0 == L.BC_OUTFLOW:
            f[n0 + HALO:n0 + 2 * HALO] = f[n0 + HALO - 1:n0 + HALO]
        if d.bc_lo0 == L.BC_PERIODIC:
            f[0:HALO] = f[n0:n0 + HALO]
        if d.bc_hi0 == L.BC_PERIODIC:
            f[n0 + HALO:n0 + 2 * HALO] = f[HALO:2 * HALO]

    # ---- one Runge-Kutta stage + ghost exchange ------------------------------
    def _launch_main(self, u_in, u_base, u_out, dt, weight, rows):
        """The bulk launch on the main stream, optionally bracketed by events (bench.py's roofline leg)."""
        if self.timers is None:
            self.stage(u_in, u_base, u_out, dt, weight, rows)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.stage(u_in, u_base, u_out, dt, weight, rows)
        e1.record()
        self.timers.append((weight, e0, e1))

    def compute_stage(self, u_in, u_base, u_out, dt, weight):
        """All rows of one stage, no exchange (single-process multi-slab tests drive the exchange themselves)."""
        n0, e = self.n0, self.edge_rows
        if e:
            self.edge_stage(u_in, u_base, u_out, dt, weight, [(0, e), (n0 - e, n0)])
        self.stage(u_in, u_base, u_out, dt, weight, [(e, n0 - e)])

    def _stage(self, u_in, u_base, u_out, dt, weight):
        n0, e = self.n0, self.edge_rows
        if not self.has_neighbours:
            self._launch_main(u_in, u_base, u_out, dt, weight, [(0, n0)])
            return
        if self.overlap:
            main = torch.cuda.current_stream(self.device)
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                self.edge_stage(u_in, u_base, u_out, dt, weight, [(0, e), (n0 - e, n0)])
                reqs = self.exchange.start(u_out, n0)
            self._launch_main(u_in, u_base, u_out, dt, weight, [(e, n0 - e)])
            with torch.cuda.stream(self.side):
                self.exchange.finish(reqs)
            main.wait_stream(self.side)
        else:
            self.edge_stage(u_in, u_base, u_out, dt, weight, [(0, e), (n0 - e, n0)])
            reqs = self.exchange.start(u_out, n0)
            self._launch_main(u_in, u_base, u_out, dt, weight, [(e, n0 - e)])
            self.exchange.finish(reqs)

    def capture(self, dt):
        """Capture one full time step (both RK stages, edge launches, the RCCL send/recv group and the
        two-stream overlap) into a HIP graph; step() then replays it. Removes the per-step host cost
        (8 launches + a P2P group issued from Python), which at 8 GPUs exceeds the ~60 us of device work per stage.
        Call after at least one eager step (RCCL communicators must exist before capture). RK2 only
        (RK1 swaps its two fields every step)."""
        if self.rk_order != 2:
            raise ValueError("graph capture is implemented for rk_order 2")
        if self.has_neighbours:
            # torch's process group cannot be captured on this stack (its watchdog polls events recorded in the
            # capturing stream; RCCL P2P inside hipStreamEndCapture segfaults): use NativeSlabStepper instead
            raise ValueError("graph capture with a ghost exchange is not supported by the torch stepper")
        assert self.device.type == "cuda"
        self.timers = None
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._step_eager(dt, 1)
        self.graph, self.graph_dt = graph, dt
        return graph

    def step(self, dt, nsteps=1):
        """nsteps full time steps. RK1: u <- advance(u). RK2: u <- u*0.5 + advance(advance(u))*0.5
        (src/subprog_cloud.cpp:682-695), the combine fused into the second stage and written in place."""
        if self.graph is not None and dt == self.graph_dt and self.timers is None:
            for _ in range(nsteps):
                self.graph.replay()
            return
        self._step_eager(dt, nsteps)

    def _step_eager(self, dt, nsteps=1):
        for _ in range(nsteps):
            if self.rk_order == 1:
                self._stage(self.u, None, self.scratch, dt, 1.0)
                self.u, self.scratch = self.scratch, self.u
            else:
                self._stage(self.u, None, self.scratch, dt, 1.0)
                self._stage(self.scratch, self.u, self.u, dt, 0.5)

    def status(self):
        st = getattr(self.stage, "status", None)
        if st is None:
            return 0
        v = int(st[0].item())
        st.zero_()
        return v



class CloudHipStage:
    """Stage function of the `cloud` path: mh_cloud_stage (C ABI) on the current torch stream."""

    def __init__(self, desc, geom, inflow):
        self.lib = L.load_library()
        self.desc, self.geom, self.inflow = desc, geom, inflow
        self.status = torch.zeros(2, dtype=torch.int32, device=geom.device)

    def __call__(self, u_in, u_base, u_out, dt, weight, row_ranges):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        base = C.c_void_p(u_base.data_ptr()) if u_base is not None else None
        for (a, b) in row_ranges:
            if b > a:
                L.check(self.lib.mh_cloud_stage(C.byref(self.desc), C.c_void_p(self.geom.data_ptr()), C.c_void_p(self.inflow.data_ptr()),
                                                C.c_void_p(u_in.data_ptr()), base, C.c_void_p(u_out.data_ptr()), dt, weight, a, b,
                                                C.c_void_p(self.status.data_ptr()), stream))


class SlabCloudStepper(SlabEulerStepper):
    """One rank's RADIAL slab of the `cloud` sub-program's grid (BASELINE config 4: slab decomposition + RCCL halo).

    Same cut, layout, exchange and stream overlap as SlabEulerStepper; the stage is CloudProblem::advance
    (src/subprog_cloud.cpp:511-584) on rows [row0, row1) of the global radial grid: rank 0 keeps the nozzle-inflow inner
    boundary, the last rank the zero-gradient outer one, cut sides are MH_BC_EXTERNAL (ghost rows from the neighbour).
    The kernel takes the GLOBAL radial vertices plus row_offset, so geometry factors are bit-identical to the single-domain
    run; physical boundaries are applied inside the kernel (no stored ghost rows to fill)."""

    def __init__(self, r_vertices, q_vertices, rk_order=1, plm_theta=1.2, temperature_floor=1e-8, gamma=4.0 / 3,
                 rank=0, world=1, device="cuda", stage_fn=None, exchange=None, overlap=True, chunk_rows=0, edge_chunk_rows=8,
                 arith="strict"):
        import numpy as np
        rv = np.ascontiguousarray(r_vertices, dtype=np.float64)
        qv = np.ascontiguousarray(q_vertices, dtype=np.float64)
        nr, nq = len(rv) - 1, len(qv) - 1
        self.global_shape = (nr, nq)
        self.rank, self.world = rank, world
        if exchange is None:
            exchange = TorchDistExchange(rank, world, False)
        self.plan = exchange.bind(nr, rk_order) if hasattr(exchange, "bind") else slab_plan(nr, world, rank, False, False, rk_order, False)
        self.row0, self.row1 = self.plan.row0, self.plan.row1
        self.n0, self.n1 = self.row1 - self.row0, nq
        if self.n0 < 2 * HALO and world > 1:
            raise ValueError("slab of %d rows is thinner than two ghost layers" % self.n0)
        self.rk_order = rk_order
        self.device = torch.device(device)
        self.exchange = exchange
        has_lo = getattr(self.exchange, "lo", None) is not None
        has_hi = getattr(self.exchange, "hi", None) is not None
        d = L.CloudDesc(nr=self.n0, nq=nq, nr_global=nr, row_offset=self.row0, gamma=gamma, plm_theta=plm_theta,
                        temperature_floor=temperature_floor, bc_lo0=L.BC_EXTERNAL if has_lo else L.BC_INFLOW,
                        bc_hi0=L.BC_EXTERNAL if has_hi else L.BC_OUTFLOW,
                        arith={"strict": L.ARITH_STRICT, "fast": L.ARITH_FAST}[arith], chunk_rows=chunk_rows)
        self.desc = d
        self.edge_desc = L.CloudDesc.from_buffer_copy(d)
        self.edge_desc.chunk_rows = edge_chunk_rows
        self.has_neighbours = has_lo or has_hi
        self.edge_rows = edge_chunk_rows if self.has_neighbours else 0
        if 2 * self.edge_rows > self.n0:
            self.edge_rows = max(HALO, self.n0 // 2) if self.has_neighbours else 0
        self.r_vertices, self.q_vertices = rv, qv
        self.inflow = torch.zeros((NQ, nq), dtype=torch.float64, device=self.device)       # [5][nq] primitives of the inner ghost row
        if stage_fn is None:
            if self.device.type != "cuda":
                raise L.MaraHipError("SlabCloudStepper has no CPU path: pass device='cuda' (tests inject stage_fn)")
            lib = L.load_library()
            geom = np.zeros(lib.mh_cloud_geometry_doubles(C.byref(d)))
            L.check(lib.mh_cloud_pack_geometry(C.byref(d), rv.ctypes.data_as(C.c_void_p), qv.ctypes.data_as(C.c_void_p), geom.ctypes.data_as(C.c_void_p)))
            self.geom = torch.from_numpy(geom).to(self.device)
            self.stage = CloudHipStage(self.desc, self.geom, self.inflow)
            self.edge_stage = CloudHipStage(self.edge_desc, self.geom, self.inflow)
            self.edge_stage.status = self.stage.status
        else:
            self.stage = self.edge_stage = stage_fn
        self.overlap = overlap and self.has_neighbours and self.device.type == "cuda"
        self.side = torch.cuda.Stream(device=self.device, priority=-1) if self.overlap else None
        self.timers = None
        self.graph, self.graph_dt = None, None
        shape = (self.n0 + 2 * HALO, NQ, self.n1)
        self.u = torch.zeros(shape, dtype=torch.float64, device=self.device)
        self.scratch = torch.zeros(shape, dtype=torch.float64, device=self.device)

    def set_inflow(self, inflow_prims):
        """Nozzle primitives [nq][5] of the inner ghost row at the step-start time (src/subprog_cloud.cpp:466-493); every rank
        may call it, only rank 0's kernel reads it."""
        t = torch.as_tensor(inflow_prims, dtype=torch.float64)
        assert tuple(t.shape) == (self.n1, NQ)
        self.inflow.copy_(t.t().contiguous())

    def fill_ghosts_physical_only(self, f):
        pass        # inflow / zero-gradient rows are formed inside the kernel from the nozzle row / the last real row

def euler_cart_desc(global_shape, dl, gamma, plm_theta, riemann, bc, chunk_rows=0, arith="strict", fuse=None, planar=None):
    """mh_euler_cart_desc of a WHOLE uniform-cartesian grid (the form mh_slab_create / mh_slab_group_create take)"""
    rank_ = len(global_shape)
    d = L.EulerCartDesc()
    d.rank = rank_
    for a in range(3):
        d.n[a] = global_shape[a] if a < rank_ else 1
        d.dl[a] = dl[a] if a < rank_ else 1.0
    d.gamma, d.plm_theta = gamma, plm_theta
    d.riemann = {"hlle": L.RIEMANN_HLLE, "hllc": L.RIEMANN_HLLC}[riemann]
    phys = L.BC_PERIODIC if bc == "periodic" else L.BC_OUTFLOW
    d.bc_lo0 = d.bc_hi0 = d.bc_transverse = phys
    d.arith = {"strict": L.ARITH_STRICT, "fast": L.ARITH_FAST}[arith]
    d.chunk_rows = chunk_rows
    d.fuse_stages = 0 if fuse is None else (1 if fuse else -1)
    d.planar = 0 if planar is None else (1 if planar else -1)
    return d


class NativeSlabStepper:
    """The native (C++/HIP/RCCL) slab stepper of libmara_hip.so: same cut, ghost layout and message pattern as
    SlabEulerStepper, but the exchange is RCCL called from the library and the whole step is one HIP graph.
    `comm_id` is the 128-byte RCCL unique id every rank must share (see `native_comm_id`)."""

    def __init__(self, global_shape, dl, gamma, plm_theta=1.5, riemann="hllc", rk_order=2, bc="outflow",
                 rank=0, world=1, comm_id=None, self_exchange=False, device=0, chunk_rows=0, arith="strict", handle=None, fuse=None, planar=None):
        """comm_id=None on a rank with neighbours defers the RCCL communicator to connect() (ncclCommInitRank is collective: the host
        first makes sure every rank got this far). handle: wrap a slab created elsewhere (a member of NativeSlabGroup)."""
        import numpy as np
        self.np = np
        self.lib = L.load_library()
        if handle is not None:
            self.handle = handle
        else:
            d = euler_cart_desc(global_shape, dl, gamma, plm_theta, riemann, bc, chunk_rows, arith, fuse, planar)
            self.handle = C.c_void_p()
            idbuf = C.create_string_buffer(bytes(comm_id), 128) if comm_id is not None else None
            L.check(self.lib.mh_slab_create(C.byref(self.handle), C.byref(d), rk_order, rank, world, idbuf,
                                            1 if self_exchange else 0, device))
        a, b = C.c_int(), C.c_int()
        L.check(self.lib.mh_slab_rows(self.handle, C.byref(a), C.byref(b)))
        self.row0, self.row1 = a.value, b.value
        self.n0, self.n1 = self.row1 - self.row0, global_shape[1]
        self.slab_shape = (self.n0,) + tuple(global_shape[1:]) + (NQ,)

    def connect(self, comm_id):
        L.check(self.lib.mh_slab_connect(self.handle, C.create_string_buffer(bytes(comm_id), 128)))

    def use_comm(self, comm):
        """lend the slab the process's communicator (NativeComm); not collective"""
        L.check(self.lib.mh_slab_use_comm(self.handle, comm.handle))
        self._comm = comm          # keeps it alive

    def load_slab(self, u_aos_slab):
        u = self.np.ascontiguousarray(u_aos_slab, dtype=self.np.float64)
        assert u.shape == self.slab_shape, (u.shape, self.slab_shape)
        L.check(self.lib.mh_slab_upload(self.handle, u.ctypes.data_as(C.c_void_p)))

    def is_planar(self):
        return bool(self.lib.mh_slab_is_planar(self.handle))

    def slab_host(self):
        u = self.np.empty(self.slab_shape)
        L.check(self.lib.mh_slab_download(self.handle, u.ctypes.data_as(C.c_void_p)))
        return u

    def step(self, dt, nsteps=1, graph=True):
        L.check(self.lib.mh_slab_step(self.handle, dt, nsteps, 1 if graph else 0))

    def synchronize(self):
        L.check(self.lib.mh_slab_synchronize(self.handle))

    def status(self):
        s = C.c_int32()
        L.check(self.lib.mh_slab_status_word(self.handle, C.byref(s)))
        return s.value

    def status_result(self):
        """(status bits, flat index in the GLOBAL host array of the first failing cell or None); clears the device word"""
        r = L.StepResult()
        L.check(self.lib.mh_slab_status(self.handle, C.byref(r)))
        return r.status, (None if r.status == 0 else int(r.first_bad_index))

    def profile(self, on=True):
        L.check(self.lib.mh_slab_profile_enable(self.handle, 1 if on else 0))

    def profile_read(self):
        ms, n, rows = (C.c_double * 2)(), (C.c_int * 2)(), C.c_int()
        L.check(self.lib.mh_slab_profile_read(self.handle, ms, n, C.byref(rows)))
        return (ms[0], ms[1]), (n[0], n[1]), rows.value

    def field_tensor(self):
        """The solution field as a torch tensor view is not needed by the product path; tests compare host copies."""
        raise NotImplementedError

    def close(self):
        if self.handle:
            self.lib.mh_slab_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def native_comm_id(rank, world, device=None):
    """Create the RCCL unique id on rank 0 and broadcast it over the default torch.distributed group."""
    lib = L.load_library()
    buf = C.create_string_buffer(128)
    if rank == 0:
        L.check(lib.mh_comm_unique_id(buf))
    if world == 1:
        return bytes(buf.raw)
    t = torch.tensor(list(buf.raw), dtype=torch.uint8, device=device if device is not None else "cpu")
    dist.broadcast(t, src=0)
    return bytes(t.cpu().tolist())


class NativeComm:
    """One RCCL communicator per process (mh_comm_create), lent to every stepper the process builds (use_comm): the collective
    ncclCommInitRank is entered once. `comm_id` as for NativeSlabStepper (native_comm_id). Collective over the ranks."""

    def __init__(self, comm_id, rank, world, device=0):
        self.lib = L.load_library()
        self.handle = C.c_void_p()
        self.rank, self.world, self.device = rank, world, device
        L.check(self.lib.mh_comm_create(C.byref(self.handle), C.create_string_buffer(bytes(comm_id), 128), rank, world, device))

    def close(self):
        if self.handle:
            self.lib.mh_comm_destroy(self.handle)
            self.handle = C.c_void_p()


class NativeSlabGroup:
    """All `world` slabs of a decomposition as objects of ONE process on one GPU, exchanging ghost rows through the native stepper's
    LOOPBACK backend (stream-ordered device-to-device copies under the event protocol of the RCCL ranks; include/mara_hip.h,
    mh_slab_group_*). Euler (2-D, 3-D) with global_shape / dl, or the `cloud` grid with r_vertices / q_vertices."""

    def __init__(self, global_shape=None, dl=None, gamma=5.0 / 3, plm_theta=1.5, riemann="hllc", rk_order=2, bc="outflow", world=2,
                 device=0, chunk_rows=0, arith="strict", r_vertices=None, q_vertices=None, temperature_floor=1e-8, devices=None, planar=None, fuse=None):
        """devices: one device id per member (mh_slab_group_create_on: ONE process driving several GPUs, receives as peer copies)
        fuse (`cloud`): mh_cloud_desc.fuse_stages - None = the one-launch RK2 step across the radial cuts where available (FAST, PLM, from 384
        rows per slab on), False = never, True = required"""
        import numpy as np
        self.np = np
        self.lib = L.load_library()
        self.world = world
        self.handles = (C.c_void_p * world)()
        ids = (C.c_int * world)(*[int(x) for x in devices]) if devices is not None else None
        assert devices is None or len(devices) == world
        self.cloud = r_vertices is not None
        if self.cloud:
            rv = np.ascontiguousarray(r_vertices, dtype=np.float64)
            qv = np.ascontiguousarray(q_vertices, dtype=np.float64)
            nr, nq = len(rv) - 1, len(qv) - 1
            d = L.CloudDesc(nr=nr, nq=nq, nr_global=nr, row_offset=0, gamma=gamma, plm_theta=plm_theta, temperature_floor=temperature_floor,
                            bc_lo0=L.BC_INFLOW, bc_hi0=L.BC_OUTFLOW, arith={"strict": L.ARITH_STRICT, "fast": L.ARITH_FAST}[arith],
                            chunk_rows=chunk_rows, planar=0 if planar is None else (1 if planar else -1),
                            fuse_stages=0 if fuse is None else (1 if fuse else -1))
            if ids is not None:
                L.check(self.lib.mh_slab_cloud_group_create_on(self.handles, C.byref(d), rv.ctypes.data_as(C.c_void_p), qv.ctypes.data_as(C.c_void_p),
                                                               rk_order, world, ids))
            else:
                L.check(self.lib.mh_slab_cloud_group_create(self.handles, C.byref(d), rv.ctypes.data_as(C.c_void_p), qv.ctypes.data_as(C.c_void_p),
                                                            rk_order, world, device))
            self.global_shape = (nr, nq)
        else:
            d = euler_cart_desc(global_shape, dl, gamma, plm_theta, riemann, bc, chunk_rows, arith, None, planar)
            if ids is not None:
                L.check(self.lib.mh_slab_group_create_on(self.handles, C.byref(d), rk_order, world, ids))
            else:
                L.check(self.lib.mh_slab_group_create(self.handles, C.byref(d), rk_order, world, device))
            self.global_shape = tuple(global_shape)
        self.rows = []
        for r in range(world):
            a, b = C.c_int(), C.c_int()
            L.check(self.lib.mh_slab_rows(C.c_void_p(self.handles[r]), C.byref(a), C.byref(b)))
            self.rows.append((a.value, b.value))

    def upload(self, u_aos_global):
        u = self.np.ascontiguousarray(u_aos_global, dtype=self.np.float64)
        assert u.shape == self.global_shape + (NQ,), (u.shape, self.global_shape)
        L.check(self.lib.mh_slab_group_upload(self.handles, self.world, u.ctypes.data_as(C.c_void_p)))

    def download(self):
        u = self.np.empty(self.global_shape + (NQ,))
        L.check(self.lib.mh_slab_group_download(self.handles, self.world, u.ctypes.data_as(C.c_void_p)))
        return u

    def launches_per_step(self):
        """per member: 1 where its RK2 step is one fused launch (across its cuts), else the number of RK stages"""
        return [int(self.lib.mh_slab_launches_per_step(C.c_void_p(self.handles[r]))) for r in range(self.world)]

    def is_planar(self):
        """True while every member's launches take the planar kernels (descriptor field `planar`)"""
        return all(bool(self.lib.mh_slab_is_planar(C.c_void_p(self.handles[r]))) for r in range(self.world))

    def set_inflow(self, inflow_prims):
        p = self.np.ascontiguousarray(inflow_prims, dtype=self.np.float64)
        assert p.shape == (self.global_shape[1], NQ)
        # every member learns of the row in one call (the group's planarity is resolved over all members' rows and the row together)
        L.check(self.lib.mh_slab_group_set_inflow(self.handles, self.world, p.ctypes.data_as(C.c_void_p)))

    def step(self, dt, nsteps=1):
        L.check(self.lib.mh_slab_group_step(self.handles, self.world, dt, nsteps))

    def synchronize(self):
        for r in range(self.world):
            L.check(self.lib.mh_slab_synchronize(C.c_void_p(self.handles[r])))

    def status(self):
        """(OR of the members' status bits, smallest global flat index of a failing cell or None)"""
        bits, first = 0, None
        for r in range(self.world):
            res = L.StepResult()
            L.check(self.lib.mh_slab_status(C.c_void_p(self.handles[r]), C.byref(res)))
            bits |= res.status
            if res.status:
                first = int(res.first_bad_index) if first is None else min(first, int(res.first_bad_index))
        return bits, first

    def member_host(self, r):
        a, b = self.rows[r]
        u = self.np.empty((b - a,) + self.global_shape[1:] + (NQ,))
        L.check(self.lib.mh_slab_download(C.c_void_p(self.handles[r]), u.ctypes.data_as(C.c_void_p)))
        return u

    def close(self):
        for r in range(self.world):
            if self.handles[r]:
                self.lib.mh_slab_destroy(C.c_void_p(self.handles[r]))
                self.handles[r] = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

// Native slab stepper: one rank's share of a uniform-cartesian Euler run (2-D, 3-D) or of the `cloud` sub-program's
// spherical-polar SRHD grid (radial slabs), with the ghost-row exchange as RCCL point-to-point send/recv over xGMI,
// driven from C++ (HIP graph replay when the rank has no neighbours).
//
// Exchange backends. RCCL: one process per GPU, ncclSend/ncclRecv in one group per stage. LOOPBACK: the slabs of a GROUP
// live in one process (on one GPU) and a "receive" is a stream-ordered device-to-device copy out of the neighbour
// object's field, ordered by the same events; every other line of the stepper - cut, ghost layout, edge / interior
// split, staggered edges, event protocol - is shared. RCCL refuses two ranks on one device ("Duplicate GPU detected"),
// so the loopback group is how ranks with lo != hi, mixed physical / external sides and uneven cuts are executed and
// compared bit for bit with the single-domain run on a one-GPU box (tests/test_gpu_slab_group.py).
//
// Why native: at 8 GPUs a 4096^2 grid leaves ~60 us of device work per stage; issuing the launches and a P2P
// group per stage from a host language costs several times that. Here the whole multi-step loop - edge launch,
// the ncclSend/ncclRecv group and the interior launch on two streams - is issued from C++; without neighbours
// the step is additionally captured into one HIP graph and replayed.
//
// The slab cut is nd::partition_shape (src/core_ndarray.hpp:820-836): rank n of N owns axis-0 rows
// [n*Ni/N, (n+1)*Ni/N). The reference's slabs share an address space and exchange nothing; per stage each rank
// sends its two edge row-blocks (contiguous 2*5*n1 doubles in the device layout) to the axis-0 neighbours and
// receives their ghost row-blocks, all four operations in one RCCL group, overlapped with the interior update.
//
// RCCL is bound at run time (dlopen "librccl.so.1": inside a torch process that is the copy torch already
// loaded), so the library has no link-time dependency on it and single-GPU hosts never touch it.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <dlfcn.h>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#include "launch.hpp"
#include "slab_plan.hpp"
#include "rccl_api.hpp"

namespace mh {

RcclApi* rccl()
{
    static RcclApi api;
    static bool tried = false;
    if (! tried)
    {
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        {
            api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (api.handle)
        {
#define MH_SYM(field, sym) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, sym))
            MH_SYM(GetUniqueId, "ncclGetUniqueId");
            MH_SYM(CommInitRank, "ncclCommInitRank");
            MH_SYM(CommDestroy, "ncclCommDestroy");
            MH_SYM(GroupStart, "ncclGroupStart");
            MH_SYM(GroupEnd, "ncclGroupEnd");
            MH_SYM(Send, "ncclSend");
            MH_SYM(Recv, "ncclRecv");
            MH_SYM(AllReduce, "ncclAllReduce");
            MH_SYM(AllGather, "ncclAllGather");
            MH_SYM(GetErrorString, "ncclGetErrorString");
#undef MH_SYM
            if (! (api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.GroupStart && api.GroupEnd && api.Send && api.Recv && api.AllReduce && api.AllGather))
            {
                dlclose(api.handle);
                api.handle = nullptr;
            }
        }
    }
    return api.handle ? &api : nullptr;
}

int rccl_fail(ncclResult_t r, const char* what)
{
    RcclApi* a = rccl();
    set_error("RCCL error %d (%s) in %s", (int) r, a && a->GetErrorString ? a->GetErrorString(r) : "?", what);
    return MH_E_HIP;
}

} // namespace mh

using namespace mh;

enum { SLAB_EULER = 0, SLAB_CLOUD = 1 };
enum { EXCHANGE_NONE = 0, EXCHANGE_RCCL = 1, EXCHANGE_LOOPBACK = 2 };

struct mh_slab
{
    int kind = SLAB_EULER;
    int backend = EXCHANGE_NONE;
    int device = 0, rank = 0, world = 1, rk_order = 2;
    int lo = -1, hi = -1;                 // neighbour ranks on the low / high side of axis 0 (-1: physical boundary)
    int row0 = 0, row1 = 0, n0 = 0, n1 = 0, edge_rows = 0;
    int stagger = 0;                  // stages per stagger period (0: every stage synchronises both ways), see stage_begin
    int phase = 0;                    // stage index within the period
    int test_delay = 0;               // MH_SLAB_TEST_DELAY, see slab_test_delay_kernel
    bool event_on_launch = true;      // the stage kernels signal the cross-stream events themselves (hipExtLaunchKernel's stopEvent)
    mh_euler_cart_desc desc, edge_desc;
    mh_slab_plan plan = {};                      // rows, neighbours, ghost rows and the messages of an exchange in issue order (slab_plan.hpp)
    mh_cloud_desc cloud, cloud_edge;             // SLAB_CLOUD: this rank's rows of the global radial grid
    double* geom = nullptr;                      // SLAB_CLOUD: packed geometry of the GLOBAL grid (mh_cloud_pack_geometry)
    double* inflow = nullptr;                    // SLAB_CLOUD: [5][nq] nozzle primitives (read by the rank that owns row 0)
    double* field[2] = {nullptr, nullptr};       // [0] solution, [1] stage scratch; layout of include/mara_hip.h
    double* staging = nullptr;
    int32_t* status = nullptr;
    hipStream_t main = nullptr, side = nullptr;
    hipEvent_t ev_edge = nullptr, ev_interior = nullptr, join = nullptr;
    ncclComm_t comm = nullptr;
    bool owns_comm = true;                       // false: borrowed from an mh_comm (mh_slab_use_comm)
    mh_slab* peer_lo = nullptr;                  // EXCHANGE_LOOPBACK: the neighbour objects
    mh_slab* peer_hi = nullptr;
    hipEvent_t ev_copied = nullptr;              // EXCHANGE_LOOPBACK: this slab's copies out of its peers' rows have completed
    double* cur_out = nullptr;                   // output field of the stage being issued (read by the peers' loopback copies)
    bool skew_pending = false;                   // group member r >= 1: its first interior launch after an upload starts behind member r-1's
    hipGraphExec_t exec = nullptr;
    double graph_dt = 0.0;
    // fused RK2 (euler2d_fused.hip): one launch per step from field[0] into field[1], then the two swap. One captured step per direction.
    bool fused = false;
    // ... and WITH neighbours (round 3): the slab keeps FOUR rows of each neighbour (two per RK stage) and exchanges once per step - the
    // fused launch recomputes the neighbours' first-stage rows it needs instead of receiving them between two launches (group_fused_cut_step)
    bool fused_cut = false;
    mh_euler_cart_desc fused_desc;               // desc with the caller's chunk_rows (desc.chunk_rows is tuned for the two-launch interior)
    size_t pad_doubles = 0;                      // doubles allocated in front of (and behind) each field for rows -4, -3 (n0 + 2, n0 + 3)
    double* fused_src[2] = {nullptr, nullptr};
    hipGraphExec_t fused_exec[2] = {nullptr, nullptr};
    // planarity of the 2-D Euler field (mh_euler_cart_desc.planar): the caller's request, what this slab's own rows showed at the last upload, and
    // what the fused launches are told (desc.planar / fused_desc.planar = +1 / -1)
    // (`cloud`: the same for the azimuthal momentum, mh_cloud_desc.planar; inflow_planar = the nozzle row last handed in has no azimuthal velocity)
    int planar_request = 0;
    bool planar_local = false, planar_now = false, inflow_planar = true;
    int32_t* planar_flag = nullptr;
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events[2];   // bulk launches of stage 1 / stage 2
    std::vector<int> event_launches[2];                          // launches between the two events of each pair
    std::string error;
};

static int slab_fail(mh_slab* s, int code) { if (s) s->error = mh_last_error(nullptr); return code; }
static bool has_neighbours(const mh_slab* s) { return s->lo >= 0 || s->hi >= 0; }

// one launch of the stage kernel over rows [a, b) (and, 2-D Euler only, a second range [a2, b2) in the same launch)
static hipError_t stage_launch(mh_slab* s, bool edge, const double* in, const double* base, double* out, double dt, double w,
                               int a, int b, int a2, int b2, hipStream_t stream, LaunchEvents ev)
{
    if (s->kind == SLAB_CLOUD)
    {
        const mh_cloud_desc* d = edge ? &s->cloud_edge : &s->cloud;
        if (s->fused_cut)
        {
            // the whole RK2 step of these rows in one launch (cloud_fused.hip across the cuts): `in` is the step-start field, `base` / w unused
            if (hipError_t e = cloud_fused_rk2_launch_rows(&s->cloud, s->geom, s->inflow, in, out, dt, a, b, a2, b2, s->status, stream, true,
                                                           edge ? 0 : 2 * cloud_fused_rk2_blocks_per_chunk(&s->cloud))) return e;
            if (ev.stop) return hipEventRecord(ev.stop, stream);
            return hipSuccess;
        }
        if (hipError_t e = cloud_stage_launch(d, s->geom, s->inflow, in, base, out, dt, w, a, b, s->status, stream)) return e;
        if (b2 > a2) if (hipError_t e = cloud_stage_launch(d, s->geom, s->inflow, in, base, out, dt, w, a2, b2, s->status, stream)) return e;
    }
    else
    {
        const mh_euler_cart_desc* d = edge ? &s->edge_desc : &s->desc;
        // (the interior launch of a slab with neighbours is told how many workgroups its edge launch - two segments of one chunk each, issued
        // first on the high-priority stream - holds when it starts: it then ends in shorter chunks, euler2d_fused.hip: TAPER)
        if (s->fused_cut) return euler2d_fused_rk2_launch_rows(&s->fused_desc, in, out, dt, a, b, a2, b2, s->status, stream, ev, true,
                                                               edge ? 0 : 2 * euler2d_fused_rk2_blocks_per_chunk(&s->fused_desc));
        if (d->rank == 2) return euler2d_stage_launch2(d, in, base, out, dt, w, a, b, a2, b2, s->status, stream, ev);
        if (hipError_t e = euler3d_stage_launch(d, in, base, out, dt, w, a, b, s->status, stream)) return e;
        if (b2 > a2) if (hipError_t e = euler3d_stage_launch(d, in, base, out, dt, w, a2, b2, s->status, stream)) return e;
    }
    // launchers that do not carry events: record them behind the launch
    if (ev.stop) return hipEventRecord(ev.stop, stream);
    return hipSuccess;
}
static bool launch_carries_events(const mh_slab* s) { return s->kind == SLAB_EULER && s->desc.rank == 2; }

static int ghost_rows(const mh_slab* s) { return s->plan.ghost_rows; }
static size_t row_doubles(const mh_slab* s) { return (size_t) 5 * s->n1; }
// row r of a field (r = -2: its first stored ghost row; a fused slab with neighbours also owns rows -4, -3 and n0 + 2, n0 + 3)
static double* row_ptr(const mh_slab* s, double* f, long r) { return f + (r + 2) * (long) row_doubles(s); }
static size_t ghost_block_doubles(const mh_slab* s) { return (size_t) ghost_rows(s) * row_doubles(s); }      // the rows of one side, all variables: contiguous

static int exchange_rccl(mh_slab* s, double* f, hipStream_t stream)
{
    RcclApi* a = rccl();
    if (! a || ! s->comm) { set_error("mh_slab: neighbours exist but the RCCL communicator was not connected (mh_slab_connect)"); return MH_E_STATE; }
    // the messages of the plan, in its order (slab_plan.hpp: sends first, receives mirroring the neighbours' send order - matters when lo == hi),
    // one contiguous block of all variables per neighbour and direction, all in one RCCL group
    MH_RCCL_TRY(a->GroupStart());
    for (int k = 0; k < s->plan.nmsg; ++k)
    {
        const auto& m = s->plan.msg[k];
        const size_t count = (size_t) m.rows * row_doubles(s);
        if (m.send) MH_RCCL_TRY(a->Send(row_ptr(s, f, m.first_row), count, ncclDouble, m.peer, s->comm, stream));
        else        MH_RCCL_TRY(a->Recv(row_ptr(s, f, m.first_row), count, ncclDouble, m.peer, s->comm, stream));
    }
    MH_RCCL_TRY(a->GroupEnd());
    return MH_OK;
}

// LOOPBACK "receive": copy the neighbour object's edge rows of ITS current output field into this slab's ghost rows, on this slab's
// stream, after the neighbour's edge launch (its ev_edge, as recorded for the current stage: the group driver issues every slab's edge
// launch before any slab's exchange). peer_field == nullptr: use the peers' cur_out.
static int exchange_loopback(mh_slab* s, double* f, hipStream_t stream, bool initial)
{
    const size_t blk = ghost_block_doubles(s), bytes = blk * sizeof(double);
    const int G = ghost_rows(s);
    if (s->hi >= 0)
    {
        mh_slab* p = s->peer_hi;
        if (! p) { set_error("mh_slab: loopback peer missing"); return MH_E_STATE; }
        const double* src = row_ptr(p, initial ? p->field[0] : p->cur_out, 0);                           // peer rows 0 .. G-1
        if (! initial) MH_HIP_TRY(hipStreamWaitEvent(stream, p->ev_edge, 0));
        MH_HIP_TRY(hipMemcpyAsync(row_ptr(s, f, s->n0), src, bytes, hipMemcpyDeviceToDevice, stream));
    }
    if (s->lo >= 0)
    {
        mh_slab* p = s->peer_lo;
        if (! p) { set_error("mh_slab: loopback peer missing"); return MH_E_STATE; }
        const double* src = row_ptr(p, initial ? p->field[0] : p->cur_out, p->n0 - G);                   // peer rows n0-G .. n0-1
        if (! initial) MH_HIP_TRY(hipStreamWaitEvent(stream, p->ev_edge, 0));
        MH_HIP_TRY(hipMemcpyAsync(row_ptr(s, f, -G), src, bytes, hipMemcpyDeviceToDevice, stream));
    }
    MH_HIP_TRY(hipEventRecord(s->ev_copied, stream));
    return MH_OK;
}

static int slab_exchange(mh_slab* s, double* f, hipStream_t stream, bool initial)
{
    if (! has_neighbours(s)) return MH_OK;
    return s->backend == EXCHANGE_LOOPBACK ? exchange_loopback(s, f, stream, initial) : exchange_rccl(s, f, stream);
}

// Test instrument (MH_SLAB_TEST_DELAY, tests/test_gpu_rccl.py): one wave that sleeps a bounded ~150 us, queued in front of the edge
// launch (bit 0) and / or the interior launch (bit 1). It shifts the relative timing of the two chains by more than a stage, so a
// dependency that only held by luck of timing shows up as a wrong result on one GPU.
__global__ void slab_test_delay_kernel(int iters)
{
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(127);
}

// One Runge-Kutta stage of one slab, issued in two halves so that the slabs of a loopback group can be driven in lockstep
// (every slab's first half, then every slab's second half); a single slab simply runs them back to back.
struct StageArgs
{
    const double* in; const double* base; double* out; double dt, w; int which;
    int e = 0;       // edge rows per side of this stage (filled by stage_begin)
};

// Staggered edges (2-D): over a period of S stages the edge strips grow by two rows per stage (2, 4, .., 2S) and the interior
// shrinks accordingly. Stage k's interior [e_k, n0 - e_k) then reads only rows the previous interior wrote (k >= 1), and the
// first edge of the next period (rows 0,1) reads only rows 0..3, which the last edge wrote: per PERIOD the main stream waits for
// the side stream once (before interior 0); the side stream waits for the main stream before edges 1..S-1, off the interior's
// critical path. Hazards are checked in the comment at the waits.
//
// first half: stream waits, edge launch (side stream) - after it the rows that the neighbours need exist (ev_edge)
static int stage_begin(mh_slab* s, StageArgs& st)
{
    s->cur_out = st.out;
    if (! has_neighbours(s)) return MH_OK;
    const int S = s->stagger, k = S ? s->phase : 0;
    if (S) s->phase = (s->phase + 1) % S;
    const int n0 = s->n0, e = s->edge_rows * (k + 1);
    st.e = e;
    // Two dependency chains instead of a fork/join per stage:
    //   side stream:  edge(k) -> exchange(k)          edge(k) needs interior(k-1) [event] and exchange(k-1) [stream order]
    //   main stream:  interior(k)                      needs edge(k-1) [event] and interior(k-1) [stream order]
    // The interior rows [8, n0-8) read rows 6..n0-7 of the previous stage: edge and interior output, never ghost rows,
    // so the main stream does not wait for the exchange at all; its latency only sits on the (short) edge chain.
    // The waits are issued BEFORE the events are re-recorded, so they bind to the previous stage's records.
    // Staggered, stage k of the period, e_k = 2 (k + 1), buffers alternate (out(k) = in(k-1)):
    //   interior(k) reads in rows >= e_k - 2 = e_{k-1}: interior(k-1)'s, stream order; for k = 0 rows >= 0: edge(S-1)'s -> the one wait.
    //   interior(k) writes out rows >= e_k; edge(k-1) read that buffer up to row e_{k-1} + 1 < e_k: no overlap, no wait.
    //   edge(k) reads in rows up to e_k + 1: [0, e_{k-1}) edge(k-1)'s (stream order), the rest interior(k-1)'s -> wait (k >= 1); it
    //   overwrites out rows [0, e_k), of which interior(k-1) read rows >= e_{k-1} - 2 -> the same wait. For k = 0 it reads rows 0..3
    //   (edge(S-1)'s, stream order) and writes rows 0,1, while interior(S-1) reads rows >= 2S - 2 >= 2 of that buffer: no wait.
    if (! S || k == 0) MH_HIP_TRY(hipStreamWaitEvent(s->main, s->ev_edge, 0));
    if (! S || k >= 1) MH_HIP_TRY(hipStreamWaitEvent(s->side, s->ev_interior, 0));
    if (s->backend == EXCHANGE_LOOPBACK)
    {
        // edge(k) overwrites rows of out(k) = out(k-2)'s buffer, which the peers' copies of stage k-2 read; their copies of stage k-1
        // are later on the same streams, so waiting for those (the latest record of ev_copied) covers it
        if (s->peer_lo) MH_HIP_TRY(hipStreamWaitEvent(s->side, s->peer_lo->ev_copied, 0));
        if (s->peer_hi) MH_HIP_TRY(hipStreamWaitEvent(s->side, s->peer_hi->ev_copied, 0));
    }
    if (s->test_delay & 1) hipLaunchKernelGGL(slab_test_delay_kernel, dim3(1), dim3(64), 0, s->side, 40);
    if (s->test_delay & 2) hipLaunchKernelGGL(slab_test_delay_kernel, dim3(1), dim3(64), 0, s->main, 40);
    // both edge strips; the event rides on the launch where the launcher can carry it, else it is recorded behind it
    LaunchEvents ev;
    const bool on_launch = launch_carries_events(s) && s->event_on_launch;
    if (on_launch) ev.stop = s->ev_edge;
    MH_HIP_TRY(stage_launch(s, true, st.in, st.base, st.out, st.dt, st.w, 0, e, n0 - e, n0, s->side, ev));
    if (! on_launch) MH_HIP_TRY(hipEventRecord(s->ev_edge, s->side));
    return MH_OK;
}

// second half: ghost exchange (side stream) and the interior launch (main stream)
static int stage_finish(mh_slab* s, StageArgs& st)
{
    const int n0 = s->n0, e = st.e;
    const bool alone = ! has_neighbours(s);
    std::pair<hipEvent_t, hipEvent_t> pe;
    // profiling events of the bulk launch. Without neighbours (2-D) they ride on the launch itself (hipExtLaunchKernel's start / stop
    // events: the dispatch packet's own timestamps, no marker packets between consecutive stage kernels of the timed region); with
    // neighbours the stop slot of the launch belongs to the stream-ordering event, so they are recorded around it.
    const bool profile_on_launch = s->profile && alone && launch_carries_events(s);
    if (! alone) if (int rc = slab_exchange(s, st.out, s->side, false)) return rc;
    if (s->skew_pending)
    {
        // Band skew (loopback groups on ONE device): member r's first interior launch waits for member r-1's, so that from then on the
        // bands run about one stage apart - a first-stage (issue-bound) launch of one band overlaps a second-stage (bandwidth-bound)
        // launch of its neighbour instead of a launch of its own kind. The ghost exchange bounds the distance to one stage either way.
        s->skew_pending = false;
        if (s->peer_lo && s->peer_lo->rank < s->rank) MH_HIP_TRY(hipStreamWaitEvent(s->main, s->peer_lo->ev_interior, 0));
    }
    LaunchEvents ev;
    const bool on_launch = ! alone && launch_carries_events(s) && s->event_on_launch;
    if (on_launch) ev.stop = s->ev_interior;
    if (s->profile)
    {
        hipEventCreate(&pe.first);
        hipEventCreate(&pe.second);
        if (profile_on_launch) { ev.start = pe.first; ev.stop = pe.second; }
        else                   hipEventRecord(pe.first, s->main);
    }
    const hipError_t be = stage_launch(s, false, st.in, st.base, st.out, st.dt, st.w, e, n0 - e, 0, 0, s->main, ev);
    if (s->profile)
    {
        if (! profile_on_launch) hipEventRecord(pe.second, s->main);
        s->events[st.which].push_back(pe);
        s->event_launches[st.which].push_back(1);
    }
    MH_HIP_TRY(be);
    if (! alone && ! on_launch) MH_HIP_TRY(hipEventRecord(s->ev_interior, s->main));
    return MH_OK;
}

// order the main stream after everything queued on the side stream (before a download / status read / return to the host)
static int slab_join(mh_slab* s)
{
    if (! has_neighbours(s)) return MH_OK;
    MH_HIP_TRY(hipEventRecord(s->join, s->side));
    MH_HIP_TRY(hipStreamWaitEvent(s->main, s->join, 0));
    return MH_OK;
}

// stage `i` (0 or 1) of a step: RK1 = one stage + field swap; RK2: u1 = advance(u0); u = u0 * 0.5 + advance(u1) * 0.5 in place over u0
static StageArgs stage_args(const mh_slab* s, int i, double dt)
{
    StageArgs st;
    if (i == 0) { st.in = s->field[0]; st.base = nullptr; st.out = s->field[1]; st.w = 1.0; }
    else        { st.in = s->field[1]; st.base = s->field[0]; st.out = s->field[0]; st.w = 0.5; }
    st.dt = dt; st.which = s->fused_cut ? 1 : i;          // (a fused step's launch is reported in the second-stage slot, as without neighbours)
    return st;
}

static int group_one_step(mh_slab** g, int n, double dt)
{
    // Fused across cuts: the whole RK2 step is ONE "stage" of the two-chain schedule - both edge strips (four rows each: what the
    // neighbours need for BOTH of their stages) in one fused launch on the side stream with the step's one exchange behind it, the rest of
    // the rows in a fused launch on the main stream, which never waits for the exchange (its rows [4, n0 - 4) read rows 0 .. n0 - 1 only).
    const int nstages = g[0]->fused_cut ? 1 : g[0]->rk_order;
    StageArgs st[64];
    for (int i = 0; i < nstages; ++i)
    {
        for (int r = 0; r < n; ++r) { MH_HIP_TRY(hipSetDevice(g[r]->device)); st[r] = stage_args(g[r], i, dt); if (int rc = stage_begin(g[r], st[r])) return rc; }
        for (int r = 0; r < n; ++r) { MH_HIP_TRY(hipSetDevice(g[r]->device)); if (int rc = stage_finish(g[r], st[r])) return rc; }
    }
    if (nstages == 1) for (int r = 0; r < n; ++r) std::swap(g[r]->field[0], g[r]->field[1]);
    return MH_OK;
}

static int slab_one_step(mh_slab* s, double dt) { return group_one_step(&s, 1, dt); }

// a whole-field RK2 step as one launch (the descriptor's fuse_stages, include/mara_hip.h): lone 2-D Euler slabs with physical sides only
static bool slab_can_fuse(const mh_slab* s)
{
    return s->kind == SLAB_EULER && s->rk_order == 2 && ! has_neighbours(s) && s->desc.fuse_stages >= 0 && euler2d_fused_rk2_available(&s->desc);
}

// ... with neighbours. Every member of the decomposition must decide alike (the exchange moves four rows instead of two), so the
// decision rests on the global descriptor and the thinnest share of the rows only. Measured on one GPU with the exchange to self through
// RCCL, 4096 columns, us per step, two launches / fused across cuts (profiles/r03/slab_fused_across_cuts.jsonl): 2048 rows 380 / 332,
// 1024 rows 191 / 179, 512 rows 99 / 92 (with one pair per workgroup the fused form lost below 1536 rows; with two it wins down to the
// 512 rows of the 8-GPU cut). Thinner slabs were not measured and keep the two launches. MH_SLAB_FUSED_CUTS: 0 never, 1 wherever it can
// run (12 rows), unset: from 384 rows per slab.
static bool slab_can_fuse_cut(const mh_slab* s, const mh_euler_cart_desc* global, const mh_cloud_desc* cglobal)
{
    int least = 384;
    if (const char* v = getenv("MH_SLAB_FUSED_CUTS")) { if (atoi(v) == 0) return false; least = 12; }          /* 12: two edge strips of four rows and an interior */
    if (s->kind == SLAB_CLOUD)
    {
        // `cloud` across radial cuts (round 5; BASELINE config 4 is a 4-slab run): both RK stages use the step-start nozzle row
        // (src/subprog_cloud.cpp:466-493, :524), so the one-launch step of cloud_fused.hip serves a slab as it serves the whole field - four
        // ghost rows, one exchange per step; the nozzle rows apply on the slab that owns row 0 only. Same threshold as the Euler slabs
        // (measured: profiles/r05/cloud_fused_cuts.md).
        mh_cloud_desc d = s->cloud;
        return s->rk_order == 2 && has_neighbours(s) && cglobal && cglobal->fuse_stages >= 0 && cglobal->nr_global / s->world >= least
            && s->n0 >= 12 && cloud_fused_rk2_available(&d, true);
    }
    mh_euler_cart_desc d = s->desc;
    d.n[0] = 8;          // (the member's own row count must not enter: only the thinnest share below does)
    return s->kind == SLAB_EULER && s->rk_order == 2 && has_neighbours(s) && global->fuse_stages >= 0 && global->n[0] / s->world >= least
        && euler2d_fused_rk2_available(&d, true);
}

static int slab_fused_step(mh_slab* s, double dt)
{
    MH_HIP_TRY(euler2d_fused_rk2_launch(&s->desc, s->field[0], s->field[1], dt, s->status, s->main, LaunchEvents()));
    std::swap(s->field[0], s->field[1]);
    s->cur_out = s->field[0];
    return MH_OK;
}

static int check_group(mh_slab** g, int n)
{
    if (! g || n < 1 || n > 64) { set_error("mh_slab group: need 1..64 slabs"); return MH_E_INVALID; }
    for (int r = 0; r < n; ++r)
        if (! g[r] || g[r]->world != n || g[r]->rank != r || g[r]->rk_order != g[0]->rk_order || g[r]->kind != g[0]->kind
            || (n > 1 && g[r]->backend != EXCHANGE_LOOPBACK))
        { set_error("mh_slab group: slab %d is not member %d of a loopback group of %d", r, r, n); return MH_E_INVALID; }
    return MH_OK;
}

// everything of mh_slab_create but the exchange backend's connection
static int slab_create_common(mh_slab** out, int kind, const mh_euler_cart_desc* global, const mh_cloud_desc* cglobal,
                              const double* rv, const double* qv, int rk_order, int rank, int world, int self_exchange, int device_id, int backend)
{
    if (! out) return MH_E_INVALID;
    if (rank < 0 || rank >= world) { set_error("mh_slab: rank %d of %d", rank, world); return MH_E_INVALID; }
    if (rk_order != 1 && rk_order != 2) { set_error("rk_order must be 1 or 2"); return MH_E_INVALID; }
    bool periodic = false;
    int nrows_global = 0;
    if (kind == SLAB_EULER)
    {
        if (! global) return MH_E_INVALID;
        if (global->rank != 2 && global->rank != 3) { set_error("mh_slab: rank must be 2 or 3"); return MH_E_INVALID; }
        periodic = global->bc_lo0 == MH_BC_PERIODIC;
        if (periodic != (global->bc_hi0 == MH_BC_PERIODIC)) { set_error("periodic axis-0 bc must be set on both sides"); return MH_E_INVALID; }
        nrows_global = global->n[0];
    }
    else
    {
        if (! cglobal || ! rv || ! qv) return MH_E_INVALID;
        if (cglobal->nr != cglobal->nr_global || cglobal->row_offset != 0) { set_error("mh_slab cloud: pass the description of the WHOLE grid (nr == nr_global, row_offset 0)"); return MH_E_INVALID; }
        if (cglobal->nr < 2 || cglobal->nq < 3 || !(cglobal->gamma > 1.0)) { set_error("mh_slab cloud: need nr >= 2, nq >= 3, gamma > 1"); return MH_E_INVALID; }
        nrows_global = cglobal->nr_global;
    }
    MH_HIP_TRY(hipSetDevice(device_id));
    mh_slab* s = new mh_slab();
    s->kind = kind;
    s->device = device_id; s->rank = rank; s->world = world; s->rk_order = rk_order;
    // rows, neighbours and message order: slab_plan.hpp decides, for this stepper and for the torch.distributed one alike (ghost rows and
    // exchanges per step are settled below, once it is known whether the slab takes the one-launch step across its cuts)
    if (slab_plan_make(nrows_global, world, rank, periodic ? 1 : 0, self_exchange, rk_order, 0, &s->plan) != MH_OK)
    { delete s; set_error("mh_slab: %d rows cannot be cut into %d slabs", nrows_global, world); return MH_E_INVALID; }
    s->row0 = s->plan.row0; s->row1 = s->plan.row1; s->n0 = s->row1 - s->row0;
    s->lo = s->plan.lo; s->hi = s->plan.hi;
    s->backend = has_neighbours(s) ? backend : EXCHANGE_NONE;
    int edge = 2;
    if (kind == SLAB_EULER)
    {
        s->n1 = global->rank == 3 ? global->n[1] * global->n[2] : global->n[1];      // row pitch: cells per axis-0 row (plane in 3-D)
        s->desc = *global;
        s->desc.n[0] = s->n0;
        s->desc.bc_lo0 = s->lo >= 0 ? MH_BC_EXTERNAL : global->bc_lo0;
        s->desc.bc_hi0 = s->hi >= 0 ? MH_BC_EXTERNAL : global->bc_hi0;
        // Edge strips: the rows whose results are sent. In 2-D exactly the two ghost layers' worth (rows 0,1 and n0-2,n0-1): the edge
        // launch sits on the stage's critical chain (exchange(k-1) -> edge(k) -> exchange(k)) while it shares the SIMDs with the interior
        // launch, so its latency - rows per wave - is what matters: 8-row strips took 35 us per stage at 512 x 4096 per rank and made the
        // side chain, not the interior, set the step time (rocprofv3 kernel trace, scripts/slab_trace.py); 2-row strips take ~15 us.
        edge = global->rank == 2 ? s->plan.edge_rows : 8;          // (2-D: the plan's two; 3-D: planes are marched in chunks of eight)
        if (edge < 2) edge = 2;
        s->edge_desc = s->desc;
        s->edge_desc.chunk_rows = edge;
        s->edge_desc.tail_rows = -1;
    }
    else
    {
        s->n1 = cglobal->nq;
        s->cloud = *cglobal;
        s->cloud.nr = s->n0;
        s->cloud.row_offset = s->row0;
        s->cloud.bc_lo0 = s->lo >= 0 ? MH_BC_EXTERNAL : MH_BC_INFLOW;
        s->cloud.bc_hi0 = s->hi >= 0 ? MH_BC_EXTERNAL : MH_BC_OUTFLOW;
        s->cloud_edge = s->cloud;
        s->cloud_edge.chunk_rows = edge;
        s->cloud_edge.tail_rows = -1;
    }
    s->edge_rows = has_neighbours(s) ? edge : 0;
    if (2 * s->edge_rows > s->n0) s->edge_rows = s->n0 / 2;
    s->stagger = 4;          // measured at 512 / 1024 / 2048 rows per rank (scripts/slab_ab.py): 4 stages per period is ~1 us per step better than 2
    if (const char* v = getenv("MH_SLAB_STAGGER")) s->stagger = atoi(v);          // measurement switches, read once per slab
    if (kind != SLAB_EULER || global->rank != 2 || s->edge_rows != 2 || s->stagger < 2 || s->n0 < 4 * s->stagger + 4) s->stagger = 0;
    if (s->edge_rows > 0 && kind == SLAB_EULER && global->rank == 2 && global->chunk_rows == 0)
    {
        // The interior launch alone should fill one residency round of 2048 waves and no more (a handful of waves in a second
        // round run alone, see euler2d.hip's default). The concurrent edge launch is NOT counted: it has priority, squeezes in at
        // the start and is gone long before the round ends. Measured at 512 x 4096 per rank, exchange to self, us per step:
        // 18 rows per chunk (29 chunks, 2001 waves) 118; 21 (25 chunks) 125; 16 (32 chunks, 2208 waves) 139; 24 (22 chunks) 136.
        const long nstrips = (global->n[1] + 59) / 60;
        const long chunks_max = 2048 / nstrips;
        const long rows = s->n0 - 2 * s->edge_rows;
        if (chunks_max > 0)
        {
            const long c = (rows + chunks_max - 1) / chunks_max;
            if (c <= 96) s->desc.chunk_rows = (int) (c < 4 ? 4 : c);
        }
    }
    s->fused = slab_can_fuse(s);
    s->fused_cut = slab_can_fuse_cut(s, global, cglobal);
    slab_plan_make(nrows_global, world, rank, periodic ? 1 : 0, self_exchange, rk_order, s->fused_cut ? 1 : 0, &s->plan);      // four ghost rows once per step, or two per stage
    if (s->fused_cut)
    {
        s->fused_desc = s->desc;
        if (kind == SLAB_EULER) s->fused_desc.chunk_rows = global->chunk_rows;
        s->pad_doubles = 2 * row_doubles(s);
        s->edge_rows = s->plan.edge_rows;          // 4: the rows a neighbour needs for both of its stages; every step synchronises both chains (no stagger)
        s->stagger = 0;
    }
    if (kind == SLAB_CLOUD && cglobal->fuse_stages > 0 && has_neighbours(s) && ! s->fused_cut)
    {
        delete s;
        set_error("fuse_stages is required, but a fused `cloud` step across radial cuts needs MH_ARITH_FAST, PLM, rk_order 2 and twelve rows per slab");
        return MH_E_INVALID;
    }
    if (kind == SLAB_EULER && global->fuse_stages > 0 && ! s->fused && ! s->fused_cut)
    {
        delete s;
        set_error("fuse_stages is required, but a fused RK2 step needs MH_ARITH_FAST, PLM, rk_order 2, rank 2 (and, with neighbours, twelve rows per slab)");
        return MH_E_INVALID;
    }
    if (const char* v = getenv("MH_SLAB_EVENT_ON_LAUNCH")) s->event_on_launch = atoi(v) != 0;
    if (const char* v = getenv("MH_SLAB_TEST_DELAY")) s->test_delay = atoi(v);
    if (has_neighbours(s) && s->n0 < 4) { const int thin = s->n0; delete s; set_error("slab of %d rows is thinner than two ghost layers", thin); return MH_E_INVALID; }

    auto cleanup = [&] () { mh_slab_destroy(s); };
    if (hipStreamCreateWithFlags(&s->main, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithPriority(&s->side, hipStreamNonBlocking, -1) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_edge, hipEventDisableTiming) != hipSuccess ||            // followed by the send of those rows: system-scope release
        // ev_interior is consumed by the edge launches of this device only: device-scope release
        hipEventCreateWithFlags(&s->ev_interior, hipEventDisableTiming | hipEventReleaseToDevice) != hipSuccess ||
        hipEventCreateWithFlags(&s->ev_copied, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->join, hipEventDisableTiming) != hipSuccess)
    { cleanup(); set_error("mh_slab: stream/event creation failed"); return MH_E_HIP; }
    const size_t doubles = (size_t) 5 * (s->n0 + 4) * s->n1 + 2 * s->pad_doubles;
    for (auto& f : s->field)
    {
        double* base = nullptr;
        if (hipMalloc((void**) &base, doubles * sizeof(double)) != hipSuccess) { cleanup(); set_error("mh_slab: hipMalloc failed"); return MH_E_NOMEM; }
        hipMemsetAsync(base, 0, doubles * sizeof(double), s->main);
        f = base + s->pad_doubles;          // (freed as f - pad_doubles)
    }
    if (hipMalloc((void**) &s->status, 2 * sizeof(int32_t)) != hipSuccess) { cleanup(); return MH_E_NOMEM; }
    hipMemsetAsync(s->status, 0, 2 * sizeof(int32_t), s->main);
    s->planar_request = kind == SLAB_EULER ? global->planar : cglobal->planar;
    s->desc.planar = s->edge_desc.planar = s->fused_desc.planar = s->cloud.planar = s->cloud_edge.planar = -1;          // until an upload has looked at the field
    if (hipMalloc((void**) &s->planar_flag, sizeof(int32_t)) != hipSuccess) { cleanup(); return MH_E_NOMEM; }
    if (hipMalloc((void**) &s->staging, (size_t) 5 * s->n0 * s->n1 * sizeof(double)) != hipSuccess) { cleanup(); return MH_E_NOMEM; }
    if (kind == SLAB_CLOUD)
    {
        s->inflow_planar = false;          // no nozzle row seen yet: general kernels until mh_slab_set_inflow / mh_slab_group_set_inflow has looked at one
        std::vector<double> geom(mh_cloud_geometry_doubles(cglobal));
        if (int rc = mh_cloud_pack_geometry(cglobal, rv, qv, geom.data())) { cleanup(); return rc; }
        if (hipMalloc((void**) &s->geom, geom.size() * sizeof(double)) != hipSuccess ||
            hipMalloc((void**) &s->inflow, (size_t) 5 * s->n1 * sizeof(double)) != hipSuccess) { cleanup(); set_error("mh_slab cloud: hipMalloc failed"); return MH_E_NOMEM; }
        if (hipMemcpy(s->geom, geom.data(), geom.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset(s->inflow, 0, (size_t) 5 * s->n1 * sizeof(double)) != hipSuccess) { cleanup(); set_error("mh_slab cloud: geometry upload failed"); return MH_E_HIP; }
    }
    hipStreamSynchronize(s->main);
    *out = s;
    return MH_OK;
}

// device_ids == nullptr: every member on device_id. Otherwise member r lives on device_ids[r]: ONE process, one host thread, several
// GPUs - what the reference's evaluate_on<N> thread slabs become when the slabs are devices. The "receives" are then peer copies
// (peer access is enabled between neighbouring members' devices); events order streams of different devices as they do on one.
static int group_create(mh_slab** slabs, int kind, const mh_euler_cart_desc* global, const mh_cloud_desc* cglobal, const double* rv,
                        const double* qv, int rk_order, int world, int device_id, const int* device_ids)
{
    if (! slabs || world < 1 || world > 64) { set_error("mh_slab group: need 1..64 slabs"); return MH_E_INVALID; }
    for (int r = 0; r < world; ++r) slabs[r] = nullptr;
    for (int r = 0; r < world; ++r)
        if (int rc = slab_create_common(&slabs[r], kind, global, cglobal, rv, qv, rk_order, r, world, 0, device_ids ? device_ids[r] : device_id, EXCHANGE_LOOPBACK))
        {
            for (int q = 0; q < r; ++q) { mh_slab_destroy(slabs[q]); slabs[q] = nullptr; }
            return rc;
        }
    for (int r = 0; r < world; ++r)
    {
        slabs[r]->peer_lo = slabs[r]->lo >= 0 ? slabs[slabs[r]->lo] : nullptr;
        slabs[r]->peer_hi = slabs[r]->hi >= 0 ? slabs[slabs[r]->hi] : nullptr;
        // the branch behind hipDeviceEnablePeerAccess cannot be reached on a one-GPU box: CHECK builds only (-DMH_TEST_HOOKS, libmara_hip_check.so;
        // the product library reads no such variable) take it on request, as if a neighbour lived on a device that refuses peer access
#ifdef MH_TEST_HOOKS
        const bool refuse = getenv("MH_SLAB_TEST_PEER_FAIL") && atoi(getenv("MH_SLAB_TEST_PEER_FAIL")) != 0;
#else
        const bool refuse = false;
#endif
        for (mh_slab* p : {slabs[r]->peer_lo, slabs[r]->peer_hi})
            if (p && (p->device != slabs[r]->device || refuse))
            {
                hipError_t e = hipSetDevice(slabs[r]->device);
                if (e == hipSuccess) e = refuse ? hipErrorInvalidDevice : hipDeviceEnablePeerAccess(p->device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                {
                    const int rc = hip_fail(e, "hipDeviceEnablePeerAccess");      // the error text before the destructors' own HIP calls
                    const std::string text = mh_last_error(nullptr);
                    for (int q = 0; q < world; ++q) { mh_slab_destroy(slabs[q]); slabs[q] = nullptr; }
                    set_error("%s", text.c_str());
                    return rc;
                }
                (void) hipGetLastError();
            }
    }
    return MH_OK;
}

// what the fused launches of this slab are told about the third momentum from now on; a change makes the captured steps stale
static void slab_set_planar(mh_slab* s, bool planar)
{
    if (planar != s->planar_now)
    {
        for (auto& e : s->fused_exec) { if (e) hipGraphExecDestroy(e); e = nullptr; }
        if (s->exec) { hipGraphExecDestroy(s->exec); s->exec = nullptr; }
    }
    s->planar_now = planar;
    s->desc.planar = s->edge_desc.planar = s->fused_desc.planar = s->cloud.planar = s->cloud_edge.planar = planar ? 1 : -1;
}

static int slab_upload_rows(mh_slab* s, const double* u_aos_slab_host)
{
    MH_HIP_TRY(hipSetDevice(s->device));
    const size_t ncell = (size_t) s->n0 * s->n1;
    MH_HIP_TRY(hipMemcpyAsync(s->staging, u_aos_slab_host, ncell * 5 * sizeof(double), hipMemcpyHostToDevice, s->main));
    MH_HIP_TRY(aos_to_soa_launch(s->staging, s->field[0], 5, s->n0, (size_t) s->n1, s->main));
    // planarity of this slab's own rows (mh_euler_cart_desc.planar): one pass per upload, none per step
    s->planar_local = false;
    const bool has_planar_kernels = s->kind == SLAB_EULER ? s->desc.rank == 2 && s->desc.plm_theta >= 0.0 : s->cloud.plm_theta >= 0.0;
    if (has_planar_kernels && s->planar_request >= 0 && s->planar_flag)
    {
        int32_t nonzero = 0;
        MH_HIP_TRY(hipMemsetAsync(s->planar_flag, 0, sizeof(int32_t), s->main));
        MH_HIP_TRY(plane_nonzero_launch(s->field[0], 5, 3, s->n0, (size_t) s->n1, s->planar_flag, s->main, (s->kind == SLAB_EULER ? s->desc.arith : s->cloud.arith) == MH_ARITH_STRICT));
        MH_HIP_TRY(hipMemcpyAsync(&nonzero, s->planar_flag, sizeof nonzero, hipMemcpyDeviceToHost, s->main));
        MH_HIP_TRY(hipStreamSynchronize(s->main));
        s->planar_local = nonzero == 0;
        if (s->planar_request > 0 && ! s->planar_local) { set_error("upload: `planar` was asserted, but rows [%d, %d) carry a third momentum", s->row0, s->row1); return MH_E_INVALID; }
    }
    // physical ghost rows (cloud: none stored - inflow / zero-gradient rows are formed inside the kernel)
    if (s->kind == SLAB_EULER) MH_HIP_TRY(fill_ghost_rows_launch(s->field[0], 5, s->n0, (size_t) s->n1, s->desc.bc_lo0, s->desc.bc_hi0, s->main));
    return MH_OK;
}

// both dependency chains start from "everything done"
static int slab_reset_chains(mh_slab* s)
{
    MH_HIP_TRY(hipSetDevice(s->device));
    MH_HIP_TRY(hipStreamSynchronize(s->main));
    MH_HIP_TRY(hipStreamSynchronize(s->side));
    s->phase = 0;
    MH_HIP_TRY(hipEventRecord(s->ev_interior, s->main));
    MH_HIP_TRY(hipEventRecord(s->ev_edge, s->side));
    return MH_OK;
}

extern "C" {

int mh_comm_unique_id(void* id128)
{
    RcclApi* a = rccl();
    if (! a) { set_error("librccl.so.1 could not be loaded"); return MH_E_STATE; }
    ncclUniqueId id;
    MH_RCCL_TRY(a->GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof id);
    return MH_OK;
}

int mh_slab_connect(mh_slab* s, const void* comm_id128)
{
    if (! s) return MH_E_INVALID;
    if (! has_neighbours(s) || s->comm) return MH_OK;
    if (s->backend != EXCHANGE_RCCL) { set_error("mh_slab_connect: not an RCCL slab"); return slab_fail(s, MH_E_STATE); }
    RcclApi* api = rccl();
    if (! api) { set_error("librccl.so.1 could not be loaded"); return slab_fail(s, MH_E_STATE); }
    if (! comm_id128) { set_error("mh_slab: neighbours exist but no RCCL unique id was given"); return slab_fail(s, MH_E_INVALID); }
    MH_HIP_TRY(hipSetDevice(s->device));
    ncclUniqueId id;
    std::memcpy(&id, comm_id128, sizeof id);
    ncclResult_t r = api->CommInitRank(&s->comm, s->world, id, s->rank);
    if (r != ncclSuccess) return slab_fail(s, rccl_fail(r, "ncclCommInitRank"));
    return MH_OK;
}

int mh_comm_create(mh_comm** out, const void* comm_id128, int rank, int world, int device_id)
{
    if (! out || ! comm_id128 || world < 1 || rank < 0 || rank >= world) return MH_E_INVALID;
    RcclApi* api = rccl();
    if (! api) { set_error("librccl.so.1 could not be loaded"); return MH_E_STATE; }
    MH_HIP_TRY(hipSetDevice(device_id));
    ncclUniqueId id;
    std::memcpy(&id, comm_id128, sizeof id);
    mh_comm* c = new mh_comm;
    c->rank = rank; c->world = world; c->device = device_id;
    ncclResult_t r = api->CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { delete c; return rccl_fail(r, "ncclCommInitRank"); }
    *out = c;
    return MH_OK;
}

void mh_comm_destroy(mh_comm* c)
{
    if (! c) return;
    hipSetDevice(c->device);
    if (c->comm && rccl()) rccl()->CommDestroy(c->comm);
    delete c;
}

int mh_slab_use_comm(mh_slab* s, mh_comm* c)
{
    if (! s || ! c) return MH_E_INVALID;
    if (! has_neighbours(s)) return MH_OK;
    if (s->backend != EXCHANGE_RCCL) { set_error("mh_slab_use_comm: not an RCCL slab"); return slab_fail(s, MH_E_STATE); }
    if (s->comm) { set_error("mh_slab_use_comm: the slab has a communicator already"); return slab_fail(s, MH_E_STATE); }
    if (c->world != s->world || c->rank != s->rank || c->device != s->device)
    {
        set_error("mh_slab_use_comm: communicator is rank %d of %d on device %d, the slab rank %d of %d on device %d", c->rank, c->world, c->device, s->rank, s->world, s->device);
        return slab_fail(s, MH_E_INVALID);
    }
    s->comm = c->comm;
    s->owns_comm = false;
    return MH_OK;
}

int mh_slab_create(mh_slab** out, const mh_euler_cart_desc* global, int rk_order, int rank, int world,
                   const void* comm_id128, int self_exchange, int device_id)
{
    mh_slab* s = nullptr;
    if (int rc = slab_create_common(&s, SLAB_EULER, global, nullptr, nullptr, nullptr, rk_order, rank, world, self_exchange, device_id, EXCHANGE_RCCL)) return rc;
    // comm_id128 == NULL with neighbours: the caller connects later (mh_slab_connect), after every rank has agreed that creation succeeded
    if (comm_id128) if (int rc = mh_slab_connect(s, comm_id128)) { mh_slab_destroy(s); return rc; }
    *out = s;
    return MH_OK;
}

int mh_slab_cloud_create(mh_slab** out, const mh_cloud_desc* global, const double* r_vertices_host, const double* q_vertices_host,
                         int rk_order, int rank, int world, const void* comm_id128, int device_id)
{
    mh_slab* s = nullptr;
    if (int rc = slab_create_common(&s, SLAB_CLOUD, nullptr, global, r_vertices_host, q_vertices_host, rk_order, rank, world, 0, device_id, EXCHANGE_RCCL)) return rc;
    if (comm_id128) if (int rc = mh_slab_connect(s, comm_id128)) { mh_slab_destroy(s); return rc; }
    *out = s;
    return MH_OK;
}

int mh_slab_group_create(mh_slab** slabs, const mh_euler_cart_desc* global, int rk_order, int world, int device_id)
{
    return group_create(slabs, SLAB_EULER, global, nullptr, nullptr, nullptr, rk_order, world, device_id, nullptr);
}

int mh_slab_group_create_on(mh_slab** slabs, const mh_euler_cart_desc* global, int rk_order, int world, const int* device_ids)
{
    if (! device_ids) return MH_E_INVALID;
    return group_create(slabs, SLAB_EULER, global, nullptr, nullptr, nullptr, rk_order, world, 0, device_ids);
}

int mh_slab_cloud_group_create(mh_slab** slabs, const mh_cloud_desc* global, const double* r_vertices_host, const double* q_vertices_host,
                               int rk_order, int world, int device_id)
{
    return group_create(slabs, SLAB_CLOUD, nullptr, global, r_vertices_host, q_vertices_host, rk_order, world, device_id, nullptr);
}

int mh_slab_cloud_group_create_on(mh_slab** slabs, const mh_cloud_desc* global, const double* r_vertices_host, const double* q_vertices_host,
                                  int rk_order, int world, const int* device_ids)
{
    if (! device_ids) return MH_E_INVALID;
    return group_create(slabs, SLAB_CLOUD, nullptr, global, r_vertices_host, q_vertices_host, rk_order, world, 0, device_ids);
}

void mh_slab_destroy(mh_slab* s)
{
    if (! s) return;
    hipSetDevice(s->device);
    if (s->main) hipStreamSynchronize(s->main);
    if (s->side) hipStreamSynchronize(s->side);
    if (s->exec) hipGraphExecDestroy(s->exec);
    for (auto& g : s->fused_exec) if (g) hipGraphExecDestroy(g);
    if (s->comm && s->owns_comm && rccl()) rccl()->CommDestroy(s->comm);
    for (auto& v : s->events) for (auto& ev : v) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
    for (auto& f : s->field) if (f) hipFree(f - s->pad_doubles);
    if (s->staging) hipFree(s->staging);
    if (s->status) hipFree(s->status);
    if (s->planar_flag) hipFree(s->planar_flag);
    if (s->geom) hipFree(s->geom);
    if (s->inflow) hipFree(s->inflow);
    if (s->ev_edge) hipEventDestroy(s->ev_edge);
    if (s->ev_interior) hipEventDestroy(s->ev_interior);
    if (s->ev_copied) hipEventDestroy(s->ev_copied);
    if (s->join) hipEventDestroy(s->join);
    if (s->main) hipStreamDestroy(s->main);
    if (s->side) hipStreamDestroy(s->side);
    delete s;
}

int mh_slab_rows(const mh_slab* s, int* row0, int* row1)
{
    if (! s) return MH_E_INVALID;
    if (row0) *row0 = s->row0;
    if (row1) *row1 = s->row1;
    return MH_OK;
}

int mh_slab_upload(mh_slab* s, const double* u_aos_slab_host)
{
    if (! s || ! u_aos_slab_host) return MH_E_INVALID;
    if (s->backend == EXCHANGE_LOOPBACK) { set_error("mh_slab_upload: member of a loopback group (use mh_slab_group_upload)"); return slab_fail(s, MH_E_STATE); }
    if (int rc = slab_upload_rows(s, u_aos_slab_host)) return slab_fail(s, rc);
    // alone: what the rows showed; with neighbours in OTHER processes: only what the caller asserted for the whole grid (and these rows confirmed)
    slab_set_planar(s, s->planar_local && s->inflow_planar && (! has_neighbours(s) || s->backend == EXCHANGE_NONE || s->planar_request > 0));
    if (int rc = slab_exchange(s, s->field[0], s->main, true)) return slab_fail(s, rc);
    if (int rc = slab_reset_chains(s)) return slab_fail(s, rc);
    return MH_OK;
}

int mh_slab_group_upload(mh_slab** g, int n, const double* u_aos_global_host)
{
    if (int rc = check_group(g, n)) return rc;
    if (! u_aos_global_host) return MH_E_INVALID;
    for (int r = 0; r < n; ++r)
        if (int rc = slab_upload_rows(g[r], u_aos_global_host + (size_t) g[r]->row0 * g[r]->n1 * 5)) return slab_fail(g[r], rc);
    for (int r = 0; r < n; ++r) { MH_HIP_TRY(hipSetDevice(g[r]->device)); MH_HIP_TRY(hipStreamSynchronize(g[r]->main)); }
    {
        bool all = true;          // a loopback group sees every member's rows: planar only if the whole grid is
        for (int r = 0; r < n; ++r) all = all && g[r]->planar_local && g[r]->inflow_planar;
        for (int r = 0; r < n; ++r) slab_set_planar(g[r], all);
    }
    for (int r = 0; r < n; ++r)
    {
        MH_HIP_TRY(hipSetDevice(g[r]->device));
        if (int rc = slab_exchange(g[r], g[r]->field[0], g[r]->main, true)) return slab_fail(g[r], rc);
    }
    for (int r = 0; r < n; ++r)
        if (int rc = slab_reset_chains(g[r])) return slab_fail(g[r], rc);
    int skew = 0;
    if (const char* v = getenv("MH_SLAB_GROUP_SKEW")) skew = atoi(v);
    for (int r = 1; r < n; ++r) g[r]->skew_pending = skew != 0;
    return MH_OK;
}

int mh_slab_is_planar(const mh_slab* s) { return s && s->planar_now ? 1 : 0; }

int mh_slab_plan_make(int nrows_global, int world, int rank, int periodic, int self_exchange, int rk_order, int fused_cut, mh_slab_plan* plan)
{
    const int rc = slab_plan_make(nrows_global, world, rank, periodic, self_exchange, rk_order, fused_cut, plan);
    if (rc != MH_OK) set_error("mh_slab_plan_make: %d rows, rank %d of %d, rk_order %d", nrows_global, rank, world, rk_order);
    return rc;
}
// the plan a slab was created with (its ghost rows and exchanges per step say whether it takes the one-launch step across its cuts)
int mh_slab_plan_of(const mh_slab* s, mh_slab_plan* plan) { if (! s || ! plan) return MH_E_INVALID; *plan = s->plan; return MH_OK; }

// bulk launches per time step: 1 where the RK2 step is one fused launch (lone slabs and, from 384 rows per slab on, slabs across their cuts)
int mh_slab_launches_per_step(const mh_slab* s) { return ! s ? 0 : ((s->fused || s->fused_cut) ? 1 : s->rk_order); }

int mh_slab_download(mh_slab* s, double* u_aos_slab_host)
{
    if (! s || ! u_aos_slab_host) return MH_E_INVALID;
    MH_HIP_TRY(hipSetDevice(s->device));
    const size_t ncell = (size_t) s->n0 * s->n1;
    MH_HIP_TRY(soa_to_aos_launch(s->field[0], s->staging, 5, s->n0, (size_t) s->n1, s->main));
    MH_HIP_TRY(hipMemcpyAsync(u_aos_slab_host, s->staging, ncell * 5 * sizeof(double), hipMemcpyDeviceToHost, s->main));
    MH_HIP_TRY(hipStreamSynchronize(s->main));
    return MH_OK;
}

int mh_slab_group_download(mh_slab** g, int n, double* u_aos_global_host)
{
    if (int rc = check_group(g, n)) return rc;
    if (! u_aos_global_host) return MH_E_INVALID;
    for (int r = 0; r < n; ++r)
        if (int rc = mh_slab_download(g[r], u_aos_global_host + (size_t) g[r]->row0 * g[r]->n1 * 5)) return rc;
    return MH_OK;
}

// does the nozzle row carry an azimuthal velocity? (STRICT: anything but the bit pattern of +0.0 counts, srhd_device.hpp)
static bool inflow_row_is_planar(const mh_slab* s, const double* row_aos)
{
    bool planar = true;
    for (size_t j = 0; j < (size_t) s->n1; ++j)
    {
        const double up = row_aos[5 * j + 3];
        planar = planar && up == 0.0 && ! (s->cloud.arith == MH_ARITH_STRICT && std::signbit(up));
    }
    return planar;
}

// the row itself, on the slab that owns the nozzle-side boundary (after the stages already queued: they read the previous row)
static int slab_store_inflow(mh_slab* s, const double* row_aos)
{
    if (s->row0 != 0) return MH_OK;
    const size_t nq = (size_t) s->n1;
    MH_HIP_TRY(hipSetDevice(s->device));
    std::vector<double> soa(5 * nq);
    for (size_t j = 0; j < nq; ++j) for (int q = 0; q < 5; ++q) soa[q * nq + j] = row_aos[5 * j + q];
    MH_HIP_TRY(hipStreamSynchronize(s->main));
    MH_HIP_TRY(hipStreamSynchronize(s->side));
    MH_HIP_TRY(hipMemcpy(s->inflow, soa.data(), soa.size() * sizeof(double), hipMemcpyHostToDevice));      // the staging vector is consumed before return
    return MH_OK;
}

static int slab_quiesce(mh_slab* s)
{
    MH_HIP_TRY(hipSetDevice(s->device));
    MH_HIP_TRY(hipStreamSynchronize(s->main));
    MH_HIP_TRY(hipStreamSynchronize(s->side));
    return MH_OK;
}

int mh_slab_set_inflow(mh_slab* s, const double* inflow_prims_aos_host)
{
    if (! s || s->kind != SLAB_CLOUD || ! inflow_prims_aos_host) { set_error("mh_slab_set_inflow: not a cloud slab"); return MH_E_STATE; }
    // planarity (mh_cloud_desc.planar): a row with an azimuthal velocity sends THIS slab to the general kernels until the next upload has looked
    // at the field again. A slab learns of the row only through this call, and a cloud slab starts out not knowing one (inflow_planar = false):
    //  * a lone slab / a slab whose neighbours live in other processes re-resolves here as at upload (the latter only under planar > 0);
    //  * members of a loopback group are resolved TOGETHER by mh_slab_group_set_inflow - through this per-slab call a member can only lose
    //    its planar kernels, never gain them, so a host that hands the row to one member only stays correct (on the general kernels).
    const bool row_planar = inflow_row_is_planar(s, inflow_prims_aos_host);
    if (! row_planar && s->planar_request > 0) { set_error("mh_slab_set_inflow: `planar` was asserted, but the nozzle row has an azimuthal velocity"); return slab_fail(s, MH_E_INVALID); }
    s->inflow_planar = row_planar;
    bool want = s->planar_now;
    if (! row_planar) want = false;
    else if (s->backend != EXCHANGE_LOOPBACK) want = s->planar_local && (! has_neighbours(s) || s->backend == EXCHANGE_NONE || s->planar_request > 0);
    if (want != s->planar_now)
    {
        if (int rc = slab_quiesce(s)) return slab_fail(s, rc);
        slab_set_planar(s, want);
    }
    if (! row_planar && s->backend == EXCHANGE_LOOPBACK)
    {
        // a member of a loopback group handed a rotating row on its own: its azimuthal momentum will cross the cuts, so EVERY member leaves the
        // planar kernels with it (radial slabs form an open chain: walk it both ways)
        for (int dir = 0; dir < 2; ++dir)
            for (mh_slab* p = dir ? s->peer_hi : s->peer_lo; p && p != s; p = dir ? p->peer_hi : p->peer_lo)
            {
                p->inflow_planar = false;
                if (p->planar_now) { if (int rc = slab_quiesce(p)) return slab_fail(s, rc); slab_set_planar(p, false); }
            }
    }
    if (int rc = slab_store_inflow(s, inflow_prims_aos_host)) return slab_fail(s, rc);
    return MH_OK;
}

int mh_slab_group_set_inflow(mh_slab** g, int n, const double* inflow_prims_aos_host)
{
    if (int rc = check_group(g, n)) return rc;
    if (g[0]->kind != SLAB_CLOUD || ! inflow_prims_aos_host) { set_error("mh_slab_group_set_inflow: not a group of cloud slabs"); return MH_E_STATE; }
    // every member learns of the row in ONE call: the group is planar only if every member's rows AND the row are (mh_slab_group_upload's rule)
    const bool row_planar = inflow_row_is_planar(g[0], inflow_prims_aos_host);
    if (! row_planar && g[0]->planar_request > 0) { set_error("mh_slab_group_set_inflow: `planar` was asserted, but the nozzle row has an azimuthal velocity"); return slab_fail(g[0], MH_E_INVALID); }
    bool all = row_planar, change = false;
    for (int r = 0; r < n; ++r) { g[r]->inflow_planar = row_planar; all = all && g[r]->planar_local; }
    for (int r = 0; r < n; ++r) change = change || g[r]->planar_now != all;
    if (change)
    {
        for (int r = 0; r < n; ++r) if (int rc = slab_quiesce(g[r])) return slab_fail(g[r], rc);
        for (int r = 0; r < n; ++r) slab_set_planar(g[r], all);
    }
    for (int r = 0; r < n; ++r) if (int rc = slab_store_inflow(g[r], inflow_prims_aos_host)) return slab_fail(g[r], rc);
    return MH_OK;
}

int mh_slab_step(mh_slab* s, double dt, int nsteps, int use_graph)
{
    if (! s) return MH_E_INVALID;
    if (s->backend == EXCHANGE_LOOPBACK) { set_error("mh_slab_step: member of a loopback group (use mh_slab_group_step)"); return slab_fail(s, MH_E_STATE); }
    MH_HIP_TRY(hipSetDevice(s->device));
    // RCCL point-to-point inside a stream capture crashes this stack (RCCL 2.26 / HIP 7.0: segfault in
    // hipStreamEndCapture, also through torch), so only the neighbour-less step is replayed from a graph; with
    // neighbours the step is issued eagerly from this loop (about ten HIP/RCCL calls per stage, no host language).
    if (s->fused)
    {
        // One launch per step, issued as it is: replaying it from a one-node graph costs 29 us MORE per step on this stack (4096^2, HIP 7.0:
        // 0.633 against 0.604 ms per step, alternating in one process, profiles/r03/fused_step_overhead.jsonl) - a graph pays where it
        // replaces several launches and their event traffic, and here there is one launch and none. MH_SLAB_FUSED_GRAPH=1 brings the replay
        // back for that measurement.
        const char* const env = getenv("MH_SLAB_FUSED_GRAPH");
        const bool replay = env && atoi(env) > 0;
        // profile: ONE pair of events around the call's launches (reported in the second-stage slot; the first stays empty). Events riding
        // on each launch (hipExtLaunchKernelGGL) read 3 % long for this kernel - longer than the steps of the timed region they follow.
        std::pair<hipEvent_t, hipEvent_t> pe = {nullptr, nullptr};
        if (s->profile && nsteps > 0)
        {
            MH_HIP_TRY(hipEventCreate(&pe.first));
            MH_HIP_TRY(hipEventCreate(&pe.second));
            MH_HIP_TRY(hipEventRecord(pe.first, s->main));
        }
        auto close_profile = [&] () -> int
        {
            if (! pe.first) return MH_OK;
            MH_HIP_TRY(hipEventRecord(pe.second, s->main));
            s->events[1].push_back(pe);
            s->event_launches[1].push_back(nsteps);
            return MH_OK;
        };
        for (int n = 0; n < nsteps; ++n)
        {
            if (! replay || ! use_graph) { if (int rc = slab_fused_step(s, dt)) return slab_fail(s, rc); continue; }
            // one captured step per direction (field A -> B, B -> A): the pointers are part of the graph
            const int k = s->field[0] == s->fused_src[0] ? 0 : 1;
            if (s->graph_dt != dt || s->fused_src[k] != s->field[0] || ! s->fused_exec[k])
            {
                if (s->graph_dt != dt)
                    for (int j = 0; j < 2; ++j) if (s->fused_exec[j]) { hipGraphExecDestroy(s->fused_exec[j]); s->fused_exec[j] = nullptr; }
                if (s->fused_exec[k]) { hipGraphExecDestroy(s->fused_exec[k]); s->fused_exec[k] = nullptr; }
                hipGraph_t graph = nullptr;
                double* const src = s->field[0];
                MH_HIP_TRY(hipStreamSynchronize(s->main));
                MH_HIP_TRY(hipStreamBeginCapture(s->main, hipStreamCaptureModeRelaxed));
                int rc = slab_fused_step(s, dt);
                hipError_t e = hipStreamEndCapture(s->main, &graph);
                if (rc) return slab_fail(s, rc);
                if (e != hipSuccess) return slab_fail(s, hip_fail(e, "hipStreamEndCapture"));
                std::swap(s->field[0], s->field[1]);          // the capture recorded the launch without running it
                e = hipGraphInstantiate(&s->fused_exec[k], graph, nullptr, nullptr, 0);
                hipGraphDestroy(graph);
                if (e != hipSuccess) { s->fused_exec[k] = nullptr; return slab_fail(s, hip_fail(e, "hipGraphInstantiate")); }
                s->fused_src[k] = src;
                s->graph_dt = dt;
            }
            MH_HIP_TRY(hipGraphLaunch(s->fused_exec[k], s->main));
            std::swap(s->field[0], s->field[1]);
            s->cur_out = s->field[0];
        }
        return close_profile();
    }
    if (use_graph && s->rk_order == 2 && ! s->profile && ! has_neighbours(s) && s->kind == SLAB_EULER)
    {
        if (! s->exec || s->graph_dt != dt)
        {
            if (s->exec) { hipGraphExecDestroy(s->exec); s->exec = nullptr; }
            hipGraph_t graph = nullptr;
            MH_HIP_TRY(hipStreamSynchronize(s->main));
            MH_HIP_TRY(hipStreamBeginCapture(s->main, hipStreamCaptureModeRelaxed));
            int rc = slab_one_step(s, dt);
            hipError_t e = hipStreamEndCapture(s->main, &graph);
            if (rc) return slab_fail(s, rc);
            if (e != hipSuccess) return slab_fail(s, hip_fail(e, "hipStreamEndCapture"));
            e = hipGraphInstantiate(&s->exec, graph, nullptr, nullptr, 0);
            hipGraphDestroy(graph);
            if (e != hipSuccess) { s->exec = nullptr; return slab_fail(s, hip_fail(e, "hipGraphInstantiate")); }
            s->graph_dt = dt;
        }
        for (int n = 0; n < nsteps; ++n) MH_HIP_TRY(hipGraphLaunch(s->exec, s->main));
        return MH_OK;
    }
    for (int n = 0; n < nsteps; ++n)
        if (int rc = slab_one_step(s, dt)) return slab_fail(s, rc);
    if (int rc = slab_join(s)) return slab_fail(s, rc);
    return MH_OK;
}

int mh_slab_group_step(mh_slab** g, int n, double dt, int nsteps)
{
    if (int rc = check_group(g, n)) return rc;
    for (int k = 0; k < nsteps; ++k)
        if (int rc = group_one_step(g, n, dt)) return slab_fail(g[0], rc);
    for (int r = 0; r < n; ++r) { MH_HIP_TRY(hipSetDevice(g[r]->device)); if (int rc = slab_join(g[r])) return slab_fail(g[r], rc); }
    return MH_OK;
}

int mh_slab_synchronize(mh_slab* s)
{
    if (! s) return MH_E_INVALID;
    MH_HIP_TRY(hipSetDevice(s->device));
    MH_HIP_TRY(hipStreamSynchronize(s->main));
    MH_HIP_TRY(hipStreamSynchronize(s->side));
    return MH_OK;
}

int mh_slab_status(mh_slab* s, mh_step_result* result)
{
    if (! s || ! result) return MH_E_INVALID;
    MH_HIP_TRY(hipSetDevice(s->device));
    uint32_t h[2] = {0, 0};
    MH_HIP_TRY(hipStreamSynchronize(s->side));
    MH_HIP_TRY(hipMemcpyAsync(h, s->status, sizeof h, hipMemcpyDeviceToHost, s->main));
    MH_HIP_TRY(hipMemsetAsync(s->status, 0, sizeof h, s->main));
    MH_HIP_TRY(hipStreamSynchronize(s->main));
    result->status = (int32_t) h[0];
    result->reserved = 0;
    // device word: 0xFFFFFFFF - local flat index (status_device.hpp); here: flat index in the GLOBAL host array
    result->first_bad_index = h[1] ? (uint64_t) (0xFFFFFFFFu - h[1]) + (uint64_t) s->row0 * (uint64_t) s->n1 : UINT64_MAX;
    return MH_OK;
}

int mh_slab_status_word(mh_slab* s, int32_t* status)
{
    if (! status) return MH_E_INVALID;
    mh_step_result r;
    if (int rc = mh_slab_status(s, &r)) return rc;
    *status = r.status;
    return MH_OK;
}

double* mh_slab_field_ptr(mh_slab* s, int which)
{
    return s && which >= 0 && which <= 1 ? s->field[which] : nullptr;
}

int mh_slab_profile_enable(mh_slab* s, int on)
{
    if (! s) return MH_E_INVALID;
    for (auto& v : s->events) { for (auto& ev : v) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); } v.clear(); }
    for (auto& v : s->event_launches) v.clear();
    s->profile = on != 0;
    return MH_OK;
}

int mh_slab_profile_read(mh_slab* s, double avg_ms[2], int nlaunches[2], int* bulk_rows)
{
    if (! s) return MH_E_INVALID;
    MH_HIP_TRY(hipSetDevice(s->device));
    MH_HIP_TRY(hipStreamSynchronize(s->main));
    for (int k = 0; k < 2; ++k)
    {
        double total = 0.0;
        for (auto& ev : s->events[k])
        {
            float ms = 0.f;
            MH_HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
            total += ms;
        }
        int count = 0;
        for (int c : s->event_launches[k]) count += c;
        if (avg_ms) avg_ms[k] = count == 0 ? 0.0 : total / count;
        if (nlaunches) nlaunches[k] = count;
    }
    // rows of a bulk launch; with staggered edges the average over the period (edges of 2 (k + 1) rows per side, k = 0 .. S-1)
    if (bulk_rows) *bulk_rows = s->n0 - 2 * (s->stagger ? s->edge_rows * (s->stagger + 1) / 2 : s->edge_rows);
    return MH_OK;
}

} // extern "C"

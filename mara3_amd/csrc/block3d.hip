// Block stepper: one rank's BLOCK of a 3-D uniform-cartesian Euler run under a 3-axis decomposition (BASELINE config 5:
// 1024^3 over 8 GPUs as (2,2,2) blocks), with the ghost exchange on all three axes.
//
// The cut is the reference's: blocks per axis from mara::propose_block_decomposition<3>(world) (src/app_parallel.hpp:119-131,
// mh_propose_block_decomposition), extents per axis from nd::divvy as create_access_pattern_array applies it (:148-179): block b
// of B on an axis of N cells owns [b N / B, (b + 1) N / B); the block at (c0, c1, c2) is rank (c0 B1 + c1) B2 + c2, the row-major
// position in the reference's array of access patterns. Upstream those blocks are views into one address space; here each is a
// device field with two stored ghost layers on every cut side, refreshed once per Runge-Kutta stage:
//   axis 0   the two edge planes are contiguous in the device layout: sent straight out of the field, received straight into
//            the ghost planes (as the slab stepper does);
//   axis 1   two rows of every interior plane: packed by a small kernel into a contiguous message [n0][5][2][n2], unpacked
//   axis 2   two columns of every row:        likewise, [n0][5][n1][2]                               into the ghost layers.
// Up to six messages per rank and stage, all in ONE RCCL group on the side stream (or, LOOPBACK backend: the blocks of a group
// are objects of one process and a receive is a stream-ordered device-to-device copy, see slab.hip). Edges and corners are not
// exchanged: the scheme is dimension-by-dimension (a cell reads i +- 2 along each axis separately).
//
// Overlap. A stage is two launches of the same kernel over lists of boxes (launch.hpp: Euler3dBox): the boundary SHELL - every
// cell within the kernel's tile granularity (8 planes / one tile of 4 or 8 rows / one 60-column strip) of a cut side - on the side stream,
// followed by pack -> exchange -> unpack, and the INTERIOR on the main stream. The interior reads no ghost cell, so the exchange
// has the whole interior launch to hide behind (512^3 per rank: ~21 MB per face, ~0.15 ms over xGMI, against ~6 ms of interior).
// Event protocol as the slab stepper's non-staggered one: main waits for the previous shell, side for the previous interior.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <cstring>
#include <string>
#include <vector>
#include "launch.hpp"
#include "rccl_api.hpp"

using namespace mh;

enum { BLOCK_EXCHANGE_NONE = 0, BLOCK_EXCHANGE_RCCL = 1, BLOCK_EXCHANGE_LOOPBACK = 2 };

struct mh_block
{
    int backend = BLOCK_EXCHANGE_NONE;
    int device = 0, rank = 0, world = 1, rk_order = 2;
    int B[3] = {1, 1, 1}, c[3] = {0, 0, 0};      // blocks per axis; this block's coordinates
    int start[3] = {0, 0, 0}, n[3] = {0, 0, 0};  // first global cell and extent per axis
    int nbr[3][2] = {{-1, -1}, {-1, -1}, {-1, -1}};      // neighbour rank on the low / high side of each axis (-1: physical boundary)
    mh_euler_cart_desc desc;                     // this block: n[] local, bc_lo0 / bc_hi0 per side
    Euler3dLayout lay;
    int ntiles1 = 0, nstrips = 0;
    Euler3dBox shell[6];
    int nshell = 0;
    Euler3dBox interior;
    size_t plane = 0, field_doubles = 0;
    size_t face_doubles[3] = {0, 0, 0};          // message size per axis
    double* field[2] = {nullptr, nullptr};
    double* staging = nullptr;
    int32_t* status = nullptr;
    double* sendbuf[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};       // axes 1, 2 only
    double* recvbuf[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    hipStream_t main = nullptr, side = nullptr;
    hipEvent_t ev_shell = nullptr, ev_interior = nullptr, ev_copied = nullptr, join = nullptr;
    ncclComm_t comm = nullptr;
    bool owns_comm = true;            // false: borrowed from an mh_comm (mh_block_use_comm)
    mh_block* peer[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    double* cur_out = nullptr;
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events[2];
    std::string error;
};

static int block_fail(mh_block* b, int code) { if (b) b->error = mh_last_error(nullptr); return code; }
static bool block_has_neighbours(const mh_block* b)
{
    for (int a = 0; a < 3; ++a) if (b->nbr[a][0] >= 0 || b->nbr[a][1] >= 0) return true;
    return false;
}

// where the two layers that go to / come from the neighbour on (axis, side) sit: edge layers (send) or ghost layers (receive)
static int layer_origin(const mh_block* b, int axis, int side, bool ghost)
{
    if (side == 0) return ghost ? -2 : 0;
    return ghost ? b->n[axis] : b->n[axis] - 2;
}

static double* axis0_block(const mh_block* b, double* f, int side, bool ghost)
{
    return f + (size_t) (layer_origin(b, 0, side, ghost) + 2) * 5 * b->plane;
}

// pack (to_buffers) the edge layers of axes 1 and 2 of field f into the send buffers / unpack the receive buffers into its ghost layers
static int block_faces(mh_block* b, double* f, bool pack, hipStream_t stream)
{
    for (int a = 1; a < 3; ++a)
        for (int s = 0; s < 2; ++s)
        {
            if (b->nbr[a][s] < 0) continue;
            const int o = layer_origin(b, a, s, ! pack);
            double* buf = pack ? b->sendbuf[a][s] : b->recvbuf[a][s];
            if (a == 1) MH_HIP_TRY(block_face_launch(pack, f, buf, b->n[0], o, 2, 0, b->n[2], b->n[1], b->n[2], b->lay.g1, b->lay.g2, stream));
            else        MH_HIP_TRY(block_face_launch(pack, f, buf, b->n[0], 0, b->n[1], o, 2, b->n[1], b->n[2], b->lay.g1, b->lay.g2, stream));
        }
    return MH_OK;
}

static int block_exchange_rccl(mh_block* b, double* f, hipStream_t stream)
{
    RcclApi* api = rccl();
    if (! api || ! b->comm) { set_error("mh_block: neighbours exist but the RCCL communicator was not connected (mh_block_connect)"); return MH_E_STATE; }
    MH_RCCL_TRY(api->GroupStart());
    // per axis: sends low side first, receives high side first - the order in which the two messages of a pair of ranks that are each
    // other's neighbour on BOTH sides (2 blocks on a periodic axis) match up
    for (int a = 0; a < 3; ++a)
        for (int s = 0; s < 2; ++s)
            if (b->nbr[a][s] >= 0)
                MH_RCCL_TRY(api->Send(a == 0 ? axis0_block(b, f, s, false) : b->sendbuf[a][s], b->face_doubles[a], ncclDouble, b->nbr[a][s], b->comm, stream));
    for (int a = 0; a < 3; ++a)
        for (int s = 1; s >= 0; --s)
            if (b->nbr[a][s] >= 0)
                MH_RCCL_TRY(api->Recv(a == 0 ? axis0_block(b, f, s, true) : b->recvbuf[a][s], b->face_doubles[a], ncclDouble, b->nbr[a][s], b->comm, stream));
    MH_RCCL_TRY(api->GroupEnd());
    return MH_OK;
}

// LOOPBACK "receive": copy what the neighbour object would have sent - its edge planes (axis 0, out of ITS current output field) or
// its send buffer of the facing side (axes 1, 2) - after its shell launch and packing (its ev_shell of this stage)
static int block_exchange_loopback(mh_block* b, double* f, hipStream_t stream, bool initial)
{
    for (int a = 0; a < 3; ++a)
        for (int s = 0; s < 2; ++s)
        {
            if (b->nbr[a][s] < 0) continue;
            mh_block* p = b->peer[a][s];
            if (! p) { set_error("mh_block: loopback peer missing"); return MH_E_STATE; }
            if (! initial) MH_HIP_TRY(hipStreamWaitEvent(stream, p->ev_shell, 0));
            const double* src = a == 0 ? axis0_block(p, initial ? p->field[0] : p->cur_out, 1 - s, false) : p->sendbuf[a][1 - s];
            double* dst = a == 0 ? axis0_block(b, f, s, true) : b->recvbuf[a][s];
            MH_HIP_TRY(hipMemcpyAsync(dst, src, b->face_doubles[a] * sizeof(double), hipMemcpyDeviceToDevice, stream));
        }
    MH_HIP_TRY(hipEventRecord(b->ev_copied, stream));
    return MH_OK;
}

static int block_exchange(mh_block* b, double* f, hipStream_t stream, bool initial)
{
    if (! block_has_neighbours(b)) return MH_OK;
    return b->backend == BLOCK_EXCHANGE_LOOPBACK ? block_exchange_loopback(b, f, stream, initial) : block_exchange_rccl(b, f, stream);
}

struct BlockStage { const double* in; const double* base; double* out; double dt, w; int which; };

static BlockStage block_stage_args(const mh_block* b, int i, double dt)
{
    BlockStage st;
    if (i == 0) { st.in = b->field[0]; st.base = nullptr; st.out = b->field[1]; st.w = 1.0; }
    else        { st.in = b->field[1]; st.base = b->field[0]; st.out = b->field[0]; st.w = 0.5; }
    st.dt = dt; st.which = i;
    return st;
}

// first half of a stage: the boundary shell and the packing of what the neighbours need (side stream)
static int block_stage_begin(mh_block* b, BlockStage& st)
{
    b->cur_out = st.out;
    if (! block_has_neighbours(b)) return MH_OK;
    // shell(k) reads interior(k-1)'s cells and overwrites cells of out(k) = in(k-1) that interior(k-1) read: wait for it.
    // interior(k) reads shell(k-1)'s cells and overwrites cells that shell(k-1) read: wait for that (issued before the re-record below).
    MH_HIP_TRY(hipStreamWaitEvent(b->main, b->ev_shell, 0));
    MH_HIP_TRY(hipStreamWaitEvent(b->side, b->ev_interior, 0));
    if (b->backend == BLOCK_EXCHANGE_LOOPBACK)
        for (int a = 0; a < 3; ++a) for (int s = 0; s < 2; ++s)       // the peers' copies out of this block's planes / send buffers are done
            if (b->peer[a][s]) MH_HIP_TRY(hipStreamWaitEvent(b->side, b->peer[a][s]->ev_copied, 0));
    MH_HIP_TRY(euler3d_stage_launch_boxes(&b->desc, b->lay, b->shell, b->nshell, st.in, st.base, st.out, st.dt, st.w, b->status, b->side));
    if (int rc = block_faces(b, st.out, true, b->side)) return rc;
    MH_HIP_TRY(hipEventRecord(b->ev_shell, b->side));
    return MH_OK;
}

// second half: exchange + unpacking (side stream), interior (main stream)
static int block_stage_finish(mh_block* b, BlockStage& st)
{
    const bool alone = ! block_has_neighbours(b);
    if (! alone)
    {
        if (int rc = block_exchange(b, st.out, b->side, false)) return rc;
        if (int rc = block_faces(b, st.out, false, b->side)) return rc;
    }
    std::pair<hipEvent_t, hipEvent_t> pe;
    if (b->profile) { hipEventCreate(&pe.first); hipEventCreate(&pe.second); hipEventRecord(pe.first, b->main); }
    const hipError_t e = euler3d_stage_launch_boxes(&b->desc, b->lay, &b->interior, 1, st.in, st.base, st.out, st.dt, st.w, b->status, b->main);
    if (b->profile) { hipEventRecord(pe.second, b->main); b->events[st.which].push_back(pe); }
    MH_HIP_TRY(e);
    if (! alone) MH_HIP_TRY(hipEventRecord(b->ev_interior, b->main));
    return MH_OK;
}

static int block_join(mh_block* b)
{
    if (! block_has_neighbours(b)) return MH_OK;
    MH_HIP_TRY(hipEventRecord(b->join, b->side));
    MH_HIP_TRY(hipStreamWaitEvent(b->main, b->join, 0));
    return MH_OK;
}

static int block_group_one_step(mh_block** g, int n, double dt)
{
    const int nstages = g[0]->rk_order;
    BlockStage st[64];
    for (int i = 0; i < nstages; ++i)
    {
        for (int r = 0; r < n; ++r) { MH_HIP_TRY(hipSetDevice(g[r]->device)); st[r] = block_stage_args(g[r], i, dt); if (int rc = block_stage_begin(g[r], st[r])) return rc; }
        for (int r = 0; r < n; ++r) { MH_HIP_TRY(hipSetDevice(g[r]->device)); if (int rc = block_stage_finish(g[r], st[r])) return rc; }
    }
    if (nstages == 1) for (int r = 0; r < n; ++r) std::swap(g[r]->field[0], g[r]->field[1]);
    return MH_OK;
}

static int check_block_group(mh_block** g, int n)
{
    if (! g || n < 1 || n > 64) { set_error("mh_block group: need 1..64 blocks"); return MH_E_INVALID; }
    for (int r = 0; r < n; ++r)
        if (! g[r] || g[r]->world != n || g[r]->rank != r || g[r]->rk_order != g[0]->rk_order || (block_has_neighbours(g[r]) && g[r]->backend != BLOCK_EXCHANGE_LOOPBACK))
        { set_error("mh_block group: block %d is not member %d of a loopback group of %d", r, r, n); return MH_E_INVALID; }
    return MH_OK;
}

// the rows of axis a that are cut sides' tile-granular boundary layers: [0, lo) and [count - hi, count) in units of `unit` cells
static void shell_range(int cells, int unit, bool cut_lo, bool cut_hi, int min_layers, int* lo, int* hi, int* count)
{
    const int units = (cells + unit - 1) / unit;
    *count = units;
    // the units that cover the first / last min_layers cells
    int l = cut_lo ? (min_layers + unit - 1) / unit : 0;
    int h = cut_hi ? units - (cells - min_layers) / unit : 0;
    if (l > units) l = units;
    if (h > units) h = units;
    if (l + h > units) { l = units; h = 0; }
    *lo = l; *hi = h;
}

static int block_create_common(mh_block** out, const mh_euler_cart_desc* global, int rk_order, int rank, int world, int device_id, int backend,
                               bool self_exchange = false)
{
    if (! out || ! global) return MH_E_INVALID;
    if (global->rank != 3) { set_error("mh_block: a 3-D grid is required"); return MH_E_INVALID; }
    if (rank < 0 || rank >= world) { set_error("mh_block: rank %d of %d", rank, world); return MH_E_INVALID; }
    if (rk_order != 1 && rk_order != 2) { set_error("rk_order must be 1 or 2"); return MH_E_INVALID; }
    const bool periodic0 = global->bc_lo0 == MH_BC_PERIODIC, periodic_t = global->bc_transverse == MH_BC_PERIODIC;
    if (periodic0 != (global->bc_hi0 == MH_BC_PERIODIC)) { set_error("periodic axis-0 bc must be set on both sides"); return MH_E_INVALID; }
    int B[3], c[3], start[3], count[3];
    if (int rc = mh_block_layout(global->n, world, rank, B, c, start, count)) return rc;
    MH_HIP_TRY(hipSetDevice(device_id));
    mh_block* b = new mh_block();
    b->device = device_id; b->rank = rank; b->world = world; b->rk_order = rk_order;
    for (int a = 0; a < 3; ++a) { b->B[a] = B[a]; b->c[a] = c[a]; b->start[a] = start[a]; b->n[a] = count[a]; }
    bool any = false;
    for (int a = 0; a < 3; ++a)
    {
        const bool periodic = a == 0 ? periodic0 : periodic_t;
        int cl[3] = {b->c[0], b->c[1], b->c[2]}, ch[3] = {b->c[0], b->c[1], b->c[2]};
        const bool wrap = periodic && (b->B[a] > 1 || self_exchange);          // self_exchange: a periodic axis of ONE block wraps through the exchange, to itself
        cl[a] = b->c[a] > 0 ? b->c[a] - 1 : (wrap ? b->B[a] - 1 : -1);
        ch[a] = b->c[a] < b->B[a] - 1 ? b->c[a] + 1 : (wrap ? 0 : -1);
        b->nbr[a][0] = cl[a] < 0 ? -1 : (cl[0] * b->B[1] + cl[1]) * b->B[2] + cl[2];
        b->nbr[a][1] = ch[a] < 0 ? -1 : (ch[0] * b->B[1] + ch[1]) * b->B[2] + ch[2];
        any = any || b->nbr[a][0] >= 0 || b->nbr[a][1] >= 0;
        if ((b->nbr[a][0] >= 0 || b->nbr[a][1] >= 0) && b->n[a] < 2)
        { const int thin = b->n[a]; delete b; set_error("mh_block: %d cells on axis %d are fewer than the two ghost layers a neighbour needs", thin, a); return MH_E_INVALID; }
        if (b->n[a] < 2) { delete b; set_error("mh_block: too many blocks for the global domain size"); return MH_E_INVALID; }
    }
    b->backend = any ? backend : BLOCK_EXCHANGE_NONE;
    b->desc = *global;
    for (int a = 0; a < 3; ++a) b->desc.n[a] = b->n[a];
    b->desc.bc_lo0 = b->nbr[0][0] >= 0 ? MH_BC_EXTERNAL : global->bc_lo0;
    b->desc.bc_hi0 = b->nbr[0][1] >= 0 ? MH_BC_EXTERNAL : global->bc_hi0;
    b->lay.bc_lo1 = b->nbr[1][0] >= 0 ? MH_BC_EXTERNAL : global->bc_transverse;
    b->lay.bc_hi1 = b->nbr[1][1] >= 0 ? MH_BC_EXTERNAL : global->bc_transverse;
    b->lay.bc_lo2 = b->nbr[2][0] >= 0 ? MH_BC_EXTERNAL : global->bc_transverse;
    b->lay.bc_hi2 = b->nbr[2][1] >= 0 ? MH_BC_EXTERNAL : global->bc_transverse;
    b->lay.g1 = (b->nbr[1][0] >= 0 || b->nbr[1][1] >= 0) ? 2 : 0;
    b->lay.g2 = (b->nbr[2][0] >= 0 || b->nbr[2][1] >= 0) ? 2 : 0;
    b->plane = (size_t) (b->n[1] + 2 * b->lay.g1) * (b->n[2] + 2 * b->lay.g2);
    b->field_doubles = (size_t) 5 * (b->n[0] + 4) * b->plane;
    b->face_doubles[0] = (size_t) 2 * 5 * b->plane;
    b->face_doubles[1] = (size_t) b->n[0] * 5 * 2 * b->n[2];
    b->face_doubles[2] = (size_t) b->n[0] * 5 * b->n[1] * 2;
    euler3d_tiling(&b->desc, &b->ntiles1, &b->nstrips);

    // boundary shell / interior in the kernel's work-item granularity: 8 planes (one short chunk), tiles of 4 (STRICT) or 8 (FAST) rows, 60-column strips
    int r_lo, r_hi, r_n, t_lo, t_hi, t_n, s_lo, s_hi, s_n;
    shell_range(b->n[0], 1, b->nbr[0][0] >= 0, b->nbr[0][1] >= 0, 8, &r_lo, &r_hi, &r_n);
    shell_range(b->n[1], euler3d_tile_rows(&b->desc), b->nbr[1][0] >= 0, b->nbr[1][1] >= 0, 2, &t_lo, &t_hi, &t_n);
    shell_range(b->n[2], 60, b->nbr[2][0] >= 0, b->nbr[2][1] >= 0, 2, &s_lo, &s_hi, &s_n);
    const Euler3dBox boxes[6] = {
        {0, r_lo, 0, t_n, 0, s_n}, {r_n - r_hi, r_n, 0, t_n, 0, s_n},
        {r_lo, r_n - r_hi, 0, t_lo, 0, s_n}, {r_lo, r_n - r_hi, t_n - t_hi, t_n, 0, s_n},
        {r_lo, r_n - r_hi, t_lo, t_n - t_hi, 0, s_lo}, {r_lo, r_n - r_hi, t_lo, t_n - t_hi, s_n - s_hi, s_n}};
    b->nshell = 0;
    for (const Euler3dBox& bx : boxes)
        if (bx.r1 > bx.r0 && bx.t1 > bx.t0 && bx.s1 > bx.s0) b->shell[b->nshell++] = bx;
    b->interior = {r_lo, r_n - r_hi, t_lo, t_n - t_hi, s_lo, s_n - s_hi};

    auto cleanup = [&] () { mh_block_destroy(b); };
    if (hipStreamCreateWithFlags(&b->main, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithPriority(&b->side, hipStreamNonBlocking, -1) != hipSuccess ||
        hipEventCreateWithFlags(&b->ev_shell, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b->ev_interior, hipEventDisableTiming | hipEventReleaseToDevice) != hipSuccess ||
        hipEventCreateWithFlags(&b->ev_copied, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b->join, hipEventDisableTiming) != hipSuccess)
    { cleanup(); set_error("mh_block: stream/event creation failed"); return MH_E_HIP; }
    for (auto& f : b->field)
    {
        if (hipMalloc((void**) &f, b->field_doubles * sizeof(double)) != hipSuccess) { cleanup(); set_error("mh_block: hipMalloc of %zu bytes failed", b->field_doubles * sizeof(double)); return MH_E_NOMEM; }
        hipMemsetAsync(f, 0, b->field_doubles * sizeof(double), b->main);
    }
    if (hipMalloc((void**) &b->status, 2 * sizeof(int32_t)) != hipSuccess) { cleanup(); return MH_E_NOMEM; }
    hipMemsetAsync(b->status, 0, 2 * sizeof(int32_t), b->main);
    if (hipMalloc((void**) &b->staging, (size_t) 5 * b->n[0] * b->n[1] * b->n[2] * sizeof(double)) != hipSuccess) { cleanup(); set_error("mh_block: hipMalloc(staging) failed"); return MH_E_NOMEM; }
    for (int a = 1; a < 3; ++a)
        for (int s = 0; s < 2; ++s)
            if (b->nbr[a][s] >= 0)
                if (hipMalloc((void**) &b->sendbuf[a][s], b->face_doubles[a] * sizeof(double)) != hipSuccess ||
                    hipMalloc((void**) &b->recvbuf[a][s], b->face_doubles[a] * sizeof(double)) != hipSuccess)
                { cleanup(); set_error("mh_block: hipMalloc(face buffers) failed"); return MH_E_NOMEM; }
    hipStreamSynchronize(b->main);
    *out = b;
    return MH_OK;
}

static int block_upload_cells(mh_block* b, const double* u_aos_block_host)
{
    MH_HIP_TRY(hipSetDevice(b->device));
    const size_t ncell = (size_t) b->n[0] * b->n[1] * b->n[2];
    MH_HIP_TRY(hipMemcpyAsync(b->staging, u_aos_block_host, ncell * 5 * sizeof(double), hipMemcpyHostToDevice, b->main));
    MH_HIP_TRY(block_transpose_launch(true, b->staging, b->field[0], b->n[0], b->n[1], b->n[2], b->lay.g1, b->lay.g2, b->main));
    MH_HIP_TRY(fill_ghost_rows_launch(b->field[0], 5, b->n[0], b->plane, b->desc.bc_lo0, b->desc.bc_hi0, b->main));
    if (int rc = block_faces(b, b->field[0], true, b->main)) return rc;
    return MH_OK;
}

static int block_finish_upload(mh_block* b)
{
    MH_HIP_TRY(hipSetDevice(b->device));
    if (int rc = block_exchange(b, b->field[0], b->main, true)) return rc;
    if (block_has_neighbours(b)) if (int rc = block_faces(b, b->field[0], false, b->main)) return rc;
    return MH_OK;
}

static int block_reset_chains(mh_block* b)
{
    MH_HIP_TRY(hipSetDevice(b->device));
    MH_HIP_TRY(hipStreamSynchronize(b->main));
    MH_HIP_TRY(hipStreamSynchronize(b->side));
    MH_HIP_TRY(hipEventRecord(b->ev_interior, b->main));
    MH_HIP_TRY(hipEventRecord(b->ev_shell, b->side));
    return MH_OK;
}

extern "C" {

int mh_block_connect(mh_block* b, const void* comm_id128)
{
    if (! b) return MH_E_INVALID;
    if (! block_has_neighbours(b) || b->comm) return MH_OK;
    if (b->backend != BLOCK_EXCHANGE_RCCL) { set_error("mh_block_connect: not an RCCL block"); return block_fail(b, MH_E_STATE); }
    RcclApi* api = rccl();
    if (! api) { set_error("librccl.so.1 could not be loaded"); return block_fail(b, MH_E_STATE); }
    if (! comm_id128) { set_error("mh_block: neighbours exist but no RCCL unique id was given"); return block_fail(b, MH_E_INVALID); }
    MH_HIP_TRY(hipSetDevice(b->device));
    ncclUniqueId id;
    std::memcpy(&id, comm_id128, sizeof id);
    ncclResult_t r = api->CommInitRank(&b->comm, b->world, id, b->rank);
    if (r != ncclSuccess) return block_fail(b, rccl_fail(r, "ncclCommInitRank"));
    return MH_OK;
}

int mh_block_use_comm(mh_block* b, mh_comm* c)
{
    if (! b || ! c) return MH_E_INVALID;
    if (! block_has_neighbours(b)) return MH_OK;
    if (b->backend != BLOCK_EXCHANGE_RCCL) { set_error("mh_block_use_comm: not an RCCL block"); return block_fail(b, MH_E_STATE); }
    if (b->comm) { set_error("mh_block_use_comm: the block has a communicator already"); return block_fail(b, MH_E_STATE); }
    if (c->world != b->world || c->rank != b->rank || c->device != b->device)
    {
        set_error("mh_block_use_comm: communicator is rank %d of %d on device %d, the block rank %d of %d on device %d", c->rank, c->world, c->device, b->rank, b->world, b->device);
        return block_fail(b, MH_E_INVALID);
    }
    b->comm = c->comm;
    b->owns_comm = false;
    return MH_OK;
}

int mh_block_create(mh_block** out, const mh_euler_cart_desc* global, int rk_order, int rank, int world, const void* comm_id128, int self_exchange,
                    int device_id)
{
    mh_block* b = nullptr;
    if (int rc = block_create_common(&b, global, rk_order, rank, world, device_id, BLOCK_EXCHANGE_RCCL, world == 1 && self_exchange != 0)) return rc;
    if (comm_id128) if (int rc = mh_block_connect(b, comm_id128)) { mh_block_destroy(b); return rc; }
    *out = b;
    return MH_OK;
}

int mh_block_group_create(mh_block** blocks, const mh_euler_cart_desc* global, int rk_order, int world, int device_id)
{
    if (! blocks || world < 1 || world > 64) { set_error("mh_block group: need 1..64 blocks"); return MH_E_INVALID; }
    for (int r = 0; r < world; ++r) blocks[r] = nullptr;
    for (int r = 0; r < world; ++r)
        if (int rc = block_create_common(&blocks[r], global, rk_order, r, world, device_id, BLOCK_EXCHANGE_LOOPBACK))
        {
            for (int q = 0; q < r; ++q) { mh_block_destroy(blocks[q]); blocks[q] = nullptr; }
            return rc;
        }
    for (int r = 0; r < world; ++r)
        for (int a = 0; a < 3; ++a) for (int s = 0; s < 2; ++s)
            blocks[r]->peer[a][s] = blocks[r]->nbr[a][s] >= 0 ? blocks[blocks[r]->nbr[a][s]] : nullptr;
    return MH_OK;
}

void mh_block_destroy(mh_block* b)
{
    if (! b) return;
    hipSetDevice(b->device);
    if (b->main) hipStreamSynchronize(b->main);
    if (b->side) hipStreamSynchronize(b->side);
    if (b->comm && b->owns_comm && rccl()) rccl()->CommDestroy(b->comm);
    for (auto& v : b->events) for (auto& ev : v) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
    for (auto& f : b->field) if (f) hipFree(f);
    for (int a = 0; a < 3; ++a) for (int s = 0; s < 2; ++s) { if (b->sendbuf[a][s]) hipFree(b->sendbuf[a][s]); if (b->recvbuf[a][s]) hipFree(b->recvbuf[a][s]); }
    if (b->staging) hipFree(b->staging);
    if (b->status) hipFree(b->status);
    for (hipEvent_t e : {b->ev_shell, b->ev_interior, b->ev_copied, b->join}) if (e) hipEventDestroy(e);
    if (b->main) hipStreamDestroy(b->main);
    if (b->side) hipStreamDestroy(b->side);
    delete b;
}

int mh_block_extent(const mh_block* b, int blocks_per_axis[3], int coords[3], int start[3], int count[3])
{
    if (! b) return MH_E_INVALID;
    for (int a = 0; a < 3; ++a)
    {
        if (blocks_per_axis) blocks_per_axis[a] = b->B[a];
        if (coords) coords[a] = b->c[a];
        if (start) start[a] = b->start[a];
        if (count) count[a] = b->n[a];
    }
    return MH_OK;
}

int mh_block_neighbours(const mh_block* b, int ranks[6], size_t message_doubles[3])
{
    if (! b) return MH_E_INVALID;
    for (int a = 0; a < 3; ++a)
    {
        if (ranks) { ranks[2 * a] = b->nbr[a][0]; ranks[2 * a + 1] = b->nbr[a][1]; }
        if (message_doubles) message_doubles[a] = b->face_doubles[a];
    }
    return MH_OK;
}

int mh_block_upload(mh_block* b, const double* u_aos_block_host)
{
    if (! b || ! u_aos_block_host) return MH_E_INVALID;
    if (b->backend == BLOCK_EXCHANGE_LOOPBACK) { set_error("mh_block_upload: member of a loopback group (use mh_block_group_upload)"); return block_fail(b, MH_E_STATE); }
    if (int rc = block_upload_cells(b, u_aos_block_host)) return block_fail(b, rc);
    if (int rc = block_finish_upload(b)) return block_fail(b, rc);
    if (int rc = block_reset_chains(b)) return block_fail(b, rc);
    return MH_OK;
}

int mh_block_download(mh_block* b, double* u_aos_block_host)
{
    if (! b || ! u_aos_block_host) return MH_E_INVALID;
    MH_HIP_TRY(hipSetDevice(b->device));
    const size_t ncell = (size_t) b->n[0] * b->n[1] * b->n[2];
    MH_HIP_TRY(block_transpose_launch(false, b->field[0], b->staging, b->n[0], b->n[1], b->n[2], b->lay.g1, b->lay.g2, b->main));
    MH_HIP_TRY(hipMemcpyAsync(u_aos_block_host, b->staging, ncell * 5 * sizeof(double), hipMemcpyDeviceToHost, b->main));
    MH_HIP_TRY(hipStreamSynchronize(b->main));
    return MH_OK;
}

// the block's cells out of / into the global host array [N0][N1][N2][5] (rows of n2 cells are contiguous there)
static void host_gather(const mh_block* b, const double* global, const int N[3], double* block)
{
    for (int i = 0; i < b->n[0]; ++i)
        for (int j = 0; j < b->n[1]; ++j)
            std::memcpy(block + ((size_t) i * b->n[1] + j) * b->n[2] * 5,
                        global + (((size_t) (b->start[0] + i) * N[1] + (b->start[1] + j)) * N[2] + b->start[2]) * 5, (size_t) b->n[2] * 5 * sizeof(double));
}
static void host_scatter(const mh_block* b, double* global, const int N[3], const double* block)
{
    for (int i = 0; i < b->n[0]; ++i)
        for (int j = 0; j < b->n[1]; ++j)
            std::memcpy(global + (((size_t) (b->start[0] + i) * N[1] + (b->start[1] + j)) * N[2] + b->start[2]) * 5,
                        block + ((size_t) i * b->n[1] + j) * b->n[2] * 5, (size_t) b->n[2] * 5 * sizeof(double));
}
static void global_shape(mh_block** g, int n, int N[3])
{
    for (int a = 0; a < 3; ++a) N[a] = 0;
    for (int r = 0; r < n; ++r) for (int a = 0; a < 3; ++a) if (g[r]->start[a] + g[r]->n[a] > N[a]) N[a] = g[r]->start[a] + g[r]->n[a];
}

int mh_block_group_upload(mh_block** g, int n, const double* u_aos_global_host)
{
    if (int rc = check_block_group(g, n)) return rc;
    if (! u_aos_global_host) return MH_E_INVALID;
    int N[3];
    global_shape(g, n, N);
    std::vector<double> tmp;
    for (int r = 0; r < n; ++r)
    {
        tmp.resize((size_t) 5 * g[r]->n[0] * g[r]->n[1] * g[r]->n[2]);
        host_gather(g[r], u_aos_global_host, N, tmp.data());
        if (int rc = block_upload_cells(g[r], tmp.data())) return block_fail(g[r], rc);
        MH_HIP_TRY(hipStreamSynchronize(g[r]->main));          // tmp is reused
    }
    for (int r = 0; r < n; ++r) if (int rc = block_finish_upload(g[r])) return block_fail(g[r], rc);
    for (int r = 0; r < n; ++r) if (int rc = block_reset_chains(g[r])) return block_fail(g[r], rc);
    return MH_OK;
}

int mh_block_group_download(mh_block** g, int n, double* u_aos_global_host)
{
    if (int rc = check_block_group(g, n)) return rc;
    if (! u_aos_global_host) return MH_E_INVALID;
    int N[3];
    global_shape(g, n, N);
    std::vector<double> tmp;
    for (int r = 0; r < n; ++r)
    {
        tmp.resize((size_t) 5 * g[r]->n[0] * g[r]->n[1] * g[r]->n[2]);
        if (int rc = mh_block_download(g[r], tmp.data())) return rc;
        host_scatter(g[r], u_aos_global_host, N, tmp.data());
    }
    return MH_OK;
}

int mh_block_step(mh_block* b, double dt, int nsteps)
{
    if (! b) return MH_E_INVALID;
    if (b->backend == BLOCK_EXCHANGE_LOOPBACK) { set_error("mh_block_step: member of a loopback group (use mh_block_group_step)"); return block_fail(b, MH_E_STATE); }
    MH_HIP_TRY(hipSetDevice(b->device));
    for (int k = 0; k < nsteps; ++k)
        if (int rc = block_group_one_step(&b, 1, dt)) return block_fail(b, rc);
    if (int rc = block_join(b)) return block_fail(b, rc);
    return MH_OK;
}

int mh_block_group_step(mh_block** g, int n, double dt, int nsteps)
{
    if (int rc = check_block_group(g, n)) return rc;
    for (int k = 0; k < nsteps; ++k)
        if (int rc = block_group_one_step(g, n, dt)) return block_fail(g[0], rc);
    for (int r = 0; r < n; ++r) { MH_HIP_TRY(hipSetDevice(g[r]->device)); if (int rc = block_join(g[r])) return block_fail(g[r], rc); }
    return MH_OK;
}

int mh_block_synchronize(mh_block* b)
{
    if (! b) return MH_E_INVALID;
    MH_HIP_TRY(hipSetDevice(b->device));
    MH_HIP_TRY(hipStreamSynchronize(b->main));
    MH_HIP_TRY(hipStreamSynchronize(b->side));
    return MH_OK;
}

int mh_block_status(mh_block* b, mh_step_result* result, const int global_n[3])
{
    if (! b || ! result) return MH_E_INVALID;
    MH_HIP_TRY(hipSetDevice(b->device));
    uint32_t h[2] = {0, 0};
    MH_HIP_TRY(hipStreamSynchronize(b->side));
    MH_HIP_TRY(hipMemcpyAsync(h, b->status, sizeof h, hipMemcpyDeviceToHost, b->main));
    MH_HIP_TRY(hipMemsetAsync(b->status, 0, sizeof h, b->main));
    MH_HIP_TRY(hipStreamSynchronize(b->main));
    result->status = (int32_t) h[0];
    result->reserved = 0;
    result->first_bad_index = UINT64_MAX;
    if (h[1])
    {
        // device word: 0xFFFFFFFF - flat index within the block; with the global shape: flat index in the global host array
        uint64_t local = 0xFFFFFFFFu - h[1];
        if (global_n)
        {
            const uint64_t k = local % b->n[2], j = (local / b->n[2]) % b->n[1], i = local / ((uint64_t) b->n[1] * b->n[2]);
            local = ((i + b->start[0]) * global_n[1] + (j + b->start[1])) * (uint64_t) global_n[2] + (k + b->start[2]);
        }
        result->first_bad_index = local;
    }
    return MH_OK;
}

int mh_block_profile(mh_block* b, int enable, double avg_ms[2], int nlaunches[2], long* interior_cells)
{
    if (! b) return MH_E_INVALID;
    MH_HIP_TRY(hipSetDevice(b->device));
    MH_HIP_TRY(hipStreamSynchronize(b->main));
    for (int k = 0; k < 2; ++k)
    {
        double total = 0.0;
        for (auto& ev : b->events[k])
        {
            float ms = 0.f;
            MH_HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
            total += ms;
        }
        if (avg_ms) avg_ms[k] = b->events[k].empty() ? 0.0 : total / b->events[k].size();
        if (nlaunches) nlaunches[k] = (int) b->events[k].size();
        for (auto& ev : b->events[k]) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
        b->events[k].clear();
    }
    if (interior_cells)
    {
        const Euler3dBox& x = b->interior;
        const long j0 = (long) x.t0 * 8, j1 = std::min((long) x.t1 * 8, (long) b->n[1]), k0 = (long) x.s0 * 60, k1 = std::min((long) x.s1 * 60, (long) b->n[2]);
        *interior_cells = (x.r1 > x.r0 && j1 > j0 && k1 > k0) ? (long) (x.r1 - x.r0) * (j1 - j0) * (k1 - k0) : 0;
    }
    b->profile = enable != 0;
    return MH_OK;
}

} // extern "C"

// C ABI of the `binary` sub-program path (include/mara_hip.h, "binary" section): the stateless stage launchers and the
// solver object that replaces binary::next_solution (src/subprog_binary.cpp:258-293) around them.
//
// One time step = [maximum wavespeed reduction ->] stage 1 -> stage 2 (fused with the RK combine) -> ONE host
// synchronisation that brings back 2 x 18 totals, the status word and (when the binary is not live, i.e. always before
// begin_live_binary) the maximum wavespeed of the NEW state for the next step's dt. The reference synchronises
// implicitly after every array expression; here the field never leaves the device and the host only sees ~300 bytes
// per step. The step is transactional: stage outputs go to alternate buffers and the solution pointer is swapped
// only after the status word came back clean, because the reference's safe-mode retry restarts from the OLD solution.
//
// Multi-GPU (uniform-depth trees): the mesh is cut into BANDS of whole rows of tree blocks, rank r of N owning block rows
// partition_shape(n / block_size, N)[r] (the reference distributes whole blocks over its thread pool, tree.map(fn, pool)
// src/core_tree.hpp:615-625; blocks stay whole because work_done_on is nonlinear in each block's sink sums, scheme.cpp:356-365).
// Per stage: the stage launch over the band, one two-row ghost exchange with the (periodic) neighbours, and per host
// synchronisation one sum of the 2 x 18 totals (+ max of the wavespeed, + the status words) over the ranks, after which every
// rank does the same scalar bookkeeping on the same numbers. Backends as the slab stepper: RCCL (one process per GPU: send/recv
// and three small all-reduces) or LOOPBACK (the bands as objects of one process sharing one stream: copies and a host sum).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "launch.hpp"
#include "rccl_api.hpp"
#include "binary_host.hpp"
#include "binary_device.hpp"

namespace mh {

size_t binary_scratch_doubles(const mh_binary_desc* d, const BinaryBand* band);
hipError_t binary_stage_launch(const mh_binary_desc* d, const double* xv, const double* yv, const double* u_in, const double* u_base,
                               double* u_out, const double* u_init, const double* br, const double bodies[10], double dt, double weight,
                               double theta, double* totals, double* scratch, int32_t* status, hipStream_t stream, const BinaryBand* band,
                               const BinaryTotalsOverlap* overlap = nullptr, int32_t* status_clear = nullptr, const BinaryRows* rows = nullptr);
int binary_edge_rows(int n0);
hipError_t binary_maxw_launch(const mh_binary_desc* d, const double* xv, const double* yv, const double* u, const double bodies[10],
                              double* result, hipStream_t stream, const BinaryBand* band);

// graded trees (binary_tree.hip)
struct TreeGeom { const int32_t* topo; const int32_t* level; const double* edges; int nb, bs; };
struct TreeBuffers { double *prim, *gx, *gy, *fx, *fy, *block_out, *block_vals, *tile_maxw; };
enum { TREE_PRIM_GRAD = 1, TREE_FLUX = 2, TREE_UPDATE = 4, TREE_TOTALS = 8, TREE_ALL = 15 };
struct TreeRun { int phases, b0, b1; const int32_t* order; const int32_t* ids; };
hipError_t binary_tree_stage_launch(const mh_binary_desc* d, const TreeGeom& g, const TreeBuffers& w, const double* u_in, const double* u_base,
                                    double* u_out, const double* u_init, const double* br, const double bodies[10], double dt, double weight,
                                    double theta, double* totals, int32_t* status, hipStream_t stream, const double* bodies_next, double* maxw_result,
                                    const TreeRun* part = nullptr);
hipError_t binary_tree_min_dt_launch(const mh_binary_desc* d, const TreeGeom& g, const double* u, const double bodies[10], double* result, hipStream_t stream);

// diagnostics (binary_diag.hip)
hipError_t binary_diag_sums_launch(const double* u, const double* xv, const double* yv, const double* edges, int n, int nb, int bs, bool tree,
                                   bool qform, double* partial, double* out, hipStream_t stream);
hipError_t binary_diag_fields_launch(const double* u, const double* xv, const double* yv, const double* edges, int n, int nb, int bs, bool tree,
                                     bool qform, double* fields, hipStream_t stream);

static int check_binary_desc(const mh_binary_desc* d)
{
    if (! d) { set_error("binary: null descriptor"); return MH_E_INVALID; }
    if (d->n < 8 || d->block_size < 1 || d->n % d->block_size != 0) { set_error("binary: n = %d must be a multiple of block_size = %d (and >= 8)", d->n, d->block_size); return MH_E_INVALID; }
    int depth = 0;
    while ((d->block_size << depth) < d->n) ++depth;
    if ((d->block_size << depth) != d->n) { set_error("binary: n / block_size = %d is not a power of two (uniform-depth tree)", d->n / d->block_size); return MH_E_INVALID; }
    if (! (d->mach_number > 0.0) || ! (d->sink_radius > 0.0) || ! (d->domain_radius > 0.0)) { set_error("binary: mach_number, sink_radius and domain_radius must be positive"); return MH_E_INVALID; }
    return MH_OK;
}

struct HostMirror        // pinned: what one step brings back; dev_small on the device has the same layout
{
    double  totals[2][MH_BINARY_NTOTALS];
    double  maxw;
    int32_t status[2];
};

} // namespace mh

using namespace mh;

struct mh_binary
{
    int device = 0;
    hipStream_t stream = nullptr;
    mh_binary_desc desc;
    mh_binary_run run;
    double h = 0.0;
    size_t field_doubles = 0;
    double* u[4] = {nullptr, nullptr, nullptr, nullptr};     // [0] solution, [1] first-stage result, [2] step result, [3] (uniform mesh, one domain) the eager stage's
    double* u_init = nullptr;
    double* br = nullptr;
    double* xv = nullptr;
    double* yv = nullptr;
    double* scratch = nullptr;
    double* dev_small = nullptr;                     // totals[2][18], maxw
    int32_t* status = nullptr;
    double* staging = nullptr;
    HostMirror* mirror = nullptr;
    mh_binary_state state;
    double last_dt = 0.0;
    // maximum wavespeed of the current solution, computed ahead by the previous step
    bool   maxw_ready = false;
    double maxw_value = 0.0;
    mh_binary_state maxw_for;
    bool profile = false;
    // profile: ONE pair of events around the stage launches of each mh_binary_next call - from in front of its first stage to behind its
    // last one (the end-of-call fetch excluded) - and the number of stages between them. (Events around every stage put two more markers
    // between consecutive kernels: 5 - 8 us each on this stack, on a 100 us stage.)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<int> event_stages;
    hipEvent_t prof_end = nullptr;
    int prof_stages = 0;
    // graded tree (mh_binary_tree_create): block-major fields [nb][3][bs][bs], neighbour table, per-stage work arrays
    bool tree = false;
    TreeGeom geom = {nullptr, nullptr, nullptr, 0, 0};
    TreeBuffers work = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int32_t* topo_dev = nullptr;
    int32_t* level_dev = nullptr;
    double* edges_dev = nullptr;
    std::vector<double> host_staging;
    // DISTRIBUTED graded tree (mh_binary_tree_band_create / _group_create): every member holds the whole tree, stored in the order of the
    // Hilbert curve through its leaves (binary_host.cpp: binary_tree_curve_order), and runs the block kernels on its own run of that curve,
    // blocks [tb0, tb1); the members' results are gathered after each kernel (tree_stage_distributed). Totals and time-step bound are then
    // formed by every member over all blocks, in the caller's block order: the same bits on every member as on one domain.
    bool tdist = false;
    int tb0 = 0, tb1 = 0;
    std::vector<int32_t> perm;                       // perm[k]: the caller's number of the block stored k-th
    std::vector<int> tcut;                           // world + 1 offsets of the members' runs
    int32_t* order_dev = nullptr;                    // where the caller's k-th block is stored
    int32_t* ids_dev = nullptr;                      // = perm, on the device
    // band decomposition (see the header comment): rows [row0, row0 + n0) of the mesh; world == 1: the whole mesh
    int rank = 0, world = 1, row0 = 0, n0 = 0;
    bool banded = false;                             // ghost rows come from an exchange (world > 1, or the RCCL path to self at world 1)
    int backend = 0;                                 // 0 none, 1 RCCL, 2 loopback
    mh_binary* peer_lo = nullptr;
    mh_binary* peer_hi = nullptr;
    ncclComm_t comm = nullptr;
    bool owns_stream = true;                         // loopback members run on the first member's stream
    // the totals of a stage (sink sums, reduction) run on a second stream beside and behind the stage kernel (binary.hip: BinaryTotalsOverlap)
    hipStream_t side = nullptr;
    hipEvent_t ev_input[2] = {nullptr, nullptr}, ev_stage[2] = {nullptr, nullptr}, ev_totals = nullptr;
    size_t scratch_doubles = 0;                      // per RK stage and parity: each stage has its own partial sums
    bool totals_pending = false;
    // The FIRST STAGE OF THE NEXT STEP, issued before this step's totals have come back (binary_attempt): it writes its totals and status
    // into the OTHER small block and partial-sum buffers (`parity`), and this step's fetch runs on the second stream beside it.
    double* small[2] = {nullptr, nullptr};           // dev_small = small[parity]
    int parity = 0;
    bool fetch_on_side = false;                      // the pending fetch is queued on the second stream already: it does not wait for what the main stream got since
    bool eager_valid = false;                        // the first stage of the step from (eager_for, eager_dt) is in flight in the other parity
    mh_binary_state eager_for;
    double eager_dt = 0.0;
    // Edge rows first (bands): the stage kernel runs the first and last `edge` rows of the band in one small launch, their exchange travels
    // on `xstream` (RCCL) beside the interior launch, and the main stream waits for it behind the interior (team_exchange_begin / _end).
    int edge = 0;                                    // 0: one launch per stage, the exchange behind it on the main stream
    hipStream_t xstream = nullptr;
    hipEvent_t ev_edge = nullptr, ev_edone = nullptr, ev_xchg = nullptr;
    bool owns_comm = true;                           // false: borrowed from an mh_comm (mh_binary_band_use_comm)
    double* reduced_dev = nullptr;                   // RCCL: the small block summed over the ranks (out of place: the local one stays local)
    uint32_t* gather_dev = nullptr;                  // RCCL: every rank's two status words {bits, 0xFFFFFFFF - first failing whole-mesh index}
    uint32_t* gather_host = nullptr;                 // pinned mirror of it
    mh_step_result last_failure = {0, 0, UINT64_MAX};            // of the most recent failed attempt (mh_binary_last_failure)
};

enum { BAND_NONE = 0, BAND_RCCL = 1, BAND_LOOPBACK = 2 };
static BinaryBand band_of(const mh_binary* b) { return BinaryBand{b->n0, b->row0, b->banded ? 1 : 0}; }

static double* totals_dev(mh_binary* b, int stage) { return b->dev_small + stage * MH_BINARY_NTOTALS; }
static double* maxw_dev(mh_binary* b) { return b->dev_small + 2 * MH_BINARY_NTOTALS; }
static int32_t* status_of(double* small_block) { return reinterpret_cast<int32_t*>(small_block + 2 * MH_BINARY_NTOTALS + 1); }
static void use_parity(mh_binary* b, int parity)
{
    b->parity = parity;
    b->dev_small = b->small[parity];
    b->status = status_of(b->dev_small);
}

static int binary_bodies(const mh_full_orbital_elements& E, double t, mh_two_body_t* B)
{
    return mh_two_body_state(&E, t, B);
}

// Bnext (graded trees only): the stage also leaves the time-step bound of the state it writes, evaluated with these bodies, in maxw_dev
static int launch_stage(mh_binary* b, const double* u_in, const double* u_base, double* u_out, const mh_two_body_t& B, double dt,
                        double weight, double theta, int slot, const mh_two_body_t* Bnext = nullptr, int parity = -1,
                        hipEvent_t input_event = nullptr, int32_t* status_clear = nullptr, int part = BIN_ROWS_ALL, hipStream_t on = nullptr,
                        hipEvent_t edges_done = nullptr)
{
    // on: the stream of this launch (the edge rows of an RCCL band run on its exchange stream), default the main stream
    hipStream_t stream = on ? on : b->stream;
    // input_event: an event of the main stream behind which u_in is complete (a stage's ev_stage), or null: one is recorded here
    if (parity < 0) parity = b->parity;
    double* const small_block = b->small[parity] ? b->small[parity] : b->dev_small;
    if (b->profile && part != BIN_ROWS_EDGES) ++b->prof_stages;
    if (b->tree)
        MH_HIP_TRY(binary_tree_stage_launch(&b->desc, b->geom, b->work, u_in, u_base, u_out, b->u_init, b->br, B.body1, dt, weight, theta,
                                            totals_dev(b, slot), b->status, b->stream, Bnext ? Bnext->body1 : nullptr, maxw_dev(b)));
    else
    {
        const BinaryBand band = band_of(b);
        BinaryTotalsOverlap ov = {b->side, input_event ? input_event : b->ev_input[slot], b->ev_stage[slot], nullptr};
        if (b->side && ! input_event && part != BIN_ROWS_EDGES) MH_HIP_TRY(hipEventRecord(b->ev_input[slot], b->stream));
        const BinaryRows rows = {part, b->edge, edges_done};
        MH_HIP_TRY(binary_stage_launch(&b->desc, b->xv, b->yv, u_in, u_base, u_out, b->u_init, b->br, B.body1, dt, weight, theta,
                                       small_block + slot * MH_BINARY_NTOTALS, b->scratch + (size_t) (2 * parity + slot) * b->scratch_doubles,
                                       status_of(small_block), stream, &band, b->side ? &ov : nullptr, status_clear, part == BIN_ROWS_ALL ? nullptr : &rows));
        if (part != BIN_ROWS_EDGES) b->totals_pending = b->side != nullptr;
    }
    return MH_OK;
}

static bool same_point(const mh_binary_state& a, const mh_binary_state& c)
{
    return a.time == c.time && memcmp(&a.orbital_elements, &c.orbital_elements, sizeof(mh_full_orbital_elements)) == 0;
}

// A TEAM is what advances together: one solver (world 1, or one RCCL rank of several processes), or all the band objects of a
// loopback group. Every member holds the same scalar state; phase by phase the members are driven in lockstep.
struct Team { mh_binary** m; int n; };
static int check_binary_group(mh_binary** g, int n);

// two ghost rows per side of field u[k] of every member, from its periodic neighbours (after the launches that wrote the rows they send).
// A member that runs its edge rows first (edge > 0, RCCL) sends from its second stream, behind the edge launch only; team_exchange_end
// makes the main stream wait for it. Loopback members share one stream: the copies are queued where they stand.
static int team_exchange(const Team& t, int k, bool behind_edge_launch = false)
{
    for (int r = 0; r < t.n; ++r)
    {
        mh_binary* b = t.m[r];
        if (! b->banded) continue;
        const size_t n = (size_t) b->desc.n, blk = 2 * 3 * n;          // two rows, three variables: contiguous
        double* f = b->u[k];
        if (b->backend == BAND_LOOPBACK)
        {
            // members share one stream: every launch of the team that writes the rows sent is already queued in front of these copies
            const double* lo = b->peer_lo->u[k] + (size_t) b->peer_lo->n0 * 3 * n;          // the low neighbour's last two rows
            const double* hi = b->peer_hi->u[k] + blk;                                        // the high neighbour's first two rows
            MH_HIP_TRY(hipMemcpyAsync(f, lo, blk * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
            MH_HIP_TRY(hipMemcpyAsync(f + (size_t) (b->n0 + 2) * 3 * n, hi, blk * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
        }
        else
        {
            RcclApi* api = rccl();
            if (! api || ! b->comm) { set_error("binary bands: RCCL communicator missing"); return MH_E_STATE; }
            // within a stage: behind the edge launch, which ran on the exchange stream (team_exchange_end joins it); else on the main stream
            hipStream_t xs = behind_edge_launch && b->edge > 0 && b->xstream ? b->xstream : b->stream;
            const int lo = (b->rank + b->world - 1) % b->world, hi = (b->rank + 1) % b->world;
            MH_RCCL_TRY(api->GroupStart());
            MH_RCCL_TRY(api->Send(f + blk, blk, ncclDouble, lo, b->comm, xs));                                   // rows 0, 1
            MH_RCCL_TRY(api->Send(f + (size_t) b->n0 * 3 * n, blk, ncclDouble, hi, b->comm, xs));                // rows n0 - 2, n0 - 1
            MH_RCCL_TRY(api->Recv(f + (size_t) (b->n0 + 2) * 3 * n, blk, ncclDouble, hi, b->comm, xs));          // order as slab.hip (lo == hi at world 2)
            MH_RCCL_TRY(api->Recv(f, blk, ncclDouble, lo, b->comm, xs));
            MH_RCCL_TRY(api->GroupEnd());
            if (xs != b->stream) MH_HIP_TRY(hipEventRecord(b->ev_xchg, xs));
        }
    }
    return MH_OK;
}
static int team_exchange_end(const Team& t)
{
    for (int r = 0; r < t.n; ++r)
    {
        mh_binary* b = t.m[r];
        if (b->banded && b->backend == BAND_RCCL && b->edge > 0 && b->xstream) MH_HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_xchg, 0));
    }
    return MH_OK;
}

// the merged status words of a FAILED attempt as the boundary's error pair; kept on every member until the next mh_binary_next call
static void team_note_status(const Team& t)
{
    const HostMirror& m = *t.m[0]->mirror;
    if (! m.status[0]) return;
    const mh_step_result res = {m.status[0], 0, (uint64_t) (0xFFFFFFFFu - (uint32_t) m.status[1])};
    for (int r = 0; r < t.n; ++r) t.m[r]->last_failure = res;
}

// bring the small block (2 x 18 totals, maximum wavespeed, status words) to the host of every member - summed / maximised over the bands
static int team_fetch(const Team& t)
{
    for (int r = 0; r < t.n; ++r)
    {
        mh_binary* b = t.m[r];
        if (b->fetch_on_side)
        {
            // the main stream already runs the next step's first stage: this step's block travelled on the second stream (binary_attempt)
            MH_HIP_TRY(hipEventSynchronize(b->ev_totals));        // the copy queued by binary_attempt ahead of the eager stage
            b->fetch_on_side = false;
            b->totals_pending = false;
            team_note_status(t);
            return MH_OK;                 // (only ever set for a team of one)
        }
        if (b->totals_pending)          // the totals of the stages issued since the last fetch: their stream joins the main one here
        {
            MH_HIP_TRY(hipEventRecord(b->ev_totals, b->side));
            MH_HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_totals, 0));
            b->totals_pending = false;
        }
        const double* src = b->dev_small;
        if ((b->banded || b->tdist) && b->backend == BAND_RCCL)
        {
            RcclApi* api = rccl();
            if (! api || ! b->comm) { set_error("binary bands: RCCL communicator missing"); return MH_E_STATE; }
            const size_t nt = 2 * MH_BINARY_NTOTALS;
            if (! b->tdist)          // (a distributed tree's totals and time-step bound are formed by every member over all blocks: nothing to reduce)
            {
            MH_RCCL_TRY(api->AllReduce(b->dev_small, b->reduced_dev, nt, ncclDouble, ncclSum, b->comm, b->stream));
            MH_RCCL_TRY(api->AllReduce(b->dev_small + nt, b->reduced_dev + nt, 1, ncclUint64, ncclMax, b->comm, b->stream));     // wavespeeds are > 0: bit order = value order
            src = b->reduced_dev;
            }
            // status words: the bits are an OR, which no RCCL reduction forms (a max of {NEG_DENSITY} and {NAN} would drop the former), so
            // every rank's pair is gathered and merged on the host below, exactly as the loopback members' are
            MH_RCCL_TRY(api->AllGather(b->dev_small + nt + 1, b->gather_dev, 2, ncclUint32, b->comm, b->stream));
            MH_HIP_TRY(hipMemcpyAsync(b->gather_host, b->gather_dev, (size_t) b->world * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
        }
        MH_HIP_TRY(hipMemcpyAsync(b->mirror, src, sizeof(HostMirror), hipMemcpyDeviceToHost, b->stream));
    }
    for (int r = 0; r < t.n; ++r) MH_HIP_TRY(hipStreamSynchronize(t.m[r]->stream));
    for (int r = 0; r < t.n; ++r)
    {
        mh_binary* b = t.m[r];
        if (! ((b->banded || b->tdist) && b->backend == BAND_RCCL)) continue;
        uint32_t bits = 0, key = 0;               // the kernels' keys are whole-mesh already: (row0 + r) n + col (binary.hip)
        for (int q = 0; q < b->world; ++q)
        {
            bits |= b->gather_host[2 * q];
            if (b->gather_host[2 * q + 1] > key) key = b->gather_host[2 * q + 1];
        }
        b->mirror->status[0] = (int32_t) bits;
        b->mirror->status[1] = (int32_t) key;
    }
    if (t.n > 1)
    {
        // loopback: the members' blocks are summed here, in band order
        HostMirror sum = *t.m[0]->mirror;
        for (int r = 1; r < t.n; ++r)
        {
            const HostMirror& o = *t.m[r]->mirror;
            if (! t.m[0]->tdist)          // (distributed tree: every member holds the totals of the whole tree already)
            {
                for (int s2 = 0; s2 < 2; ++s2) for (int k = 0; k < MH_BINARY_NTOTALS; ++k) sum.totals[s2][k] = sum.totals[s2][k] + o.totals[s2][k];
                if (o.maxw > sum.maxw) sum.maxw = o.maxw;
            }
            sum.status[0] |= o.status[0];
            if ((uint32_t) o.status[1] > (uint32_t) sum.status[1]) sum.status[1] = o.status[1];
        }
        for (int r = 0; r < t.n; ++r) *t.m[r]->mirror = sum;
    }
    team_note_status(t);
    return MH_OK;
}

static int team_maxw(const Team& t, int k, const mh_two_body_t& B)
{
    for (int r = 0; r < t.n; ++r)
    {
        mh_binary* b = t.m[r];
        if (b->tree) MH_HIP_TRY(binary_tree_min_dt_launch(&b->desc, b->geom, b->u[k], B.body1, maxw_dev(b), b->stream));
        else { const BinaryBand band = band_of(b); MH_HIP_TRY(binary_maxw_launch(&b->desc, b->xv, b->yv, b->u[k], B.body1, maxw_dev(b), b->stream, &band)); }
    }
    return MH_OK;
}

// The members' runs of one block-major array, gathered onto every member: `doubles` per block; ptr(m) = the array of member m.
template<class Ptr>
static int tree_gather(const Team& t, size_t doubles, Ptr ptr)
{
    mh_binary* b0 = t.m[0];
    if (b0->backend == BAND_LOOPBACK)
    {
        // members share one stream: the launches that wrote the runs are queued in front of these copies
        for (int r = 0; r < t.n; ++r)
        {
            const size_t off = (size_t) t.m[r]->tb0 * doubles, cnt = (size_t) (t.m[r]->tb1 - t.m[r]->tb0) * doubles;
            if (cnt == 0) continue;
            for (int q = 0; q < t.n; ++q)
                if (q != r) MH_HIP_TRY(hipMemcpyAsync(ptr(t.m[q]) + off, ptr(t.m[r]) + off, cnt * sizeof(double), hipMemcpyDeviceToDevice, b0->stream));
        }
        return MH_OK;
    }
    if (b0->backend != BAND_RCCL || b0->world < 2) return MH_OK;          // (one rank: its run is the whole tree)
    RcclApi* api = rccl();
    if (! api || ! b0->comm) { set_error("binary tree: RCCL communicator missing"); return MH_E_STATE; }
    double* a = ptr(b0);
    MH_RCCL_TRY(api->GroupStart());
    for (int q = 0; q < b0->world; ++q)
    {
        if (q == b0->rank) continue;
        const size_t mine = (size_t) (b0->tb1 - b0->tb0) * doubles, theirs = (size_t) (b0->tcut[q + 1] - b0->tcut[q]) * doubles;
        if (mine) MH_RCCL_TRY(api->Send(a + (size_t) b0->tb0 * doubles, mine, ncclDouble, q, b0->comm, b0->stream));
        if (theirs) MH_RCCL_TRY(api->Recv(a + (size_t) b0->tcut[q] * doubles, theirs, ncclDouble, q, b0->comm, b0->stream));
    }
    MH_RCCL_TRY(api->GroupEnd());
    return MH_OK;
}

// One stage of a DISTRIBUTED graded tree: each of the three block kernels on the member's own run of blocks, the members' results
// gathered behind each (primitives and slopes; fluxes; the new field with its per-tile sums and wavespeeds), then the totals - and, behind
// a step's last stage, the time-step bound - formed by every member over ALL blocks in the caller's block order.
static int tree_stage_distributed(const Team& t, int in, int base, int outk, const mh_two_body_t& B, double dt, double weight, double theta, int slot,
                                  const mh_two_body_t* Bnext)
{
    if (t.m[0]->profile) ++t.m[0]->prof_stages;
    const int bs = t.m[0]->geom.bs;
    const size_t cell = (size_t) 3 * bs * bs, face = (size_t) 3 * (bs + 1) * bs, tiles = (size_t) (bs * bs + 255) / 256;
    auto run = [&] (int phases) -> int
    {
        for (int r = 0; r < t.n; ++r)
        {
            mh_binary* m = t.m[r];
            const bool own = (phases & TREE_TOTALS) == 0;
            const TreeRun part = {phases, own ? m->tb0 : 0, own ? m->tb1 : m->geom.nb, m->order_dev, m->ids_dev};
            MH_HIP_TRY(binary_tree_stage_launch(&m->desc, m->geom, m->work, m->u[in], base < 0 ? nullptr : m->u[base], m->u[outk], m->u_init, m->br, B.body1,
                                                dt, weight, theta, totals_dev(m, slot), m->status, m->stream, Bnext ? Bnext->body1 : nullptr, maxw_dev(m), &part));
        }
        return MH_OK;
    };
    if (int rc = run(TREE_PRIM_GRAD)) return rc;
    if (int rc = tree_gather(t, cell, [] (mh_binary* m) { return m->work.prim; })) return rc;
    if (int rc = tree_gather(t, cell, [] (mh_binary* m) { return m->work.gx; })) return rc;
    if (int rc = tree_gather(t, cell, [] (mh_binary* m) { return m->work.gy; })) return rc;
    if (int rc = run(TREE_FLUX)) return rc;
    if (int rc = tree_gather(t, face, [] (mh_binary* m) { return m->work.fx; })) return rc;
    if (int rc = tree_gather(t, face, [] (mh_binary* m) { return m->work.fy; })) return rc;
    if (int rc = run(TREE_UPDATE)) return rc;
    if (int rc = tree_gather(t, cell, [outk] (mh_binary* m) { return m->u[outk]; })) return rc;
    if (int rc = tree_gather(t, tiles * 16, [] (mh_binary* m) { return m->work.block_out; })) return rc;
    if (Bnext) if (int rc = tree_gather(t, tiles, [] (mh_binary* m) { return m->work.tile_maxw; })) return rc;
    return run(TREE_TOTALS);
}

// one attempt at a full step from (u[0], state); on success the new solution is in u[2] of every member and *out
static int binary_attempt(const Team& t, double dt, bool safe_mode, bool prefetch_maxw, mh_binary_state* out, bool* failed)
{
    static_assert(sizeof(mh_two_body_t) == 10 * sizeof(double), "bodies are passed as double[10]");
    mh_binary* b = t.m[0];                     // scalars: identical on every member
    const mh_binary_state S0 = b->state;
    const double theta = safe_mode ? 0.0 : b->desc.plm_theta;
    const bool naf = b->run.no_accretion_force != 0;
    *failed = false;
    mh_two_body_t B1, B2;
    if (int rc = binary_bodies(S0.orbital_elements, S0.time, &B1)) return rc;
    // Eager first stage (uniform mesh, one domain, fixed time step, binary not live, RK2): while the binary is not live the first stage of
    // the NEXT step needs nothing of this step's totals - its bodies follow from the elements and the time - so the previous attempt may
    // have issued it already, into the other parity's small block and partial sums; this step's fetch then ran beside it instead of
    // leaving the GPU idle for a host round trip per step (~50 of 260 us at 2048^2). It is used only if it was issued for exactly this
    // state and time step; a failed step discards it (the safe-mode retry starts from u[0], which the eager stage never writes).
    const bool eager_ok = t.n == 1 && ! b->banded && ! b->tree && b->side && b->small[1] && b->run.fixed_dt && b->run.rk_order == 2 && ! safe_mode
                          && ! (S0.time > b->run.begin_live_binary);
    const bool eager_used = eager_ok && b->eager_valid && same_point(b->eager_for, S0) && b->eager_dt == dt;
    b->eager_valid = false;
    if (eager_used)
    {
        use_parity(b, b->parity ^ 1);          // its totals and status words are in the other block,
        double* tmp = b->u[1]; b->u[1] = b->u[3]; b->u[3] = tmp;          // its result in the fourth field
    }
    else for (int r = 0; r < t.n; ++r) MH_HIP_TRY(hipMemsetAsync(t.m[r]->status, 0, 2 * sizeof(int32_t), t.m[r]->stream));

    // clear: the second stage also zeroes the status words of the OTHER small block, for the eager stage issued right behind it
    auto stage = [&t] (int in, int base, int outk, const mh_two_body_t& B, double dt_, double w, double th, int slot, const mh_two_body_t* Bnext = nullptr,
                       hipEvent_t input_event = nullptr, int32_t* clear = nullptr) -> int
    {
        if (t.m[0]->tdist) return tree_stage_distributed(t, in, base, outk, B, dt_, w, th, slot, Bnext);
        // Members with neighbours: the edge rows - RCCL: on the exchange stream, which first waits for what the main stream holds so far -
        // their exchange behind them, and the interior on the main stream beside both; the main stream (and the reduction of the totals)
        // wait for the exchange behind the interior. Loopback members share one stream: edge launches, copies, interior launches in a row.
        // A member too thin to split, or told not to, runs whole before the exchange.
        auto own_xstream = [] (const mh_binary* m) { return m->edge > 0 && m->backend == BAND_RCCL && m->xstream != nullptr; };
        for (int r = 0; r < t.n; ++r)
        {
            mh_binary* m = t.m[r];
            if (own_xstream(m))
            {
                MH_HIP_TRY(hipEventRecord(m->ev_edge, m->stream));
                MH_HIP_TRY(hipStreamWaitEvent(m->xstream, m->ev_edge, 0));
            }
            if (int rc = launch_stage(m, m->u[in], base < 0 ? nullptr : m->u[base], m->u[outk], B, dt_, w, th, slot, Bnext, -1, input_event, clear,
                                      m->edge > 0 ? BIN_ROWS_EDGES : BIN_ROWS_ALL, own_xstream(m) ? m->xstream : nullptr)) return rc;
        }
        // RCCL: the interior launch is issued BEFORE the send / recv group - enqueueing the group costs the host tens of microseconds, during
        // which the GPU would otherwise hold the edge rows only (kernel trace, profiles/r03/binary_band_edges.md)
        const bool exchange_last = own_xstream(t.m[0]);
        if (! exchange_last) if (int rc = team_exchange(t, outk, true)) return rc;
        for (int r = 0; r < t.n; ++r)
        {
            mh_binary* m = t.m[r];
            if (m->edge > 0)
            {
                if (own_xstream(m)) MH_HIP_TRY(hipEventRecord(m->ev_edone, m->xstream));          // the edge waves' partial sums are complete behind it
                if (int rc = launch_stage(m, m->u[in], base < 0 ? nullptr : m->u[base], m->u[outk], B, dt_, w, th, slot, Bnext, -1, input_event, clear,
                                          BIN_ROWS_INTERIOR, nullptr, own_xstream(m) ? m->ev_edone : nullptr)) return rc;
            }
        }
        if (exchange_last) if (int rc = team_exchange(t, outk, true)) return rc;
        return team_exchange_end(t);
    };

    // profile: behind the last stage launch of the call (team_next)
    auto mark_profile_end = [&] () -> int
    {
        if (b->profile && b->prof_end && ! prefetch_maxw) MH_HIP_TRY(hipEventRecord(b->prof_end, b->stream));
        return MH_OK;
    };
    if (b->run.rk_order == 1)
    {
        if (int rc = stage(0, -1, 2, B1, dt, 1.0, theta, 0)) return rc;
        if (int rc = mark_profile_end()) return rc;
        if (int rc = team_fetch(t)) return rc;
        if (b->mirror->status[0]) { *failed = true; return MH_OK; }
        if (binary_apply_totals(S0, B1, b->mirror->totals[0], dt, naf, b->run.begin_live_binary, out) != MH_OK) { *failed = true; return MH_OK; }
        return MH_OK;
    }

    if (! eager_used) if (int rc = stage(0, -1, 1, B1, dt, 1.0, theta, 0)) return rc;
    mh_binary_state S1;
    const bool live = S0.time > b->run.begin_live_binary;
    if (live)
    {
        // the elements the second stage is evaluated with depend on the first stage's totals
        if (int rc = team_fetch(t)) return rc;
        if (b->mirror->status[0]) { *failed = true; return MH_OK; }
        if (binary_apply_totals(S0, B1, b->mirror->totals[0], dt, naf, b->run.begin_live_binary, &S1) != MH_OK) { *failed = true; return MH_OK; }
        if (int rc = binary_bodies(S1.orbital_elements, S1.time, &B2)) return rc;
    }
    else
    {
        if (int rc = binary_bodies(S0.orbital_elements, S0.time + dt, &B2)) return rc;   // elements + (...) * 0 = elements
    }
    // look ahead: the next step's maximum wavespeed, evaluated on the step result while the totals travel - on a graded tree by
    // the second stage's own last workgroup, on the uniform mesh by a launch behind it
    mh_binary_state ahead = S0;
    bool launched_ahead = false;
    mh_two_body_t Bn;
    if (prefetch_maxw && ! live && ! b->run.fixed_dt)
    {
        ahead.time = S0.time * 0.5 + ((S0.time + dt) + dt) * 0.5;
        launched_ahead = binary_bodies(ahead.orbital_elements, ahead.time, &Bn) == MH_OK;
    }
    // the eager stage of the next step (see above): decided before the second stage is issued, which prepares its status words
    mh_two_body_t Be;
    mh_binary_state nxt = S0;
    bool eager_next = false;
    if (eager_ok && prefetch_maxw)          // (prefetch_maxw: another step follows in this call)
    {
        nxt.time = S0.time * 0.5 + ((S0.time + dt) + dt) * 0.5;          // binary_combine_scalars' expression; the elements do not change while not live
        eager_next = ! (nxt.time > b->run.begin_live_binary) && binary_bodies(nxt.orbital_elements, nxt.time, &Be) == MH_OK;
    }
    const int other = b->parity ^ 1;
    // (the other block was fetched - the host waited for the copy - before this attempt began: nothing reads or writes it now)
    if (int rc = stage(1, 0, 2, B2, dt, 0.5, theta, 1, launched_ahead && b->tree ? &Bn : nullptr, t.n == 1 && b->side && ! b->tree && ! b->banded ? b->ev_stage[0] : nullptr,
                       eager_next ? status_of(b->small[other]) : nullptr)) return rc;
    if (launched_ahead && ! b->tree)
        if (int rc = team_maxw(t, 2, Bn)) return rc;
    if (int rc = mark_profile_end()) return rc;
    if (eager_next)
    {
        // This step's fetch goes onto the second stream NOW, ahead of the eager stage's own sink sums and reduction. Stream order there puts
        // it behind this step's last reduction, which itself waited for the second stage (ev_stage[1]): totals and status words are complete.
        MH_HIP_TRY(hipMemcpyAsync(b->mirror, b->dev_small, sizeof(HostMirror), hipMemcpyDeviceToHost, b->side));
        MH_HIP_TRY(hipEventRecord(b->ev_totals, b->side));
        b->fetch_on_side = true;
        // The eager stage follows the second stage on the main stream with nothing in between: it writes the FOURTH field (the second
        // stage's sink sums may still be reading u[1]; u[3] was last read a whole step ago, before a fetch the host has waited for).
        if (int rc = launch_stage(b, b->u[2], nullptr, b->u[3], Be, dt, 1.0, b->desc.plm_theta, 0, nullptr, other, b->ev_stage[1])) return rc;
        b->eager_valid = true;
        b->eager_for = nxt;
        b->eager_dt = dt;
    }
    if (int rc = team_fetch(t)) return rc;
    if (b->mirror->status[0]) { *failed = true; return MH_OK; }
    if (! live && binary_apply_totals(S0, B1, b->mirror->totals[0], dt, naf, b->run.begin_live_binary, &S1) != MH_OK) { *failed = true; return MH_OK; }
    mh_two_body_t B2s;
    if (int rc = binary_bodies(S1.orbital_elements, S1.time, &B2s)) return rc;
    if (memcmp(&B2s, &B2, sizeof(B2)) != 0)
    {
        // cannot happen unless an orbital-element perturbation was non-finite; the stage ran with other bodies than the
        // bookkeeping assumes: treat like a failed stage
        *failed = true;
        return MH_OK;
    }
    mh_binary_state S2;
    if (binary_apply_totals(S1, B2, b->mirror->totals[1], dt, naf, b->run.begin_live_binary, &S2) != MH_OK) { *failed = true; return MH_OK; }
    binary_combine_scalars(S0, S2, out);
    if (launched_ahead)
    {
        for (int r = 0; r < t.n; ++r)
        {
            t.m[r]->maxw_ready = true;
            t.m[r]->maxw_value = b->mirror->maxw;
            t.m[r]->maxw_for = ahead;
        }
    }
    return MH_OK;
}

// nsteps x next_solution for a team (subprog_binary.cpp:258-293)
static int team_next(const Team& t, int nsteps, int* safe_mode_steps)
{
    mh_binary* b = t.m[0];
    if (safe_mode_steps) *safe_mode_steps = 0;
    for (int r = 0; r < t.n; ++r) { t.m[r]->last_failure = {0, 0, UINT64_MAX}; t.m[r]->eager_valid = false; }
    std::pair<hipEvent_t, hipEvent_t> span = {nullptr, nullptr};
    if (b->profile && nsteps > 0)
    {
        MH_HIP_TRY(hipEventCreate(&span.first));
        MH_HIP_TRY(hipEventCreate(&span.second));
        MH_HIP_TRY(hipEventRecord(span.first, b->stream));
        b->prof_end = span.second;
        b->prof_stages = 0;
    }
    auto close_span = [&] ()
    {
        if (! span.first) return;
        b->prof_end = nullptr;
        if (b->prof_stages > 0) { b->events.push_back(span); b->event_stages.push_back(b->prof_stages); }
        else { (void) hipEventDestroy(span.first); (void) hipEventDestroy(span.second); }
        span = {nullptr, nullptr};
    };
    struct SpanGuard { decltype(close_span)& f; ~SpanGuard() { f(); } } span_guard{close_span};
    for (int s = 0; s < nsteps; ++s)
    {
        // dt: subprog_binary.cpp:281-283
        double dt = b->run.recommended_time_step;
        if (! b->run.fixed_dt)
        {
            double maxw;
            if (b->maxw_ready && same_point(b->maxw_for, b->state)) maxw = b->maxw_value;
            else
            {
                mh_two_body_t B;
                if (int rc = binary_bodies(b->state.orbital_elements, b->state.time, &B)) return rc;
                if (int rc = team_maxw(t, 0, B)) return rc;
                if (int rc = team_fetch(t)) return rc;
                maxw = b->mirror->maxw;
            }
            // uniform grid: the reduction returns the largest wavespeed; graded tree: already min over blocks of spacing / wavespeed
            dt = b->tree ? b->run.cfl_number * maxw : b->run.cfl_number * (b->h / maxw);
        }
        for (int r = 0; r < t.n; ++r) t.m[r]->maxw_ready = false;
        mh_binary_state next;
        bool failed = false;
        if (int rc = binary_attempt(t, dt, false, s + 1 < nsteps, &next, &failed)) return rc;
        if (failed)
        {
            for (int r = 0; r < t.n; ++r) t.m[r]->eager_valid = false;          // an eager first stage started from the rejected result
            if (safe_mode_steps) ++*safe_mode_steps;
            dt = dt * 0.1;
            for (int r = 0; r < t.n; ++r) t.m[r]->maxw_ready = false;
            if (int rc = binary_attempt(t, dt, true, false, &next, &failed)) return rc;
            if (failed) { set_error("negative density in updated state"); return MH_E_PHYSICS; }
        }
        for (int r = 0; r < t.n; ++r)
        {
            mh_binary* m = t.m[r];
            double* tmp = m->u[0]; m->u[0] = m->u[2]; m->u[2] = tmp;      // commit
            m->state = next;
            m->last_dt = dt;
        }
    }
    return MH_OK;
}

// host [nb][bs][bs][3] -> device layout [nb][3][bs][bs]
static void tree_to_device_layout(const mh_binary* b, const double* aos, double* out)
{
    const int bs = b->geom.bs;
    for (int k = 0; k < b->geom.nb; ++k)
    {
        const size_t from = b->perm.empty() ? (size_t) k : (size_t) b->perm[k];          // (a distributed tree stores the caller's blocks in curve order)
        for (int q = 0; q < 3; ++q)
            for (int c = 0; c < bs * bs; ++c)
                out[((size_t) k * 3 + q) * bs * bs + c] = aos[(from * bs * bs + c) * 3 + q];
    }
}

extern "C" {

size_t mh_binary_field_doubles(const mh_binary_desc* d)
{
    return d ? (size_t) 3 * (d->n + 4) * d->n : 0;
}

size_t mh_binary_scratch_doubles(const mh_binary_desc* d)
{
    if (check_binary_desc(d) != MH_OK) return 0;
    return binary_scratch_doubles(d, nullptr);
}

int mh_binary_stage(const mh_binary_desc* d, const double* xv, const double* yv, const double* u_in, const double* u_base, double* u_out,
                    const double* u_init, const double* br, const double* bodies, double dt, double w, double* totals, double* scratch,
                    int32_t* status, void* stream)
{
    if (int rc = check_binary_desc(d)) return rc;
    if (! xv || ! yv || ! u_in || ! u_out || ! u_init || ! br || ! bodies || ! totals || ! scratch || u_in == u_out) { set_error("binary stage: null or aliased argument"); return MH_E_INVALID; }
    if (w != 1.0 && ! u_base) { set_error("binary stage: combine needs u_base"); return MH_E_INVALID; }
    MH_HIP_TRY(binary_stage_launch(d, xv, yv, u_in, u_base, u_out, u_init, br, bodies, dt, w, d->plm_theta, totals, scratch, status, (hipStream_t) stream, nullptr));
    return MH_OK;
}

int mh_binary_max_wavespeed(const mh_binary_desc* d, const double* xv, const double* yv, const double* u, const double* bodies,
                            double* result, void* stream)
{
    if (int rc = check_binary_desc(d)) return rc;
    if (! xv || ! yv || ! u || ! bodies || ! result) { set_error("binary max_wavespeed: null argument"); return MH_E_INVALID; }
    MH_HIP_TRY(binary_maxw_launch(d, xv, yv, u, bodies, result, (hipStream_t) stream, nullptr));
    return MH_OK;
}

// u_init_aos / br: the WHOLE mesh ([n][n][3], [n][n]); a band keeps its rows. shared_stream: loopback members run on one stream.
static int binary_create_common(mh_binary** out, int device, const mh_binary_desc* d, const mh_binary_run* run, const double* xv, const double* yv,
                                const double* u_init_aos, const double* br, int rank, int world, int backend, hipStream_t shared_stream, bool self_exchange = false)
{
    if (! out || ! run || ! xv || ! yv || ! u_init_aos || ! br) { set_error("binary create: null argument"); return MH_E_INVALID; }
    if (int rc = check_binary_desc(d)) return rc;
    if (run->rk_order != 1 && run->rk_order != 2) { set_error("binary::next_solution: rk_order must be 1 or 2"); return MH_E_INVALID; }
    const int block_rows = d->n / d->block_size;
    if (rank < 0 || rank >= world || world > block_rows) { set_error("binary bands: rank %d of %d over %d rows of tree blocks", rank, world, block_rows); return MH_E_INVALID; }
    MH_HIP_TRY(hipSetDevice(device));
    mh_binary* b = new mh_binary();
    b->device = device;
    b->desc = *d;
    b->run = *run;
    b->rank = rank; b->world = world;
    b->banded = world > 1 || self_exchange;
    b->backend = b->banded ? backend : BAND_NONE;
    size_t ba, bb;
    mh_partition_rows((size_t) block_rows, (size_t) world, (size_t) rank, &ba, &bb);          // whole rows of tree blocks, nd::partition_shape's formula
    b->row0 = (int) ba * d->block_size;
    b->n0 = (int) (bb - ba) * d->block_size;
    const size_t n = d->n, n0 = (size_t) b->n0;
    int depth = 0;
    while ((d->block_size << depth) < d->n) ++depth;
    b->h = 2.0 * d->domain_radius / d->block_size / (1 << depth);
    b->field_doubles = (size_t) 3 * (n0 + 4) * n;
    const BinaryBand band = band_of(b);
    auto fail = [&] (hipError_t e, const char* what) { mh_binary_destroy(b); return hip_fail(e, what); };
#define B_TRY(call) do { hipError_t _e = (call); if (_e != hipSuccess) return fail(_e, #call); } while (0)
    if (shared_stream) { b->stream = shared_stream; b->owns_stream = false; }
    else B_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    for (int k = 0; k < (b->banded ? 3 : 4); ++k) B_TRY(hipMalloc(&b->u[k], b->field_doubles * sizeof(double)));
    B_TRY(hipMalloc(&b->u_init, b->field_doubles * sizeof(double)));
    B_TRY(hipMalloc(&b->br, n0 * n * sizeof(double)));
    B_TRY(hipMalloc(&b->xv, (n + 1) * sizeof(double)));
    B_TRY(hipMalloc(&b->yv, (n + 1) * sizeof(double)));
    b->scratch_doubles = binary_scratch_doubles(d, &band);
    B_TRY(hipMalloc(&b->scratch, 4 * b->scratch_doubles * sizeof(double)));          // [parity][stage]
    B_TRY(hipStreamCreateWithFlags(&b->side, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k)
    {
        B_TRY(hipEventCreateWithFlags(&b->ev_input[k], hipEventDisableTiming));
        B_TRY(hipEventCreateWithFlags(&b->ev_stage[k], hipEventDisableTiming));
    }
    B_TRY(hipEventCreateWithFlags(&b->ev_totals, hipEventDisableTiming));
    for (int k = 0; k < 2; ++k)
    {
        B_TRY(hipMalloc(&b->small[k], sizeof(HostMirror)));
        B_TRY(hipMemsetAsync(b->small[k], 0, sizeof(HostMirror), b->stream));
    }
    use_parity(b, 0);
    b->edge = 0;          // one launch per stage unless mh_binary_band_set_edge_rows says otherwise (measured: DESIGN.md 7.1)
    if (b->backend == BAND_RCCL)
    {
        B_TRY(hipStreamCreateWithFlags(&b->xstream, hipStreamNonBlocking));
        B_TRY(hipEventCreateWithFlags(&b->ev_edge, hipEventDisableTiming));
        B_TRY(hipEventCreateWithFlags(&b->ev_xchg, hipEventDisableTiming));
        B_TRY(hipEventCreateWithFlags(&b->ev_edone, hipEventDisableTiming));
        B_TRY(hipMalloc(&b->reduced_dev, sizeof(HostMirror)));
        B_TRY(hipMalloc(&b->gather_dev, (size_t) world * 2 * sizeof(uint32_t)));
        B_TRY(hipHostMalloc((void**) &b->gather_host, (size_t) world * 2 * sizeof(uint32_t), hipHostMallocDefault));
    }
    B_TRY(hipMalloc(&b->staging, n0 * n * 3 * sizeof(double)));
    B_TRY(hipHostMalloc((void**) &b->mirror, sizeof(HostMirror), hipHostMallocDefault));
    B_TRY(hipMemcpyAsync(b->xv, xv, (n + 1) * sizeof(double), hipMemcpyHostToDevice, b->stream));
    B_TRY(hipMemcpyAsync(b->yv, yv, (n + 1) * sizeof(double), hipMemcpyHostToDevice, b->stream));
    B_TRY(hipMemcpyAsync(b->br, br + (size_t) b->row0 * n, n0 * n * sizeof(double), hipMemcpyHostToDevice, b->stream));
    B_TRY(hipMemcpyAsync(b->staging, u_init_aos + (size_t) b->row0 * n * 3, n0 * n * 3 * sizeof(double), hipMemcpyHostToDevice, b->stream));
    B_TRY(aos_to_soa_launch(b->staging, b->u_init, 3, b->n0, n, b->stream));
    // ghost rows of the initial field are never read (the buffer term is cell-local); the whole mesh keeps its periodic images anyway
    if (! b->banded) B_TRY(fill_ghost_rows_launch(b->u_init, 3, d->n, n, MH_BC_PERIODIC, MH_BC_PERIODIC, b->stream));
    B_TRY(hipMemcpyAsync(b->u[0], b->u_init, b->field_doubles * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
    B_TRY(hipStreamSynchronize(b->stream));
#undef B_TRY
    memset(&b->state, 0, sizeof(b->state));
    *out = b;
    return MH_OK;
}

// the ghost rows of the initial solution (collective over the ranks, like the communicator it needs): a caller may step right after create
static int band_initial_exchange(mh_binary* b)
{
    mh_binary* m[1] = {b};
    if (int rc = team_exchange(Team{m, 1}, 0)) return rc;
    MH_HIP_TRY(hipStreamSynchronize(b->stream));
    return MH_OK;
}

int mh_binary_create(mh_binary** out, int device, const mh_binary_desc* d, const mh_binary_run* run, const double* xv, const double* yv,
                     const double* u_init_aos, const double* br)
{
    return binary_create_common(out, device, d, run, xv, yv, u_init_aos, br, 0, 1, BAND_NONE, nullptr);
}

int mh_binary_band_create(mh_binary** out, int device, const mh_binary_desc* d, const mh_binary_run* run, const double* xv, const double* yv,
                          const double* u_init_aos, const double* br, int rank, int world, const void* comm_id128, int self_exchange)
{
    mh_binary* b = nullptr;
    if (int rc = binary_create_common(&b, device, d, run, xv, yv, u_init_aos, br, rank, world, BAND_RCCL, nullptr, world == 1 && self_exchange)) return rc;
    // comm_id128 == NULL on a band with neighbours defers the communicator (mh_binary_band_use_comm), as mh_slab_create does
    if (b->banded && comm_id128)
    {
        RcclApi* api = rccl();
        if (! api) { mh_binary_destroy(b); set_error("binary bands: librccl.so.1 could not be loaded"); return MH_E_STATE; }
        ncclUniqueId id;
        memcpy(&id, comm_id128, sizeof id);
        ncclResult_t r = api->CommInitRank(&b->comm, world, id, rank);
        if (r != ncclSuccess) { mh_binary_destroy(b); return rccl_fail(r, "ncclCommInitRank"); }
        if (int rc = band_initial_exchange(b)) { mh_binary_destroy(b); return rc; }
    }
    *out = b;
    return MH_OK;
}

int mh_binary_band_use_comm(mh_binary* b, mh_comm* c)
{
    if (! b || ! c) return MH_E_INVALID;
    if (! b->banded && ! b->tdist) return MH_OK;
    if (b->backend != BAND_RCCL) { set_error("mh_binary_band_use_comm: not an RCCL band"); return MH_E_STATE; }
    if (b->comm) { set_error("mh_binary_band_use_comm: the band has a communicator already"); return MH_E_STATE; }
    if (c->world != b->world || c->rank != b->rank || c->device != b->device)
    {
        set_error("mh_binary_band_use_comm: communicator is rank %d of %d on device %d, the band rank %d of %d on device %d", c->rank, c->world, c->device, b->rank, b->world, b->device);
        return MH_E_INVALID;
    }
    b->comm = c->comm;
    b->owns_comm = false;
    return b->tdist ? MH_OK : band_initial_exchange(b);
}

int mh_binary_band_set_edge_rows(mh_binary* b, int rows)
{
    if (! b) return MH_E_INVALID;
    if (! b->banded || b->tree) { set_error("mh_binary_band_set_edge_rows: not a band with neighbours"); return MH_E_STATE; }
    if (rows < 0) { b->edge = binary_edge_rows(b->n0); return MH_OK; }
    if (rows != 0 && (rows < 2 || b->n0 - 2 * rows < 1)) { set_error("mh_binary_band_set_edge_rows: %d edge rows per side do not fit a band of %d rows (2 <= rows, 2 rows < n0)", rows, b->n0); return MH_E_INVALID; }
    b->edge = rows;
    return MH_OK;
}

int mh_binary_last_failure(const mh_binary* b, mh_step_result* result)
{
    if (! b || ! result) return MH_E_INVALID;
    *result = b->last_failure;
    return MH_OK;
}

int mh_binary_group_create(mh_binary** bands, int world, int device, const mh_binary_desc* d, const mh_binary_run* run, const double* xv,
                           const double* yv, const double* u_init_aos, const double* br)
{
    if (! bands || world < 1 || world > 64) { set_error("binary group: need 1..64 bands"); return MH_E_INVALID; }
    for (int r = 0; r < world; ++r) bands[r] = nullptr;
    for (int r = 0; r < world; ++r)
        if (int rc = binary_create_common(&bands[r], device, d, run, xv, yv, u_init_aos, br, r, world, BAND_LOOPBACK, r == 0 ? nullptr : bands[0]->stream))
        {
            for (int q = r - 1; q >= 0; --q) { mh_binary_destroy(bands[q]); bands[q] = nullptr; }
            return rc;
        }
    for (int r = 0; r < world; ++r)
    {
        bands[r]->peer_lo = bands[(r + world - 1) % world];
        bands[r]->peer_hi = bands[(r + 1) % world];
    }
    // the ghost rows of the initial solution
    const Team t = {bands, world};
    if (int rc = team_exchange(t, 0)) return rc;
    MH_HIP_TRY(hipStreamSynchronize(bands[0]->stream));
    return MH_OK;
}

int mh_binary_band_rows(const mh_binary* b, int* row0, int* row1)
{
    if (! b) return MH_E_INVALID;
    if (row0) *row0 = b->row0;
    if (row1) *row1 = b->row0 + b->n0;
    return MH_OK;
}

} // extern "C"

// rank, world, backend: a member of a distributed tree (world == 1 with BAND_RCCL: the RCCL calls of the fetch, to self); BAND_NONE: one domain
static int binary_tree_create_common(mh_binary** out, int device, const mh_binary_desc* d, const mh_binary_run* run, const mh_tree_block* blocks_in, int nblocks,
                                     const double* edges_in, const double* u_init_aos, const double* br_in, int rank, int world, int backend, hipStream_t shared_stream)
{
    const mh_tree_block* blocks = blocks_in;
    const double* edges = edges_in;
    const double* br = br_in;
    if (! out || ! d || ! run || ! blocks || ! edges || ! u_init_aos || ! br || nblocks < 1) { set_error("binary tree create: null argument"); return MH_E_INVALID; }
    if (d->block_size < 2 || d->block_size % 2 != 0) { set_error("binary tree: block_size must be even"); return MH_E_INVALID; }
    if (run->rk_order != 1 && run->rk_order != 2) { set_error("binary::next_solution: rk_order must be 1 or 2"); return MH_E_INVALID; }
    if (! (d->mach_number > 0.0) || ! (d->sink_radius > 0.0) || ! (d->domain_radius > 0.0)) { set_error("binary: mach_number, sink_radius and domain_radius must be positive"); return MH_E_INVALID; }
    const int bs = d->block_size, nb = nblocks;
    // a distributed tree is stored along the Hilbert curve through its leaves, so that a member's blocks are one run of every array
    const bool dist = backend != BAND_NONE;
    std::vector<int32_t> perm, order;
    std::vector<mh_tree_block> blocks_p;
    std::vector<double> edges_p, br_p;
    if (dist)
    {
        if (rank < 0 || rank >= world || world > 64) { set_error("binary tree: rank %d of %d", rank, world); return MH_E_INVALID; }
        perm.resize(nb); order.resize(nb);
        if (int rc = binary_tree_curve_order(blocks_in, nb, perm.data())) return rc;
        blocks_p.resize(nb); edges_p.resize((size_t) nb * 2 * (bs + 1)); br_p.resize((size_t) nb * bs * bs);
        for (int k = 0; k < nb; ++k)
        {
            order[perm[k]] = k;
            blocks_p[k] = blocks_in[perm[k]];
            memcpy(&edges_p[(size_t) k * 2 * (bs + 1)], edges_in + (size_t) perm[k] * 2 * (bs + 1), (size_t) 2 * (bs + 1) * sizeof(double));
            memcpy(&br_p[(size_t) k * bs * bs], br_in + (size_t) perm[k] * bs * bs, (size_t) bs * bs * sizeof(double));
        }
        blocks = blocks_p.data(); edges = edges_p.data(); br = br_p.data();
    }
    std::vector<int32_t> topo((size_t) nb * 12), level(nb);
    if (int rc = binary_tree_topology(blocks, nb, topo.data())) return rc;
    for (int k = 0; k < nb; ++k) level[k] = blocks[k].level;
    MH_HIP_TRY(hipSetDevice(device));
    mh_binary* b = new mh_binary();
    b->device = device;
    b->desc = *d;
    b->desc.n = 0;
    b->run = *run;
    b->tree = true;
    b->rank = rank; b->world = world;
    b->backend = backend;
    b->tdist = dist;
    b->perm = perm;
    if (dist)
    {
        b->tcut.resize(world + 1);
        for (int r = 0; r < world; ++r)
        {
            size_t a, e;
            mh_partition_rows((size_t) nb, (size_t) world, (size_t) r, &a, &e);          // runs of equal length (every block has bs^2 cells), nd::partition_shape's formula
            b->tcut[r] = (int) a; b->tcut[r + 1] = (int) e;
        }
        b->tb0 = b->tcut[rank]; b->tb1 = b->tcut[rank + 1];
    }
    b->field_doubles = (size_t) nb * 3 * bs * bs;
    b->host_staging.resize(b->field_doubles);
    auto fail = [&] (hipError_t e, const char* what) { mh_binary_destroy(b); return hip_fail(e, what); };
#define B_TRY(call) do { hipError_t _e = (call); if (_e != hipSuccess) return fail(_e, #call); } while (0)
    if (shared_stream) { b->stream = shared_stream; b->owns_stream = false; }
    else B_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    for (int k = 0; k < 3; ++k) B_TRY(hipMalloc(&b->u[k], b->field_doubles * sizeof(double)));
    if (dist)
    {
        B_TRY(hipMalloc(&b->order_dev, (size_t) nb * sizeof(int32_t)));
        B_TRY(hipMalloc(&b->ids_dev, (size_t) nb * sizeof(int32_t)));
        B_TRY(hipMemcpyAsync(b->order_dev, order.data(), (size_t) nb * sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
        B_TRY(hipMemcpyAsync(b->ids_dev, perm.data(), (size_t) nb * sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
        if (backend == BAND_RCCL)
        {
            B_TRY(hipMalloc(&b->reduced_dev, sizeof(HostMirror)));
            B_TRY(hipMalloc(&b->gather_dev, (size_t) world * 2 * sizeof(uint32_t)));
            B_TRY(hipHostMalloc((void**) &b->gather_host, (size_t) world * 2 * sizeof(uint32_t), hipHostMallocDefault));
        }
    }
    B_TRY(hipMalloc(&b->u_init, b->field_doubles * sizeof(double)));
    B_TRY(hipMalloc(&b->br, (size_t) nb * bs * bs * sizeof(double)));
    B_TRY(hipMalloc(&b->topo_dev, topo.size() * sizeof(int32_t)));
    B_TRY(hipMalloc(&b->level_dev, level.size() * sizeof(int32_t)));
    B_TRY(hipMalloc(&b->edges_dev, (size_t) nb * 2 * (bs + 1) * sizeof(double)));
    B_TRY(hipMalloc(&b->work.prim, b->field_doubles * sizeof(double)));
    B_TRY(hipMalloc(&b->work.gx, b->field_doubles * sizeof(double)));
    B_TRY(hipMalloc(&b->work.gy, b->field_doubles * sizeof(double)));
    B_TRY(hipMalloc(&b->work.fx, (size_t) nb * 3 * (bs + 1) * bs * sizeof(double)));
    B_TRY(hipMalloc(&b->work.fy, (size_t) nb * 3 * (bs + 1) * bs * sizeof(double)));
    B_TRY(hipMalloc(&b->work.block_out, (size_t) nb * ((bs * bs + 255) / 256) * 16 * sizeof(double)));      // [nb][tiles of 256 cells][16 partial sums]
    B_TRY(hipMalloc(&b->work.block_vals, (size_t) nb * MH_BINARY_NTOTALS * sizeof(double)));
    B_TRY(hipMalloc(&b->work.tile_maxw, (size_t) nb * ((bs * bs + 255) / 256) * sizeof(double)));
    B_TRY(hipMalloc(&b->small[0], sizeof(HostMirror)));
    use_parity(b, 0);
    B_TRY(hipHostMalloc((void**) &b->mirror, sizeof(HostMirror), hipHostMallocDefault));
    b->geom = {b->topo_dev, b->level_dev, b->edges_dev, nb, bs};
    B_TRY(hipMemcpyAsync(b->topo_dev, topo.data(), topo.size() * sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
    B_TRY(hipMemcpyAsync(b->level_dev, level.data(), level.size() * sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
    B_TRY(hipMemcpyAsync(b->edges_dev, edges, (size_t) nb * 2 * (bs + 1) * sizeof(double), hipMemcpyHostToDevice, b->stream));
    B_TRY(hipMemcpyAsync(b->br, br, (size_t) nb * bs * bs * sizeof(double), hipMemcpyHostToDevice, b->stream));
    tree_to_device_layout(b, u_init_aos, b->host_staging.data());
    B_TRY(hipMemcpyAsync(b->u_init, b->host_staging.data(), b->field_doubles * sizeof(double), hipMemcpyHostToDevice, b->stream));
    B_TRY(hipMemcpyAsync(b->u[0], b->u_init, b->field_doubles * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
    B_TRY(hipStreamSynchronize(b->stream));
#undef B_TRY
    memset(&b->state, 0, sizeof(b->state));
    *out = b;
    return MH_OK;
}

extern "C" {

int mh_binary_tree_create(mh_binary** out, int device, const mh_binary_desc* d, const mh_binary_run* run, const mh_tree_block* blocks, int nblocks,
                          const double* edges, const double* u_init_aos, const double* br)
{
    return binary_tree_create_common(out, device, d, run, blocks, nblocks, edges, u_init_aos, br, 0, 1, BAND_NONE, nullptr);
}

int mh_binary_tree_band_create(mh_binary** out, int device, const mh_binary_desc* d, const mh_binary_run* run, const mh_tree_block* blocks, int nblocks,
                               const double* edges, const double* u_init_aos, const double* br, int rank, int world, const void* comm_id128)
{
    if (world < 1) { set_error("binary tree band: world %d", world); return MH_E_INVALID; }
    if (int rc = binary_tree_create_common(out, device, d, run, blocks, nblocks, edges, u_init_aos, br, rank, world, BAND_RCCL, nullptr)) return rc;
    if (! comm_id128) return MH_OK;          // the communicator follows: mh_binary_band_use_comm
    mh_binary* b = *out;
    RcclApi* api = rccl();
    if (! api) { mh_binary_destroy(b); *out = nullptr; set_error("binary tree: librccl.so.1 could not be loaded"); return MH_E_STATE; }
    ncclUniqueId id;
    memcpy(&id, comm_id128, sizeof id);
    const ncclResult_t r = api->CommInitRank(&b->comm, world, id, rank);
    if (r != ncclSuccess) { b->comm = nullptr; mh_binary_destroy(b); *out = nullptr; return rccl_fail(r, "ncclCommInitRank"); }
    return MH_OK;
}

int mh_binary_tree_group_create(mh_binary** members, int world, int device, const mh_binary_desc* d, const mh_binary_run* run, const mh_tree_block* blocks,
                                int nblocks, const double* edges, const double* u_init_aos, const double* br)
{
    if (! members || world < 1 || world > 64) { set_error("binary tree group: need 1..64 members"); return MH_E_INVALID; }
    for (int r = 0; r < world; ++r) members[r] = nullptr;
    for (int r = 0; r < world; ++r)
        if (int rc = binary_tree_create_common(&members[r], device, d, run, blocks, nblocks, edges, u_init_aos, br, r, world, BAND_LOOPBACK, r == 0 ? nullptr : members[0]->stream))
        {
            for (int q = r - 1; q >= 0; --q) { mh_binary_destroy(members[q]); members[q] = nullptr; }
            return rc;
        }
    return MH_OK;
}

int mh_binary_tree_owned_blocks(const mh_binary* b, int32_t* ids, int* count)
{
    if (! b || ! count) return MH_E_INVALID;
    if (! b->tree) { set_error("mh_binary_tree_owned_blocks: not a tree solver"); return MH_E_STATE; }
    const int b0 = b->tdist ? b->tb0 : 0, b1 = b->tdist ? b->tb1 : b->geom.nb;
    *count = b1 - b0;
    if (ids) for (int k = b0; k < b1; ++k) ids[k - b0] = b->perm.empty() ? k : b->perm[k];
    return MH_OK;
}

void mh_binary_destroy(mh_binary* b)
{
    if (! b) return;
    (void) hipSetDevice(b->device);
    if (b->stream) (void) hipStreamSynchronize(b->stream);
    if (b->side) { (void) hipStreamSynchronize(b->side); (void) hipStreamDestroy(b->side); }
    if (b->xstream) { (void) hipStreamSynchronize(b->xstream); (void) hipStreamDestroy(b->xstream); }
    for (hipEvent_t e : {b->ev_input[0], b->ev_input[1], b->ev_stage[0], b->ev_stage[1], b->ev_totals, b->ev_edge, b->ev_edone, b->ev_xchg}) if (e) (void) hipEventDestroy(e);
    for (auto& e : b->events) { (void) hipEventDestroy(e.first); (void) hipEventDestroy(e.second); }
    for (int k = 0; k < 4; ++k) (void) hipFree(b->u[k]);
    (void) hipFree(b->order_dev); (void) hipFree(b->ids_dev);
    (void) hipFree(b->u_init); (void) hipFree(b->br); (void) hipFree(b->xv); (void) hipFree(b->yv);
    (void) hipFree(b->scratch); (void) hipFree(b->small[0]); (void) hipFree(b->small[1]); (void) hipFree(b->staging);      // status lives inside the small blocks
    (void) hipFree(b->topo_dev); (void) hipFree(b->level_dev); (void) hipFree(b->edges_dev);
    (void) hipFree(b->work.prim); (void) hipFree(b->work.gx); (void) hipFree(b->work.gy); (void) hipFree(b->work.fx); (void) hipFree(b->work.fy); (void) hipFree(b->work.block_out); (void) hipFree(b->work.block_vals); (void) hipFree(b->work.tile_maxw);
    if (b->mirror) (void) hipHostFree(b->mirror);
    (void) hipFree(b->reduced_dev);
    (void) hipFree(b->gather_dev);
    if (b->gather_host) (void) hipHostFree(b->gather_host);
    if (b->comm && b->owns_comm && rccl()) rccl()->CommDestroy(b->comm);
    if (b->stream && b->owns_stream) (void) hipStreamDestroy(b->stream);
    delete b;
}

int mh_binary_set_solution(mh_binary* b, const double* u_aos, const mh_binary_state* state)
{
    if (! b || ! state) { set_error("binary set_solution: null argument"); return MH_E_INVALID; }
    MH_HIP_TRY(hipSetDevice(b->device));
    const size_t n = b->desc.n;
    if (u_aos && b->tree)
    {
        tree_to_device_layout(b, u_aos, b->host_staging.data());
        MH_HIP_TRY(hipMemcpyAsync(b->u[0], b->host_staging.data(), b->field_doubles * sizeof(double), hipMemcpyHostToDevice, b->stream));
    }
    else if (u_aos)
    {
        // u_aos is the WHOLE mesh [n][n][3]; a band takes its rows (its ghost rows: mh_binary_group_set_solution / the RCCL exchange below)
        const size_t n0 = (size_t) b->n0;
        MH_HIP_TRY(hipMemcpyAsync(b->staging, u_aos + (size_t) b->row0 * n * 3, n0 * n * 3 * sizeof(double), hipMemcpyHostToDevice, b->stream));
        MH_HIP_TRY(aos_to_soa_launch(b->staging, b->u[0], 3, b->n0, n, b->stream));
        if (! b->banded) MH_HIP_TRY(fill_ghost_rows_launch(b->u[0], 3, b->desc.n, n, MH_BC_PERIODIC, MH_BC_PERIODIC, b->stream));
    }
    else
    {
        MH_HIP_TRY(hipMemcpyAsync(b->u[0], b->u_init, b->field_doubles * sizeof(double), hipMemcpyDeviceToDevice, b->stream));
    }
    if (b->banded && b->backend == BAND_RCCL)
    {
        const Team t = {&b, 1};
        if (int rc = team_exchange(t, 0)) return rc;          // collective: every rank sets its solution
    }
    MH_HIP_TRY(hipStreamSynchronize(b->stream));
    b->state = *state;
    b->maxw_ready = false;
    return MH_OK;
}

int mh_binary_group_set_solution(mh_binary** g, int n, const double* u_aos, const mh_binary_state* state)
{
    if (int rc = check_binary_group(g, n)) return rc;
    for (int r = 0; r < n; ++r) if (int rc = mh_binary_set_solution(g[r], u_aos, state)) return rc;
    const Team t = {g, n};
    if (int rc = team_exchange(t, 0)) return rc;
    MH_HIP_TRY(hipStreamSynchronize(g[0]->stream));
    return MH_OK;
}

int mh_binary_group_get_solution(mh_binary** g, int n, double* u_aos, mh_binary_state* state)
{
    if (int rc = check_binary_group(g, n)) return rc;
    for (int r = 0; r < n; ++r)
    {
        if (r > 0 && g[r]->tdist) break;          // (every member of a distributed tree holds the whole tree)
        if (int rc = mh_binary_get_solution(g[r], u_aos, r == 0 ? state : nullptr)) return rc;
    }
    return MH_OK;
}

int mh_binary_get_solution(mh_binary* b, double* u_aos, mh_binary_state* state)
{
    if (! b) { set_error("binary get_solution: null solver"); return MH_E_INVALID; }
    MH_HIP_TRY(hipSetDevice(b->device));
    if (u_aos && b->tree)
    {
        MH_HIP_TRY(hipMemcpyAsync(b->host_staging.data(), b->u[0], b->field_doubles * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        MH_HIP_TRY(hipStreamSynchronize(b->stream));
        const int bs = b->geom.bs;
        for (int k = 0; k < b->geom.nb; ++k)                         // device [nb][3][bs][bs] -> host [nb][bs][bs][3]
        {
            const size_t to = b->perm.empty() ? (size_t) k : (size_t) b->perm[k];
            for (int q = 0; q < 3; ++q)
                for (int c = 0; c < bs * bs; ++c)
                    u_aos[(to * bs * bs + c) * 3 + q] = b->host_staging[((size_t) k * 3 + q) * bs * bs + c];
        }
    }
    else if (u_aos)
    {
        // into the WHOLE-mesh host array: a band writes its own rows only
        const size_t n = b->desc.n, n0 = (size_t) b->n0;
        MH_HIP_TRY(soa_to_aos_launch(b->u[0], b->staging, 3, b->n0, n, b->stream));
        MH_HIP_TRY(hipMemcpyAsync(u_aos + (size_t) b->row0 * n * 3, b->staging, n0 * n * 3 * sizeof(double), hipMemcpyDeviceToHost, b->stream));
        MH_HIP_TRY(hipStreamSynchronize(b->stream));
    }
    if (state) *state = b->state;
    return MH_OK;
}

int mh_binary_next(mh_binary* b, int nsteps, int* safe_mode_steps)
{
    if (! b || nsteps < 0) { set_error("binary next: bad argument"); return MH_E_INVALID; }
    if (b->backend == BAND_LOOPBACK) { set_error("binary next: member of a loopback group (use mh_binary_group_next)"); return MH_E_STATE; }
    MH_HIP_TRY(hipSetDevice(b->device));
    const Team t = {&b, 1};
    return team_next(t, nsteps, safe_mode_steps);
}

static int check_binary_group(mh_binary** g, int n)
{
    if (! g || n < 1 || n > 64) { set_error("binary group: need 1..64 bands"); return MH_E_INVALID; }
    for (int r = 0; r < n; ++r)
        if (! g[r] || g[r]->world != n || g[r]->rank != r || (n > 1 && g[r]->backend != BAND_LOOPBACK))
        { set_error("binary group: band %d is not member %d of a loopback group of %d", r, r, n); return MH_E_INVALID; }
    return MH_OK;
}

int mh_binary_group_next(mh_binary** g, int n, int nsteps, int* safe_mode_steps)
{
    if (int rc = check_binary_group(g, n)) return rc;
    if (nsteps < 0) { set_error("binary next: bad argument"); return MH_E_INVALID; }
    MH_HIP_TRY(hipSetDevice(g[0]->device));
    const Team t = {g, n};
    return team_next(t, nsteps, safe_mode_steps);
}

double mh_binary_last_dt(const mh_binary* b) { return b ? b->last_dt : 0.0; }

const double* mh_binary_field_ptr(mh_binary* b) { return b ? b->u[0] : nullptr; }

int mh_binary_profile(mh_binary* b, int enable, double* avg_stage_ms, int* nlaunches)
{
    if (! b) return MH_E_INVALID;
    MH_HIP_TRY(hipSetDevice(b->device));
    if (avg_stage_ms || nlaunches)
    {
        MH_HIP_TRY(hipStreamSynchronize(b->stream));
        double total = 0.0;
        int count = 0;
        for (size_t k = 0; k < b->events.size(); ++k)
        {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, b->events[k].first, b->events[k].second) != hipSuccess) continue;          // (a call that failed before its last stage)
            total += ms;
            count += b->event_stages[k];
        }
        if (avg_stage_ms) *avg_stage_ms = count == 0 ? 0.0 : total / count;
        if (nlaunches) *nlaunches = count;
    }
    for (auto& e : b->events) { (void) hipEventDestroy(e.first); (void) hipEventDestroy(e.second); }
    b->events.clear();
    b->event_stages.clear();
    b->profile = enable != 0;
    return MH_OK;
}

// The stage buffers u[1], u[2] hold nothing between steps: the diagnostics use them as scratch.
int mh_binary_disk_totals(mh_binary* b, double* disk_mass, double* disk_angular_momentum)
{
    if (! b) { set_error("binary disk_totals: null solver"); return MH_E_INVALID; }
    if (b->banded || b->tdist) { set_error("binary disk_totals: not built for band / distributed-tree decompositions (gather the solution and use a whole-mesh solver)"); return MH_E_STATE; }
    MH_HIP_TRY(hipSetDevice(b->device));
    const size_t ncell = b->tree ? (size_t) b->geom.nb * b->geom.bs * b->geom.bs : (size_t) b->desc.n * b->desc.n;
    double* partial = b->u[2];
    double* out = partial + 2 * ((ncell + 255) / 256);
    MH_HIP_TRY(binary_diag_sums_launch(b->u[0], b->xv, b->yv, b->edges_dev, b->desc.n, b->geom.nb, b->geom.bs, b->tree, b->desc.angmom_form != 0,
                                       partial, out, b->stream));
    double host[2];
    MH_HIP_TRY(hipMemcpyAsync(host, out, sizeof host, hipMemcpyDeviceToHost, b->stream));
    MH_HIP_TRY(hipStreamSynchronize(b->stream));
    if (disk_mass) *disk_mass = host[0];
    if (disk_angular_momentum) *disk_angular_momentum = host[1];
    return MH_OK;
}

int mh_binary_diagnostic_fields(mh_binary* b, double* sigma, double* radial_velocity, double* phi_velocity)
{
    if (! b) { set_error("binary diagnostic_fields: null solver"); return MH_E_INVALID; }
    if (b->banded || b->tdist) { set_error("binary diagnostic_fields: not built for band / distributed-tree decompositions (gather the solution and use a whole-mesh solver)"); return MH_E_STATE; }
    MH_HIP_TRY(hipSetDevice(b->device));
    const size_t ncell = b->tree ? (size_t) b->geom.nb * b->geom.bs * b->geom.bs : (size_t) b->desc.n * b->desc.n;
    double* fields = b->u[1];
    MH_HIP_TRY(binary_diag_fields_launch(b->u[0], b->xv, b->yv, b->edges_dev, b->desc.n, b->geom.nb, b->geom.bs, b->tree, b->desc.angmom_form != 0,
                                         fields, b->stream));
    double* dst[3] = {sigma, radial_velocity, phi_velocity};
    for (int k = 0; k < 3; ++k)
        if (dst[k]) MH_HIP_TRY(hipMemcpyAsync(dst[k], fields + k * ncell, ncell * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    MH_HIP_TRY(hipStreamSynchronize(b->stream));
    return MH_OK;
}

} // extern "C"

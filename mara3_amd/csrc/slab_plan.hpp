// The decisions of a slab decomposition that do not touch a device (pure host code, no HIP): which rows a rank owns, who its neighbours are,
// how many ghost rows travel and how often, and the messages of one ghost exchange IN ISSUE ORDER. ONE place decides this for
//   * the native stepper (slab.hip: exchange_rccl / exchange_loopback / the edge and interior launches),
//   * the Python stepper over torch.distributed (mara3_amd/slab.py: bench.py's fallback and the world-2 / world-3 gloo tests on CPU),
//     which reads the same plan through the C ABI (mh_slab_plan_make) instead of restating it.
// Reference: the cut is nd::partition_shape (src/core_ndarray.hpp:820-836) as mara::evaluate_on<N> uses it (src/app_parallel.hpp:75-103);
// the reference itself exchanges nothing (its slabs share one address space).
#pragma once
#include <stddef.h>
#include "../../include/mara_hip.h"

namespace mh {

inline int slab_plan_make(int nrows_global, int world, int rank, int periodic, int self_exchange, int rk_order, int fused_cut, mh_slab_plan* out)
{
    if (! out || world < 1 || rank < 0 || rank >= world || nrows_global < world || (rk_order != 1 && rk_order != 2)) return MH_E_INVALID;
    mh_slab_plan p = {};
    size_t a = 0, b = 0;
    mh_partition_rows((size_t) nrows_global, (size_t) world, (size_t) rank, &a, &b);
    p.row0 = (int) a; p.row1 = (int) b;
    const int n0 = p.row1 - p.row0;
    const bool wrap = periodic && (world > 1 || self_exchange);
    p.lo = rank > 0 ? rank - 1 : (wrap ? world - 1 : -1);
    p.hi = rank < world - 1 ? rank + 1 : (wrap ? 0 : -1);
    // two launches per RK stage: two ghost rows per side after every stage; the one-launch RK2 step across the cuts recomputes the neighbours'
    // first-stage rows from FOUR of their step-start rows: one exchange per step (euler2d_fused.hip, cloud_fused.hip)
    p.ghost_rows = fused_cut ? 4 : 2;
    p.exchanges_per_step = fused_cut ? 1 : rk_order;
    p.edge_rows = (p.lo >= 0 || p.hi >= 0) ? p.ghost_rows : 0;          // the rows a neighbour needs: they are stepped first, their exchange rides beside the interior
    const int G = p.ghost_rows;
    int n = 0;
    // sends first, low rows first; the receives mirror the NEIGHBOURS' send order (their low rows arrive in my high ghosts first), which matters
    // when lo == hi (two ranks on a periodic axis, or a rank exchanging with itself): messages between one pair of ranks match in issue order
    if (p.lo >= 0) p.msg[n++] = {1, p.lo, 0, G};                  // my rows 0 .. G-1            -> lo
    if (p.hi >= 0) p.msg[n++] = {1, p.hi, n0 - G, G};             // my rows n0-G .. n0-1        -> hi
    if (p.hi >= 0) p.msg[n++] = {0, p.hi, n0, G};                 // ghosts n0 .. n0+G-1         <- hi's rows 0 .. G-1
    if (p.lo >= 0) p.msg[n++] = {0, p.lo, -G, G};                 // ghosts -G .. -1             <- lo's rows n0-G .. n0-1
    p.nmsg = n;
    *out = p;
    return MH_OK;
}

} // namespace mh

// Internal: host-side scalar bookkeeping of the `binary` solution (binary_host.cpp, compiled with the host compiler).
#pragma once
#include "../../include/mara_hip.h"

namespace mh {

int  binary_apply_totals(const mh_binary_state& S, const mh_two_body_t& bodies, const double totals[MH_BINARY_NTOTALS], double dt,
                         bool no_accretion_force, double begin_live_binary, mh_binary_state* out);
void binary_grid_data(const mh_binary_model* m, int n, const double* xv, const double* yv, double* u_init, double* br, double* min_dx, double* min_dy, double* max_v);
void set_error(const char* fmt, ...);
int  binary_tree_topology(const mh_tree_block* blocks, int nblocks, int32_t* topo);
int  binary_tree_curve_order(const mh_tree_block* blocks, int nblocks, int32_t* order);
void binary_combine_scalars(const mh_binary_state& a, const mh_binary_state& b, mh_binary_state* out);

} // namespace mh

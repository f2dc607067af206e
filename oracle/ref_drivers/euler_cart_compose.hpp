/**
 * TEST INFRASTRUCTURE — the uniform-cartesian Euler PLM+HLLE step composed from the reference's own headers, shared by the
 * reference-side drivers euler_cart_ref.cpp (golden vectors, long runs) and integration_ref.cpp (the boundary compiled against the
 * reference's types). The composition follows subprog_cloud.cpp:511-584 (advance) and :676-697 (next_solution); see euler_cart_ref.cpp.
 */
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <numeric>
#include <vector>
#include <string>
#include "core_ndarray.hpp"
#include "core_ndarray_ops.hpp"
#include "core_dimensional.hpp"
#include "core_sequence.hpp"
#include "math_interpolation.hpp"
#include "physics_euler.hpp"
#include "app_parallel.hpp"

namespace euler_cart {

using cons_t = mara::euler::conserved_density_t;
using prim_t = mara::euler::primitive_t;

struct params_t
{
    double gamma, theta, dt;
    double dl[3];
    int bc;
};

template<std::size_t Rank>
using cons_array_t = nd::shared_array<cons_t, Rank>;

template<std::size_t Rank>
static cons_array_t<Rank> advance(cons_array_t<Rank> u0, params_t par)
{
    auto c2p = [g=par.gamma] (cons_t U) { return mara::euler::recover_primitive(U, g, 0.0); };
    auto p0 = u0 | nd::map(c2p) | nd::to_shared();

    auto godunov_flux_difference = [&] (std::size_t axis)
    {
        auto nh = mara::unit_vector_t::on_axis(axis);
        auto riemann = [nh, g=par.gamma] (prim_t pl, prim_t pr) { return mara::euler::riemann_hlle(pl, pr, nh, g); };
        auto L = nd::select_axis(axis).from(0).to(1).from_the_end();
        auto R = nd::select_axis(axis).from(1).to(0).from_the_end();
        auto dtdl = mara::make_time(par.dt) / mara::make_length(par.dl[axis]);

        if (par.theta < 0.0) // piecewise constant
        {
            auto F = (par.bc == 1
                ? (p0 | nd::extend_periodic_on_axis(axis, 1) | nd::to_shared())
                : (p0 | nd::extend_zero_gradient(axis)       | nd::to_shared()))
            | nd::zip_adjacent2_on_axis(axis) | nd::apply(riemann) | nd::to_shared();
            return (F | nd::difference_on_axis(axis)) * dtdl | nd::to_shared();
        }

        auto plm = [t=par.theta] (prim_t a, prim_t b, prim_t c) { return mara::plm_gradient(a, b, c, t); };

        if (par.bc == 1)
        {
            auto pe = p0 | nd::extend_periodic_on_axis(axis, 2) | nd::to_shared();
            auto G  = pe | nd::zip_adjacent3_on_axis(axis) | nd::apply(plm) | nd::to_shared();
            auto pi = pe | nd::select_axis(axis).from(1).to(1).from_the_end() | nd::to_shared();
            auto F  = nd::zip((pi | L) + (G | L) * 0.5, (pi | R) - (G | R) * 0.5) | nd::apply(riemann) | nd::to_shared();
            return (F | nd::difference_on_axis(axis)) * dtdl | nd::to_shared();
        }
        auto pe = p0 | nd::extend_zero_gradient(axis) | nd::to_shared();
        auto G  = pe | nd::zip_adjacent3_on_axis(axis) | nd::apply(plm) | nd::extend_zeros(axis) | nd::to_shared();
        auto F  = nd::zip((pe | L) + (G | L) * 0.5, (pe | R) - (G | R) * 0.5) | nd::apply(riemann) | nd::to_shared();
        return (F | nd::difference_on_axis(axis)) * dtdl | nd::to_shared();
    };

    if constexpr (Rank == 1)
    {
        return (u0 - godunov_flux_difference(0)) | nd::to_shared();
    }
    else if constexpr (Rank == 2)
    {
        return (u0 - (godunov_flux_difference(0) + godunov_flux_difference(1))) | nd::to_shared();
    }
    else
    {
        return (u0 - (godunov_flux_difference(0) + godunov_flux_difference(1) + godunov_flux_difference(2))) | nd::to_shared();
    }
}

// The same step composed AS LAZILY AS THE REFERENCE'S OWN `advance` (subprog_cloud.cpp:511-584): only the primitives, the PLM gradients and the
// result are materialised, and the primitives and the result go through the evaluator the caller hands in - nd::to_shared(), or the
// reference's threaded twin mara::evaluate_on<N>() (app_parallel.hpp:72-103: N std::threads over nd::partition_shape slabs), which upstream
// pipes at exactly these places (`| evaluate`, :525-533, :582). The gradients keep upstream's serial nd::to_shared() (:566). PLM with
// zero-gradient sides (BASELINE configs 2 and 5); element for element the operations of advance() above, so the bits are the same.
template<std::size_t Rank, class Evaluator>
static cons_array_t<Rank> advance_as_upstream_evaluates(cons_array_t<Rank> u0, params_t par, Evaluator evaluate)
{
    auto c2p = [g=par.gamma] (cons_t U) { return mara::euler::recover_primitive(U, g, 0.0); };
    auto p0 = u0 | nd::map(c2p) | evaluate;
    auto plm = [t=par.theta] (prim_t a, prim_t b, prim_t c) { return mara::plm_gradient(a, b, c, t); };

    auto godunov_flux_difference = [&] (std::size_t axis)
    {
        auto nh = mara::unit_vector_t::on_axis(axis);
        auto riemann = [nh, g=par.gamma] (prim_t pl, prim_t pr) { return mara::euler::riemann_hlle(pl, pr, nh, g); };
        auto L = nd::select_axis(axis).from(0).to(1).from_the_end();
        auto R = nd::select_axis(axis).from(1).to(0).from_the_end();
        auto dtdl = mara::make_time(par.dt) / mara::make_length(par.dl[axis]);
        auto pe = p0 | nd::extend_zero_gradient(axis);
        auto G  = pe | nd::zip_adjacent3_on_axis(axis) | nd::apply(plm) | nd::extend_zeros(axis) | nd::to_shared();
        auto F  = nd::zip((pe | L) + (G | L) * 0.5, (pe | R) - (G | R) * 0.5) | nd::apply(riemann);
        return (F | nd::difference_on_axis(axis)) * dtdl;
    };
    if constexpr (Rank == 1)      return (u0 - godunov_flux_difference(0)) | evaluate;
    else if constexpr (Rank == 2) return (u0 - (godunov_flux_difference(0) + godunov_flux_difference(1))) | evaluate;
    else                          return (u0 - (godunov_flux_difference(0) + godunov_flux_difference(1) + godunov_flux_difference(2))) | evaluate;
}

// f(evaluator) with the reference's threaded evaluator on `threads` threads (a template parameter upstream: MARA_PREFERRED_THREAD_COUNT)
template<class F>
static auto with_upstream_evaluator(int threads, F f)
{
    switch (threads)
    {
        case 1:  return f(mara::evaluate_on<1>());
        case 2:  return f(mara::evaluate_on<2>());
        case 4:  return f(mara::evaluate_on<4>());
        case 8:  return f(mara::evaluate_on<8>());
        case 16: return f(mara::evaluate_on<16>());
        case 32: return f(mara::evaluate_on<32>());
        default: return f(nd::to_shared());
    }
}

} // namespace euler_cart

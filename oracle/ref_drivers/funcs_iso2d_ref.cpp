/**
 * TEST INFRASTRUCTURE — reference-side per-function oracle driver for mara::iso2d
 * (SURVEY.md §8a rows a7-a9). Applies the reference header functions of
 * physics_iso2d.hpp to arrays read from a raw f64 file. Tuples are read and
 * written through mara::get<I> (never by address: std::tuple layout is
 * ABI dependent).
 *
 * usage: funcs_iso2d_ref <mode> <n> <axis> <in.f64> <out.f64>
 *   p2c    in [n][3] P                    out [n][3] U
 *   c2p    in [n][3] U                    out [n][4] (P, threw)
 *   p2q    in [n][5] (P, x0, x1)          out [n][3] Q            to_conserved_angmom_per_area
 *   q2p    in [n][5] (Q, x0, x1)          out [n][4] (P, threw)   recover_primitive(Q, x)
 *   flux   in [n][4] (P, cs2)             out [n][3]
 *   lam    in [n][4] (P, cs2)             out [n][3] (minus, plus, max_wavespeed)
 *   hlle   in [n][8] (Pl, Pr, cs2l, cs2r) out [n][3]
 *   hllc   in [n][8] (Pl, Pr, cs2l, cs2r) out [n][5] (F, contact speed, threw)
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <numeric>
#include <string>
#include <vector>
#include <limits>
#include "core_ndarray.hpp"
#include "core_dimensional.hpp"
#include "core_sequence.hpp"
#include "core_tuple.hpp"
#include "core_geometric.hpp"
#include "physics_iso2d.hpp"

using prim_t = mara::iso2d::primitive_t;
static prim_t load_prim(const double* x) { prim_t p; for (int q = 0; q < 3; ++q) p[q] = x[q]; return p; }
static const double nan_ = std::numeric_limits<double>::quiet_NaN();

int main(int argc, char** argv)
{
    if (argc != 6) return 1;
    std::string mode = argv[1];
    std::size_t n = std::atol(argv[2]);
    auto nh = mara::unit_vector_t::on_axis(std::size_t(std::atoi(argv[3])));
    std::size_t width = (mode == "p2c" || mode == "c2p") ? 3 : (mode == "p2q" || mode == "q2p") ? 5 : (mode == "flux" || mode == "lam") ? 4 : 8;
    std::vector<double> in(n * width), out;
    FILE* f = std::fopen(argv[4], "rb");
    if (! f || std::fread(in.data(), sizeof(double), in.size(), f) != in.size()) return 2;
    std::fclose(f);

    for (std::size_t i = 0; i < n; ++i)
    {
        const double* x = &in[i * width];
        if (mode == "p2c")
        {
            auto U = load_prim(x).to_conserved_per_area();
            out.push_back(mara::get<0>(U).value); out.push_back(mara::get<1>(U).value); out.push_back(mara::get<2>(U).value);
        }
        else if (mode == "c2p")
        {
            auto U = mara::iso2d::conserved_per_area_t()
            .set<0>(mara::make_dimensional<-2, 1, 0>(x[0])).set<1>(mara::make_dimensional<-1, 1, -1>(x[1])).set<2>(mara::make_dimensional<-1, 1, -1>(x[2]));
            try { auto p = mara::iso2d::recover_primitive(U); for (int q = 0; q < 3; ++q) out.push_back(p[q]); out.push_back(0.0); }
            catch (const std::exception&) { for (int q = 0; q < 3; ++q) out.push_back(nan_); out.push_back(1.0); }
        }
        else if (mode == "p2q")
        {
            mara::iso2d::location_2d_t loc = {{mara::make_length(x[3]), mara::make_length(x[4])}};
            auto Q = load_prim(x).to_conserved_angmom_per_area(loc);
            out.push_back(mara::get<0>(Q).value); out.push_back(mara::get<1>(Q).value); out.push_back(mara::get<2>(Q).value);
        }
        else if (mode == "q2p")
        {
            mara::iso2d::location_2d_t loc = {{mara::make_length(x[3]), mara::make_length(x[4])}};
            auto Q = mara::iso2d::conserved_angmom_per_area_t()
            .set<0>(mara::make_dimensional<-2, 1, 0>(x[0])).set<1>(mara::make_dimensional<0, 1, -1>(x[1])).set<2>(mara::make_dimensional<0, 1, -1>(x[2]));
            try { auto p = mara::iso2d::recover_primitive(Q, loc); for (int q = 0; q < 3; ++q) out.push_back(p[q]); out.push_back(0.0); }
            catch (const std::exception&) { for (int q = 0; q < 3; ++q) out.push_back(nan_); out.push_back(1.0); }
        }
        else if (mode == "flux")
        {
            auto F = load_prim(x).flux(nh, x[3]);
            out.push_back(mara::get<0>(F).value); out.push_back(mara::get<1>(F).value); out.push_back(mara::get<2>(F).value);
        }
        else if (mode == "lam")
        {
            auto A = load_prim(x).wavespeeds(nh, x[3]);
            out.push_back(A.m.value); out.push_back(A.p.value); out.push_back(load_prim(x).max_wavespeed(x[3]));
        }
        else if (mode == "hlle")
        {
            auto F = mara::iso2d::riemann_hlle(load_prim(x), load_prim(x + 3), x[6], x[7], nh);
            out.push_back(mara::get<0>(F).value); out.push_back(mara::get<1>(F).value); out.push_back(mara::get<2>(F).value);
        }
        else if (mode == "hllc")
        {
            auto vars = mara::iso2d::compute_hllc_variables(load_prim(x), load_prim(x + 3), x[6], x[7], nh);
            try
            {
                auto F = vars.interface_flux();
                out.push_back(mara::get<0>(F).value); out.push_back(mara::get<1>(F).value); out.push_back(mara::get<2>(F).value);
                out.push_back(vars.contact_speed()); out.push_back(0.0);
            }
            catch (const std::exception&) { for (int q = 0; q < 3; ++q) out.push_back(nan_); out.push_back(vars.contact_speed()); out.push_back(1.0); }
        }
        else return 3;
    }
    FILE* g = std::fopen(argv[5], "wb");
    std::fwrite(out.data(), sizeof(double), out.size(), g);
    std::fclose(g);
    return 0;
}

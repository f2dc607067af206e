/**
 * TEST INFRASTRUCTURE — reference-side per-function oracle driver for mara::srhd
 * (SURVEY.md §8a rows a10-a11): applies the reference header functions to arrays
 * read from a raw f64 file.
 *
 * usage: funcs_srhd_ref <mode> <n> <param0> <param1> <in.f64> <out.f64>
 *   c2p   in [n][5] U        param0 = gamma, param1 = temperature floor   out [n][6] (P, status: 0 ok / 1 threw)
 *   p2c   in [n][5] P        param0 = gamma                               out [n][5]
 *   hlle  in [n][10] (Pl,Pr) param0 = gamma, param1 = axis                out [n][5]
 *   lam   in [n][5] P        param0 = gamma, param1 = axis                out [n][2]
 *   src   in [n][7] (P,r,q)  param0 = gamma                               out [n][5]
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
#include "core_geometric.hpp"
#include "core_rational.hpp"
#include "physics_srhd.hpp"

using prim_t = mara::srhd::primitive_t;
using cons_t = mara::srhd::conserved_density_t;
static prim_t load_prim(const double* x) { prim_t p; for (int q = 0; q < 5; ++q) p[q] = x[q]; return p; }
static cons_t load_cons(const double* x) { cons_t u; for (int q = 0; q < 5; ++q) u[q].value = x[q]; return u; }

int main(int argc, char** argv)
{
    if (argc != 7) return 1;
    std::string mode = argv[1];
    std::size_t n = std::atol(argv[2]);
    double a0 = std::atof(argv[3]), a1 = std::atof(argv[4]);
    std::size_t width = mode == "hlle" ? 10 : (mode == "src" ? 7 : 5);
    std::vector<double> in(n * width), out;
    FILE* f = std::fopen(argv[5], "rb");
    if (! f || std::fread(in.data(), sizeof(double), in.size(), f) != in.size()) return 2;
    std::fclose(f);

    for (std::size_t i = 0; i < n; ++i)
    {
        const double* x = &in[i * width];
        if (mode == "c2p")
        {
            try
            {
                auto p = mara::srhd::recover_primitive(load_cons(x), a0, a1);
                for (int q = 0; q < 5; ++q) out.push_back(p[q]);
                out.push_back(0.0);
            }
            catch (const std::exception&)
            {
                for (int q = 0; q < 5; ++q) out.push_back(std::numeric_limits<double>::quiet_NaN());
                out.push_back(1.0);
            }
        }
        else if (mode == "p2c")
        {
            auto u = load_prim(x).to_conserved_density(a0);
            for (int q = 0; q < 5; ++q) out.push_back(u[q].value);
        }
        else if (mode == "hlle")
        {
            auto F = mara::srhd::riemann_hlle(load_prim(x), load_prim(x + 5), mara::unit_vector_t::on_axis(std::size_t(a1)), a0);
            for (int q = 0; q < 5; ++q) out.push_back(F[q].value);
        }
        else if (mode == "lam")
        {
            auto A = load_prim(x).wavespeeds(mara::unit_vector_t::on_axis(std::size_t(a1)), a0);
            out.push_back(A.m.value);
            out.push_back(A.p.value);
        }
        else if (mode == "src")
        {
            auto S = load_prim(x).spherical_geometry_source_terms(x[5], x[6], a0);
            for (int q = 0; q < 5; ++q) out.push_back(S[q].value);
        }
        else return 3;
    }
    FILE* g = std::fopen(argv[6], "wb");
    std::fwrite(out.data(), sizeof(double), out.size(), g);
    std::fclose(g);
    return 0;
}

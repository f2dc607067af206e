/**
 * TEST INFRASTRUCTURE — reference-side per-function oracle driver.
 *
 * Applies individual reference header functions to
 * arrays of inputs read from a raw f64 file and writes the outputs, so that
 * golden vectors can be generated without restating any arithmetic.
 *
 * usage: funcs_ref <mode> <n> <param0> <param1> <in.f64> <out.f64>
 *   plm        in [n][3] (yl,y0,yr)   param0 = theta            out [n]
 *   euler_c2p  in [n][5] U            param0 = gamma, param1 = temperature floor   out [n][5]
 *   euler_p2c  in [n][5] P            param0 = gamma            out [n][5]
 *   euler_hlle in [n][10] (Pl,Pr)     param0 = gamma, param1 = axis   out [n][5]
 *   euler_flux in [n][5] P            param0 = gamma, param1 = axis   out [n][5]
 *   euler_lam  in [n][5] P            param0 = gamma, param1 = axis   out [n][2]
 *   decomp     (no input file: in = "-") param0 = rank, n = max blocks; out int64 [n][rank]
 *   partition  param0 = count, n = nparts; out int64 [n][2]   (nd::partition_shape via divvy formula)
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <algorithm>
#include <numeric>
#include <string>
#include <vector>
#include "core_ndarray.hpp"
#include "core_ndarray_ops.hpp"
#include "core_dimensional.hpp"
#include "core_sequence.hpp"
#include "math_interpolation.hpp"
#include "physics_euler.hpp"
#include "app_parallel.hpp"

using prim_t = mara::euler::primitive_t;
using cons_t = mara::euler::conserved_density_t;

static prim_t load_prim(const double* x) { prim_t p; for (int q = 0; q < 5; ++q) p[q] = x[q]; return p; }
static cons_t load_cons(const double* x) { cons_t u; for (int q = 0; q < 5; ++q) u[q].value = x[q]; return u; }

template<std::size_t Rank> static void decomp(std::size_t nmax, std::vector<int64_t>& out)
{
    for (std::size_t n = 1; n <= nmax; ++n)
    {
        auto s = mara::propose_block_decomposition<Rank>(n);
        for (std::size_t a = 0; a < Rank; ++a) out.push_back(int64_t(s[a]));
    }
}

int main(int argc, char** argv)
{
    if (argc != 7) return 1;
    std::string mode = argv[1];
    std::size_t n = std::atol(argv[2]);
    double a0 = std::atof(argv[3]);
    double a1 = std::atof(argv[4]);

    if (mode == "decomp" || mode == "partition" || mode == "blocks")
    {
        std::vector<int64_t> out;
        if (mode == "decomp")
        {
            if (int(a0) == 1) decomp<1>(n, out);
            if (int(a0) == 2) decomp<2>(n, out);
            if (int(a0) == 3) decomp<3>(n, out);
        }
        else if (mode == "partition") // slabs of `count` rows in n parts: use the runtime twin of partition_shape, nd::divvy
        {
            for (auto g : nd::arange(int(a0)) | nd::divvy(n))
            {
                out.push_back(g.size() ? *g.begin() : -1);
                out.push_back(g.size());
            }
        }
        else // create_access_pattern_array on a rank-1 domain of a0 cells in n blocks
        {
            auto A = mara::create_access_pattern_array(nd::make_shape(std::size_t(a0)), nd::make_shape(n));
            for (std::size_t b = 0; b < n; ++b)
            {
                out.push_back(A(b).start[0]);
                out.push_back(A(b).final[0]);
            }
        }
        FILE* g = std::fopen(argv[6], "wb");
        std::fwrite(out.data(), sizeof(int64_t), out.size(), g);
        std::fclose(g);
        return 0;
    }

    std::size_t width = mode == "plm" ? 3 : (mode == "euler_hlle" ? 10 : 5);
    std::vector<double> in(n * width), out;
    FILE* f = std::fopen(argv[5], "rb");
    if (! f || std::fread(in.data(), sizeof(double), in.size(), f) != in.size()) return 2;
    std::fclose(f);

    for (std::size_t i = 0; i < n; ++i)
    {
        const double* x = &in[i * width];

        if (mode == "plm")
        {
            out.push_back(mara::plm_gradient(x[0], x[1], x[2], a0));
        }
        else if (mode == "euler_c2p")
        {
            auto p = mara::euler::recover_primitive(load_cons(x), a0, a1);
            for (int q = 0; q < 5; ++q) out.push_back(p[q]);
        }
        else if (mode == "euler_p2c")
        {
            auto u = load_prim(x).to_conserved_density(a0);
            for (int q = 0; q < 5; ++q) out.push_back(u[q].value);
        }
        else if (mode == "euler_hlle")
        {
            auto F = mara::euler::riemann_hlle(load_prim(x), load_prim(x + 5), mara::unit_vector_t::on_axis(std::size_t(a1)), a0);
            for (int q = 0; q < 5; ++q) out.push_back(F[q].value);
        }
        else if (mode == "euler_flux")
        {
            auto F = load_prim(x).flux(mara::unit_vector_t::on_axis(std::size_t(a1)), a0);
            for (int q = 0; q < 5; ++q) out.push_back(F[q].value);
        }
        else if (mode == "euler_lam")
        {
            auto A = load_prim(x).wavespeeds(mara::unit_vector_t::on_axis(std::size_t(a1)), a0);
            out.push_back(A.m.value);
            out.push_back(A.p.value);
        }
        else return 3;
    }
    FILE* g = std::fopen(argv[6], "wb");
    std::fwrite(out.data(), sizeof(double), out.size(), g);
    std::fclose(g);
    return 0;
}

// CloudProblem::make_diagnostic_fields (src/subprog_cloud.cpp:334-433) of the device-resident `cloud` solution (SURVEY.md §8 row
// f-4): the five 2-D fields of a diagnostics file and the fifteen per-polar-angle arrays (total energy, shock radii found by
// post_shock_locator.hpp:73-170, post-shock flow power...), computed where the state lives when the write_diagnostics task is due,
// so that what crosses PCIe is the finished product, not the conserved state.
//
// Arithmetic: the STRICT primitive recovery and flux (bit-identical to the reference), IEEE division and sqrt; log and pow come from
// the device math library, so the entropy - and through it, in a near-tie, a shock index - can differ from the reference's glibc
// result in the last place. Tolerances: tests/test_gpu_cloud_diagnostics.py.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "euler_device.hpp"
#include "srhd_device.hpp"
#include "launch.hpp"

namespace mh {

struct CloudDiagParams
{
    const double* u;          // device layout [(nr + 4)][5][nq]
    const double* rv;         // [nr + 1]
    const double* dmu;        // [nq]
    int nr, nq;
    double gamma, tfloor;
    double u_length, u_energy, u_mass_density, u_energy_density, u_power;
    double* fields;           // [5][nr][nq]: mass_density, gas_pressure, specific_entropy, radial_gamma_beta, radial_energy_flow
    double* work;             // [4][nr][nq]: entropy, pressure (code units), L = energy flux x mid-cell area x power, Lorentz factor
    double* columns;          // [15][nq]
    int32_t* status;
};

__device__ inline double cloud_dAr(const double* rv, const double* dmu, int i, int j)     // radial_face_areas :260-266
{
    return rv[i] * rv[i] * dmu[j] * 2 * M_PI;
}

__global__ __launch_bounds__(256)
void cloud_diag_fields_kernel(CloudDiagParams p)
{
    const long ncell = (long) p.nr * p.nq;
    const srhd::Gamma g = srhd::make_gamma(p.gamma);
    const Recip three = make_recip(3.0, 1.0);
    int bad = 0;
    for (long c = (long) blockIdx.x * blockDim.x + threadIdx.x; c < ncell; c += (long) gridDim.x * blockDim.x)
    {
        const int i = (int) (c / p.nq), j = (int) (c - (long) i * p.nq);
        const double r0 = p.rv[i], r1 = p.rv[i + 1];
        const double d3 = r1 * r1 * r1 - r0 * r0 * r0;
        const double dv = divide(d3 * p.dmu[j] * 2 * M_PI, three);                      // cell_volumes :277-283
        double x[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) x[q] = p.u[((long) (i + 2) * 5 + q) * p.nq + j];
        divide_group<5>(x, make_recip(dv, 1.0));
        State5 U, P, Uc, F;
#pragma unroll
        for (int q = 0; q < 5; ++q) U[q] = x[q];
        bad |= srhd::recover_primitive(U, g, p.tfloor, P);
        double lm, lp;
        srhd::side<0>(P, g, Uc, F, lm, lp);                                              // p.flux(rhat, gamma)
        const double entropy = log(P[4] / pow(P[0], p.gamma));                           // specific_entropy physics_srhd.hpp:138-141
        const double a_lo = cloud_dAr(p.rv, p.dmu, i, j), a_hi = cloud_dAr(p.rv, p.dmu, i + 1, j);
        p.fields[0 * ncell + c] = P[0] * p.u_mass_density;
        p.fields[1 * ncell + c] = P[4] * p.u_energy_density;
        p.fields[2 * ncell + c] = entropy;
        p.fields[3 * ncell + c] = P[1];
        p.fields[4 * ncell + c] = (F[4] * a_lo) * p.u_power;
        p.work[0 * ncell + c] = entropy;
        p.work[1 * ncell + c] = P[4];
        p.work[2 * ncell + c] = (F[4] * ((a_lo + a_hi) * 0.5)) * p.u_power;
        p.work[3 * ncell + c] = sqrt(1.0 + srhd::gamma_beta_squared(P));
    }
    if (bad) atomicOr(p.status, bad);
}

// one thread per polar index: the scans of post_shock_locator.hpp run along the radius (coalesced across the threads of a wave)
__global__ __launch_bounds__(64)
void cloud_diag_columns_kernel(CloudDiagParams p)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= p.nq) return;
    const long ncell = (long) p.nr * p.nq;
    const unsigned nr = (unsigned) p.nr;
    const double* s0 = p.work + j;
    const double* pr = p.work + ncell + j;
    const double* L = p.work + 2 * ncell + j;
    const double* W = p.work + 3 * ncell + j;
    auto at = [nq = (long) p.nq] (const double* a, unsigned i) { return a[(long) i * nq]; };

    // find_shock_index :73-82: the first index of the minimum of the entropy difference
    unsigned mid = 0;
    {
        double prev = at(s0, 0), dsmin = 0.0;
        for (unsigned i = 0; i + 1 < nr; ++i)
        {
            const double next = at(s0, i + 1), ds = next - prev;
            if (i == 0 || ds < dsmin) { dsmin = ds; mid = i; }
            prev = next;
        }
    }
    // find_index_of_pressure_plateau_ahead :150-170; an index outside dlogp's nr - 1 entries is the reference's exception -> 0
    unsigned up = mid;
    for (;;)
    {
        const unsigned a = up - 1, b = up - 2;
        if (a >= nr - 1 || b >= nr - 1) { up = 0; break; }
        const double da = log(at(pr, a + 1)) - log(at(pr, a));
        const double db = log(at(pr, b + 1)) - log(at(pr, b));
        if (da < 0.5 * db) ++up; else break;
    }
    // find_index_of_maximum_behind :98-114 on the pressure and on L
    unsigned pi = mid, li = mid;
    for (;;)
    {
        const unsigned a = pi - 1;
        if (a >= nr || pi >= nr) { pi = 0; break; }
        if (at(pr, a) > at(pr, pi)) --pi; else break;
    }
    for (;;)
    {
        const unsigned a = li - 1;
        if (a >= nr || li >= nr) { li = 0; break; }
        if (at(L, a) > at(L, li)) --li; else break;
    }
    double total = 0.0;                                                                   // :386, nd::sum from 0
    for (unsigned i = 0; i < nr; ++i) total = total + p.u[((long) (i + 2) * 5 + 4) * p.nq + j] * p.u_energy;
    auto rc = [&] (unsigned i) { return ((p.rv[i] + p.rv[i + 1]) * 0.5) * p.u_length; };
    double* out = p.columns + j;
    const long nq = p.nq;
    out[0 * nq] = total;
    out[1 * nq] = cloud_dAr(p.rv, p.dmu, 0, j) / p.rv[0] / p.rv[0];
    out[2 * nq] = rc(mid);
    out[3 * nq] = rc(up);
    out[4 * nq] = rc(pi);
    out[5 * nq] = rc(li);
    out[6 * nq] = at(W, pi);
    out[7 * nq] = at(L, pi);
    const unsigned back[6] = {2, 4, 8, 16, 32, 64};
    for (int k = 0; k < 6; ++k) out[(8 + k) * nq] = at(L, mid > back[k] ? mid - back[k] : 0);
    out[14 * nq] = at(L, li);
}

hipError_t cloud_diagnostics_launch(const mh_cloud_desc* d, const double* geom_dev, const double* u, const double units[3],
                                    double* fields, double* work, double* columns, int32_t* status, hipStream_t stream)
{
    const double light_speed_cgs = 2.998e10;                      // subprog_cloud.cpp:50, unit_system_t :177-195
    CloudDiagParams p;
    p.u = u;
    p.rv = geom_dev;                                              // rv[nr_global + 1] | dmu[nq] | ...  (mh_cloud_pack_geometry)
    p.dmu = geom_dev + d->nr_global + 1;
    p.nr = d->nr;
    p.nq = d->nq;
    p.gamma = d->gamma;
    p.tfloor = d->temperature_floor;
    const double u_length = units[0], u_mass = units[1], u_time = units[2];
    p.u_length = u_length;
    p.u_energy = u_mass * pow(light_speed_cgs, 2);
    p.u_mass_density = u_mass / pow(u_length, 3);
    p.u_energy_density = p.u_energy / pow(u_length, 3);
    p.u_power = p.u_energy / u_time;
    p.fields = fields;
    p.work = work;
    p.columns = columns;
    p.status = status;
    const long ncell = (long) p.nr * p.nq;
    const int blocks = (int) ((ncell + 255) / 256 > 4096 ? 4096 : (ncell + 255) / 256);
    hipLaunchKernelGGL(cloud_diag_fields_kernel, dim3(blocks), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(cloud_diag_columns_kernel, dim3((p.nq + 63) / 64), dim3(64), 0, stream, p);
    return hipGetLastError();
}

} // namespace mh

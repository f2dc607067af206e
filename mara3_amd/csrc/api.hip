// C ABI of libmara_hip.so (see include/mara_hip.h for the contract and the
// reference interfaces each entry point stands in for).
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <type_traits>
#include "launch.hpp"
#include "euler_device.hpp"
#include "euler_device_fast.hpp"
#include "srhd_device.hpp"
#include "iso2d_device.hpp"

namespace mh {

static thread_local std::string g_error;

void set_error(const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
}

int hip_fail(hipError_t e, const char* what)
{
    set_error("HIP error %d (%s) in %s", (int) e, hipGetErrorString(e), what);
    // the runtime keeps the last error until somebody reads it: a REPORTED failure (e.g. a device id the box does not have) must not come
    // back out of the hipGetLastError() behind the next, healthy launch of this thread (tests/test_gpu_slab_group.py: a group that failed to
    // build, then one that steps)
    (void) hipGetLastError();
    return MH_E_HIP;
}

static int check_desc(const mh_euler_cart_desc* d)
{
    if (! d) { set_error("null descriptor"); return MH_E_INVALID; }
    if (d->rank != 2 && d->rank != 3) { set_error("mh_euler_cart: rank %d not supported (2 or 3)", d->rank); return MH_E_INVALID; }
    if (d->n[0] < 2 || d->n[1] < 2 || (d->rank == 3 && d->n[2] < 2)) { set_error("mh_euler_cart: need at least 2 cells per axis"); return MH_E_INVALID; }
    if (d->riemann != MH_RIEMANN_HLLE && d->riemann != MH_RIEMANN_HLLC) { set_error("unknown riemann solver %d", d->riemann); return MH_E_INVALID; }
    if (d->bc_transverse != MH_BC_OUTFLOW && d->bc_transverse != MH_BC_PERIODIC) { set_error("transverse bc must be outflow or periodic"); return MH_E_INVALID; }
    for (int bc : {d->bc_lo0, d->bc_hi0})
        if (bc != MH_BC_OUTFLOW && bc != MH_BC_PERIODIC && bc != MH_BC_EXTERNAL) { set_error("axis-0 bc must be outflow, periodic or external"); return MH_E_INVALID; }
    if ((d->bc_lo0 == MH_BC_PERIODIC) != (d->bc_hi0 == MH_BC_PERIODIC)) { set_error("periodic axis-0 bc must be set on both sides"); return MH_E_INVALID; }
    if (d->arith != MH_ARITH_STRICT && d->arith != MH_ARITH_FAST) { set_error("unknown arith mode %d", d->arith); return MH_E_INVALID; }
    if (!(d->gamma > 1.0)) { set_error("gamma must be > 1"); return MH_E_INVALID; }
    return MH_OK;
}

static hipError_t cart_stage_launch(const mh_euler_cart_desc* d, const double* u_in, const double* u_base, double* u_out,
                                    double dt, double weight, int row_begin, int row_end, int32_t* status, hipStream_t stream)
{
    return d->rank == 3 ? euler3d_stage_launch(d, u_in, u_base, u_out, dt, weight, row_begin, row_end, status, stream)
                        : euler2d_stage_launch(d, u_in, u_base, u_out, dt, weight, row_begin, row_end, status, stream);
}

static size_t row_pitch_of(const mh_euler_cart_desc* d) { return d->rank == 3 ? (size_t) d->n[1] * d->n[2] : (size_t) d->n[1]; }

// ---- per-function kernels (one thread per item; AoS rows) ----------------
template<class A>
__global__ void plm_kernel(size_t n, const double* yl, const double* y0, const double* yr, double theta, double* g)
{
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if constexpr (std::is_same<A, FastArith>::value) g[i] = fast::plm_gradient(yl[i], y0[i], yr[i], theta);
    else                                             g[i] = plm_gradient(yl[i], y0[i], yr[i], theta);
}
__device__ inline State5 load5(const double* p) { State5 s; for (int q = 0; q < 5; ++q) s[q] = p[q]; return s; }
__device__ inline void store5(double* p, const State5& s) { for (int q = 0; q < 5; ++q) p[q] = s[q]; }

template<class A>
__global__ void c2p_kernel(size_t n, const double* U, double gamma, double tfloor, double* P)
{
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if constexpr (std::is_same<A, FastArith>::value) store5(P + 5 * i, fast::recover_primitive(load5(U + 5 * i), fast::make_gamma_law(gamma), tfloor));
    else                                             store5(P + 5 * i, recover_primitive(load5(U + 5 * i), gamma, tfloor));
}
template<class A>
__global__ void p2c_kernel(size_t n, const double* P, double gamma, double* U)
{
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if constexpr (std::is_same<A, FastArith>::value)
    {
        State5 Uc, F;
        double vn, cs;
        fast::face_quantities<0>(load5(P + 5 * i), fast::make_gamma_law(gamma), Uc, F, vn, cs);
        store5(U + 5 * i, Uc);
    }
    else store5(U + 5 * i, to_conserved_density(load5(P + 5 * i), make_gamma_law(gamma)));
}
template<class A, int RIEMANN, int AXIS>
__global__ void riemann_kernel(size_t n, const double* Pl, const double* Pr, double gamma, double* F)
{
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) store5(F + 5 * i, A::template flux<RIEMANN, AXIS>(load5(Pl + 5 * i), load5(Pr + 5 * i), A::gamma_law(gamma)));
}

__global__ void srhd_c2p_kernel(size_t n, const double* U, double gamma, double tfloor, double* P, int32_t* status)
{
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    State5 Pi;
    status[i] = srhd::recover_primitive(load5(U + 5 * i), srhd::make_gamma(gamma), tfloor, Pi);
    store5(P + 5 * i, Pi);
}
__global__ void srhd_p2c_kernel(size_t n, const double* P, double gamma, double* U)
{
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) store5(U + 5 * i, srhd::to_conserved_density(load5(P + 5 * i), srhd::make_gamma(gamma)));
}
template<int AXIS>
__global__ void srhd_hlle_kernel(size_t n, const double* Pl, const double* Pr, double gamma, double* F)
{
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) store5(F + 5 * i, srhd::riemann_hlle<AXIS>(load5(Pl + 5 * i), load5(Pr + 5 * i), srhd::make_gamma(gamma)));
}
__global__ void srhd_src_kernel(size_t n, const double* P, const double* r, const double* cotq, double gamma, double* S)
{
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) store5(S + 5 * i, srhd::source_terms(load5(P + 5 * i), r[i], cotq[i], srhd::make_gamma(gamma)));
}

// ---- iso2d per-function kernels; mode selects the function, AXIS the face normal -------------------------
__device__ inline iso2d::State3 load3(const double* p) { iso2d::State3 s; for (int q = 0; q < 3; ++q) s[q] = p[q]; return s; }
__device__ inline void store3(double* p, const iso2d::State3& s) { for (int q = 0; q < 3; ++q) p[q] = s[q]; }

enum { ISO_P2C, ISO_C2P, ISO_P2Q, ISO_Q2P, ISO_FLUX, ISO_LAM, ISO_HLLE, ISO_HLLC };

template<int AXIS>
__global__ void iso2d_kernel(int mode, size_t n, const double* a, const double* b, const double* c, const double* d,
                             double* out, double* out2, int32_t* flag)
{
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    iso2d::State3 R;
    bool threw = false;
    switch (mode)
    {
        case ISO_P2C: R = iso2d::to_conserved(load3(a + 3 * i)); break;
        case ISO_C2P: threw = iso2d::recover_primitive(load3(a + 3 * i), R); break;
        case ISO_P2Q: R = iso2d::to_conserved_angmom(load3(a + 3 * i), b[2 * i], b[2 * i + 1]); break;
        case ISO_Q2P: threw = iso2d::recover_primitive_angmom(load3(a + 3 * i), b[2 * i], b[2 * i + 1], R); break;
        case ISO_FLUX: R = iso2d::flux<AXIS>(load3(a + 3 * i), c[i]); break;
        case ISO_LAM:
        {
            const iso2d::State3 P = load3(a + 3 * i);
            const double cs = sqrt(c[i]), vn = iso2d::velocity_along<AXIS>(P);
            R[0] = vn - cs; R[1] = vn + cs; R[2] = iso2d::max_wavespeed(P, c[i]);
            break;
        }
        case ISO_HLLE: R = iso2d::riemann_hlle<AXIS>(load3(a + 3 * i), load3(b + 3 * i), c[i], d[i]); break;
        case ISO_HLLC:
        {
            double contact;
            threw = iso2d::riemann_hllc<AXIS>(load3(a + 3 * i), load3(b + 3 * i), c[i], d[i], R, contact);
            if (out2) out2[i] = contact;
            break;
        }
    }
    store3(out + 3 * i, R);
    if (flag) flag[i] = threw ? 1 : 0;
}

} // namespace mh

using namespace mh;

struct mh_ctx
{
    int device = 0;
    hipStream_t stream = nullptr;
    bool configured = false, uploaded = false;
    enum { KIND_NONE, KIND_EULER_CART, KIND_SEDOV, KIND_CLOUD } kind = KIND_NONE;
    mh_euler_cart_desc desc;
    mh_sedov_desc sedov;
    mh_cloud_desc cloud;
    double* inflow = nullptr;                // cloud: [5][nq] inner-ghost primitives
    double* geom = nullptr;                  // sedov: dv[nz], da[nz+1], rc[nz]
    int rk_order = 2;
    size_t field_doubles = 0;
    double* field[2] = {nullptr, nullptr};   // [0] current solution, [1] stage scratch
    double* third = nullptr;                 // mh_step_checked: the step's result before it is committed (allocated on first use)
    double* staging = nullptr;               // AoS staging for upload/download
    size_t staging_doubles = 0;
    int32_t* status = nullptr;
    int32_t* planar_flag = nullptr;          // device word of the planarity check at upload (mh_euler_cart_desc.planar)
    bool planar_now = false;                 // the resident 2-D Euler field has no third momentum: the fused step takes its planar kernel
                                             // (cloud: the field has no azimuthal momentum ...
    bool inflow_planar = false;              //  ... and neither has the nozzle row of the step about to run)
    bool profile = false;
    // profile: ONE pair of events around the stage launches of each mh_step / mh_step_checked call, and how many launches lie between
    // them (events around every launch put two markers between consecutive kernels and read 3 % long on the sub-millisecond ones)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<int> event_launches;
    int span_launches = 0;
    std::string error;
};

static int ctx_fail(mh_ctx* c, int code)
{
    if (c) c->error = g_error;
    return code;
}

extern "C" {

const char* mh_last_error(const mh_ctx* ctx)
{
    return ctx && ! ctx->error.empty() ? ctx->error.c_str() : g_error.c_str();
}

size_t mh_euler_cart_field_doubles(const mh_euler_cart_desc* d)
{
    if (! d) return 0;
    return (size_t) 5 * (d->n[0] + 4) * row_pitch_of(d);
}

int mh_euler_cart_stage(const mh_euler_cart_desc* d, const double* u_in, const double* u_base, double* u_out,
                        double dt, double stage_weight, int row_begin, int row_end, int32_t* status, void* stream)
{
    if (int rc = check_desc(d)) return rc;
    if (! u_in || ! u_out || u_in == u_out) { set_error("stage: u_in and u_out must be distinct device fields"); return MH_E_INVALID; }
    if (stage_weight != 1.0 && ! u_base) { set_error("stage: combine needs u_base"); return MH_E_INVALID; }
    if (row_begin < 0 || row_end > d->n[0] || row_begin > row_end) { set_error("stage: bad row range [%d,%d)", row_begin, row_end); return MH_E_INVALID; }
    MH_HIP_TRY(cart_stage_launch(d, u_in, u_base, u_out, dt, stage_weight, row_begin, row_end, status, (hipStream_t) stream));
    return MH_OK;
}

int mh_euler_cart_fill_ghosts(const mh_euler_cart_desc* d, double* u, void* stream)
{
    if (int rc = check_desc(d)) return rc;
    MH_HIP_TRY(fill_ghost_rows_launch(u, 5, d->n[0], row_pitch_of(d), d->bc_lo0, d->bc_hi0, (hipStream_t) stream));
    return MH_OK;
}

int mh_aos_to_soa(const double* aos_dev, double* soa_dev, int nq, int n0, size_t row_pitch, void* stream)
{
    MH_HIP_TRY(aos_to_soa_launch(aos_dev, soa_dev, nq, n0, row_pitch, (hipStream_t) stream));
    return MH_OK;
}

int mh_soa_to_aos(const double* soa_dev, double* aos_dev, int nq, int n0, size_t row_pitch, void* stream)
{
    MH_HIP_TRY(soa_to_aos_launch(soa_dev, aos_dev, nq, n0, row_pitch, (hipStream_t) stream));
    return MH_OK;
}

int mh_calib_stream_copy(const double* src_dev, double* dst_dev, size_t ndoubles, void* stream)
{
    MH_HIP_TRY(stream_copy_launch(src_dev, dst_dev, ndoubles, (hipStream_t) stream));
    return MH_OK;
}

// ---- context -------------------------------------------------------------
int mh_create(mh_ctx** out, int device_id)
{
    if (! out) return MH_E_INVALID;
    int count = 0;
    MH_HIP_TRY(hipGetDeviceCount(&count));
    if (device_id < 0 || device_id >= count) { set_error("device %d not available (%d visible)", device_id, count); return MH_E_INVALID; }
    MH_HIP_TRY(hipSetDevice(device_id));
    mh_ctx* c = new mh_ctx();
    c->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return hip_fail(e, "hipStreamCreate"); }
    e = hipMalloc((void**) &c->status, 2 * sizeof(int32_t));
    if (e != hipSuccess) { hipStreamDestroy(c->stream); delete c; return hip_fail(e, "hipMalloc(status)"); }
    hipMemsetAsync(c->status, 0, 2 * sizeof(int32_t), c->stream);
    e = hipMalloc((void**) &c->planar_flag, sizeof(int32_t));
    if (e != hipSuccess) { hipFree(c->status); hipStreamDestroy(c->stream); delete c; return hip_fail(e, "hipMalloc(planar flag)"); }
    *out = c;
    return MH_OK;
}

static void release_fields(mh_ctx* c)
{
    for (auto& f : c->field) { if (f) hipFree(f); f = nullptr; }
    if (c->third) hipFree(c->third);
    c->third = nullptr;
    if (c->geom) hipFree(c->geom);
    c->geom = nullptr;
    if (c->inflow) hipFree(c->inflow);
    c->inflow = nullptr;
    if (c->staging) hipFree(c->staging);
    c->staging = nullptr;
    c->staging_doubles = 0;
}

void mh_destroy(mh_ctx* c)
{
    if (! c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (auto& ev : c->events) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
    release_fields(c);
    if (c->status) hipFree(c->status);
    if (c->planar_flag) hipFree(c->planar_flag);
    hipStreamDestroy(c->stream);
    delete c;
}

int mh_euler_cart_configure(mh_ctx* c, const mh_euler_cart_desc* d, int rk_order)
{
    if (! c) return MH_E_INVALID;
    if (int rc = check_desc(d)) return ctx_fail(c, rc);
    if (rk_order != 1 && rk_order != 2) { set_error("rk_order must be 1 or 2"); return ctx_fail(c, MH_E_INVALID); }
    if (d->fuse_stages > 0 && ! (rk_order == 2 && euler2d_fused_rk2_available(d)))
    {
        set_error("fuse_stages is required, but a fused RK2 step needs MH_ARITH_FAST, PLM, rk_order 2, rank 2 and physical (outflow / periodic) sides");
        return ctx_fail(c, MH_E_INVALID);
    }
    MH_HIP_TRY(hipSetDevice(c->device));
    release_fields(c);
    c->desc = *d;
    c->rk_order = rk_order;
    c->field_doubles = mh_euler_cart_field_doubles(d);
    for (auto& f : c->field)
    {
        hipError_t e = hipMalloc((void**) &f, c->field_doubles * sizeof(double));
        if (e != hipSuccess) { release_fields(c); set_error("hipMalloc of %zu bytes failed", c->field_doubles * sizeof(double)); return ctx_fail(c, MH_E_NOMEM); }
        hipMemsetAsync(f, 0, c->field_doubles * sizeof(double), c->stream);
    }
    c->kind = mh_ctx::KIND_EULER_CART;
    c->configured = true;
    c->uploaded = false;
    return MH_OK;
}

int mh_sedov_configure(mh_ctx* c, const mh_sedov_desc* d, const double* vertices_host)
{
    if (! c) return MH_E_INVALID;
    if (! d || ! vertices_host || d->nz < 2) { set_error("sedov: need a descriptor, vertices and nz >= 2"); return ctx_fail(c, MH_E_INVALID); }
    if (d->system != MH_SYSTEM_EULER && d->system != MH_SYSTEM_SRHD) { set_error("sedov: system must be MH_SYSTEM_EULER or MH_SYSTEM_SRHD"); return ctx_fail(c, MH_E_INVALID); }
    if (d->arith != MH_ARITH_STRICT) { set_error("sedov: only MH_ARITH_STRICT is built"); return ctx_fail(c, MH_E_INVALID); }
    MH_HIP_TRY(hipSetDevice(c->device));
    release_fields(c);
    const int n = d->nz;
    // geometry exactly as the reference builds it, on the host (src/subprog_sedov.cpp:166-181, :408)
    std::vector<double> geom((size_t) 3 * n + 1);
    double* dv = geom.data();
    double* da = dv + n;
    double* rc = da + n + 1;
    const double* v = vertices_host;
    for (int i = 0; i < n; ++i)
    {
        dv[i] = (std::pow(v[i + 1], 3) - std::pow(v[i], 3)) / 3;
        rc[i] = (v[i] + v[i + 1]) * 0.5;
    }
    for (int i = 0; i <= n; ++i) da[i] = v[i] * v[i];
    if (hipMalloc((void**) &c->geom, geom.size() * sizeof(double)) != hipSuccess) { set_error("hipMalloc(geom) failed"); return ctx_fail(c, MH_E_NOMEM); }
    MH_HIP_TRY(hipMemcpy(c->geom, geom.data(), geom.size() * sizeof(double), hipMemcpyHostToDevice));
    c->field_doubles = (size_t) 5 * n;
    for (auto& f : c->field)
        if (hipMalloc((void**) &f, c->field_doubles * sizeof(double)) != hipSuccess) { release_fields(c); set_error("hipMalloc(field) failed"); return ctx_fail(c, MH_E_NOMEM); }
    c->sedov = *d;
    c->kind = mh_ctx::KIND_SEDOV;
    c->configured = true;
    c->uploaded = false;
    return MH_OK;
}

static int check_cloud(const mh_cloud_desc* d)
{
    if (! d) { set_error("null descriptor"); return MH_E_INVALID; }
    if (d->nr < 2 || d->nq < 3) { set_error("cloud: need nr >= 2 and nq >= 3"); return MH_E_INVALID; }
    if (d->row_offset < 0 || d->row_offset + d->nr > d->nr_global) { set_error("cloud: slab [%d,%d) outside the global grid of %d rows", d->row_offset, d->row_offset + d->nr, d->nr_global); return MH_E_INVALID; }
    if (d->arith != MH_ARITH_STRICT && d->arith != MH_ARITH_FAST) { set_error("cloud: unknown arith mode %d", d->arith); return MH_E_INVALID; }
    if (d->bc_lo0 != MH_BC_INFLOW && d->bc_lo0 != MH_BC_EXTERNAL) { set_error("cloud: bc_lo0 must be inflow or external"); return MH_E_INVALID; }
    if (d->bc_hi0 != MH_BC_OUTFLOW && d->bc_hi0 != MH_BC_EXTERNAL) { set_error("cloud: bc_hi0 must be outflow or external"); return MH_E_INVALID; }
    if ((d->bc_lo0 == MH_BC_INFLOW) != (d->row_offset == 0) || (d->bc_hi0 == MH_BC_OUTFLOW) != (d->row_offset + d->nr == d->nr_global)) { set_error("cloud: physical boundary flags do not match the slab position"); return MH_E_INVALID; }
    if (!(d->gamma > 1.0)) { set_error("gamma must be > 1"); return MH_E_INVALID; }
    return MH_OK;
}

size_t mh_cloud_geometry_doubles(const mh_cloud_desc* d)
{
    // rv | dmu | sinq | cotq | per-row factors [nr_global][8] | per-column factors [nq][8]   (the last two: MH_ARITH_FAST only)
    return d ? (size_t) (d->nr_global + 1) + d->nq + (d->nq + 1) + d->nq + (size_t) 8 * d->nr_global + (size_t) 8 * d->nq : 0;
}

int mh_cloud_pack_geometry(const mh_cloud_desc* d, const double* rv, const double* qv, double* out)
{
    if (! d || ! rv || ! qv || ! out) { set_error("cloud geometry: null argument"); return MH_E_INVALID; }
    double* prv = out;
    double* dmu = prv + d->nr_global + 1;
    double* sinq = dmu + d->nq;
    double* cotq = sinq + d->nq + 1;
    for (int i = 0; i <= d->nr_global; ++i) prv[i] = rv[i];
    for (int j = 0; j < d->nq; ++j)
    {
        dmu[j] = -std::cos(qv[j + 1]) - -std::cos(qv[j]);
        cotq[j] = std::tan(M_PI_2 - (qv[j] + qv[j + 1]) * 0.5);
    }
    for (int j = 0; j <= d->nq; ++j) sinq[j] = std::sin((qv[j] + qv[j]) * 0.5);
    // MH_ARITH_FAST: the products the strict kernel forms per cell and stage in the reference's order (src/subprog_cloud.cpp:260-290)
    // factorise into a per-row and a per-column part; the fast kernel multiplies the two (not bit-exact, far inside its tolerance):
    //   dv = d3 (dmu 2 pi / 3), 1 / dv, -dAr = -(r_i r_i) (dmu 2 pi), -dAq = -(r_c dr) (sin q_j 2 pi)
    double* rowf = cotq + d->nq;
    double* colf = rowf + (size_t) 8 * d->nr_global;
    for (int i = 0; i < d->nr_global; ++i)
    {
        const double r0 = rv[i], r1 = rv[i + 1], d3 = r1 * r1 * r1 - r0 * r0 * r0, rc = (r0 + r1) * 0.5;
        const double row[8] = {r0 * r0, r1 * r1, d3, 1.0 / d3, rc * (r1 - r0), rc, 1.0 / rc, 0.0};
        for (int k = 0; k < 8; ++k) rowf[(size_t) 8 * i + k] = row[k];
    }
    for (int j = 0; j < d->nq; ++j)
    {
        const double col[8] = {dmu[j] * 2 * M_PI, dmu[j] * 2 * M_PI / 3.0, 1.0 / (dmu[j] * 2 * M_PI / 3.0), sinq[j] * 2 * M_PI, sinq[j + 1] * 2 * M_PI, cotq[j], 0.0, 0.0};
        for (int k = 0; k < 8; ++k) colf[(size_t) 8 * j + k] = col[k];
    }
    return MH_OK;
}

int mh_cloud_stage(const mh_cloud_desc* d, const double* geom_dev, const double* inflow_dev, const double* u_in,
                   const double* u_base, double* u_out, double dt, double stage_weight, int row_begin, int row_end,
                   int32_t* status, void* stream)
{
    if (int rc = check_cloud(d)) return rc;
    if (! geom_dev || ! u_in || ! u_out || u_in == u_out) { set_error("cloud stage: bad field pointers"); return MH_E_INVALID; }
    if (d->bc_lo0 == MH_BC_INFLOW && ! inflow_dev) { set_error("cloud stage: inflow row missing"); return MH_E_INVALID; }
    if (stage_weight != 1.0 && ! u_base) { set_error("cloud stage: combine needs u_base"); return MH_E_INVALID; }
    if (row_begin < 0 || row_end > d->nr || row_begin > row_end) { set_error("cloud stage: bad row range"); return MH_E_INVALID; }
    MH_HIP_TRY(cloud_stage_launch(d, geom_dev, inflow_dev, u_in, u_base, u_out, dt, stage_weight, row_begin, row_end, status, (hipStream_t) stream));
    return MH_OK;
}

int mh_cloud_configure(mh_ctx* c, const mh_cloud_desc* d, const double* rv, const double* qv, int rk_order)
{
    if (! c) return MH_E_INVALID;
    if (int rc = check_cloud(d)) return ctx_fail(c, rc);
    if (d->bc_lo0 != MH_BC_INFLOW || d->bc_hi0 != MH_BC_OUTFLOW) { set_error("cloud context: single-device form needs both radial sides physical"); return ctx_fail(c, MH_E_INVALID); }
    if (rk_order != 1 && rk_order != 2) { set_error("rk_order must be 1 or 2"); return ctx_fail(c, MH_E_INVALID); }
    if (d->fuse_stages > 0 && ! (rk_order == 2 && cloud_fused_rk2_available(d)))
    {
        set_error("cloud: fuse_stages is required, but a fused RK2 step needs MH_ARITH_FAST, PLM, rk_order 2 and both radial sides physical");
        return ctx_fail(c, MH_E_INVALID);
    }
    MH_HIP_TRY(hipSetDevice(c->device));
    release_fields(c);
    std::vector<double> geom(mh_cloud_geometry_doubles(d));
    if (int rc = mh_cloud_pack_geometry(d, rv, qv, geom.data())) return ctx_fail(c, rc);
    if (hipMalloc((void**) &c->geom, geom.size() * sizeof(double)) != hipSuccess) { set_error("hipMalloc(geom) failed"); return ctx_fail(c, MH_E_NOMEM); }
    MH_HIP_TRY(hipMemcpy(c->geom, geom.data(), geom.size() * sizeof(double), hipMemcpyHostToDevice));
    if (hipMalloc((void**) &c->inflow, (size_t) 5 * d->nq * sizeof(double)) != hipSuccess) { release_fields(c); set_error("hipMalloc(inflow) failed"); return ctx_fail(c, MH_E_NOMEM); }
    MH_HIP_TRY(hipMemset(c->inflow, 0, (size_t) 5 * d->nq * sizeof(double)));
    c->field_doubles = (size_t) 5 * (d->nr + 4) * d->nq;
    for (auto& f : c->field)
    {
        if (hipMalloc((void**) &f, c->field_doubles * sizeof(double)) != hipSuccess) { release_fields(c); set_error("hipMalloc(field) failed"); return ctx_fail(c, MH_E_NOMEM); }
        MH_HIP_TRY(hipMemset(f, 0, c->field_doubles * sizeof(double)));
    }
    c->cloud = *d;
    c->rk_order = rk_order;
    c->kind = mh_ctx::KIND_CLOUD;
    c->configured = true;
    c->uploaded = false;
    return MH_OK;
}

int mh_sedov_diagnostics(mh_ctx* c, double* fields_host, int32_t indices_host[3])
{
    if (! c || c->kind != mh_ctx::KIND_SEDOV || ! c->uploaded) { set_error("sedov diagnostics: needs a sedov context holding a solution"); return ctx_fail(c, MH_E_STATE); }
    MH_HIP_TRY(hipSetDevice(c->device));
    const int n = c->sedov.nz;
    double* fields = c->field[1];                      // the stage buffer (5 n doubles) holds nothing between steps
    int32_t* indices = nullptr;
    if (hipMalloc((void**) &indices, 3 * sizeof(int32_t)) != hipSuccess) { set_error("sedov diagnostics: hipMalloc failed"); return ctx_fail(c, MH_E_NOMEM); }
    hipError_t e = sedov_diagnostics_launch(c->sedov.system, c->field[0], c->geom, n, c->sedov.gamma, fields, indices, c->status, c->stream);
    if (e == hipSuccess && fields_host) e = hipMemcpyAsync(fields_host, fields, (size_t) 4 * n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && indices_host) e = hipMemcpyAsync(indices_host, indices, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void) hipFree(indices);
    if (e != hipSuccess) return hip_fail(e, "sedov diagnostics");
    return MH_OK;
}

int mh_cloud_diagnostics(mh_ctx* c, const double units[3], double* fields_host, double* columns_host)
{
    if (! c || c->kind != mh_ctx::KIND_CLOUD || ! c->uploaded || ! units) { set_error("cloud diagnostics: needs a cloud context holding a solution"); return ctx_fail(c, MH_E_STATE); }
    if (! (units[0] > 0.0) || ! (units[1] > 0.0) || ! (units[2] > 0.0)) { set_error("cloud diagnostics: reference units must be positive"); return ctx_fail(c, MH_E_INVALID); }
    MH_HIP_TRY(hipSetDevice(c->device));
    const size_t ncell = (size_t) c->cloud.nr * c->cloud.nq, nq = (size_t) c->cloud.nq;
    double* fields = c->field[1];                      // the stage scratch holds nothing between steps: 5 (nr + 4) nq doubles
    double* work = nullptr;
    if (hipMalloc((void**) &work, (4 * ncell + 15 * nq) * sizeof(double)) != hipSuccess) { set_error("cloud diagnostics: hipMalloc failed"); return ctx_fail(c, MH_E_NOMEM); }
    double* columns = work + 4 * ncell;
    hipError_t e = cloud_diagnostics_launch(&c->cloud, c->geom, c->field[0], units, fields, work, columns, c->status, c->stream);
    if (e == hipSuccess && fields_host) e = hipMemcpyAsync(fields_host, fields, 5 * ncell * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && columns_host) e = hipMemcpyAsync(columns_host, columns, 15 * nq * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void) hipFree(work);
    if (e != hipSuccess) return hip_fail(e, "cloud diagnostics");
    return MH_OK;
}

int mh_cloud_set_inflow(mh_ctx* c, const double* inflow_aos)
{
    if (! c || c->kind != mh_ctx::KIND_CLOUD || ! inflow_aos) { set_error("set_inflow: not a cloud context"); return ctx_fail(c, MH_E_STATE); }
    MH_HIP_TRY(hipSetDevice(c->device));
    const size_t nq = (size_t) c->cloud.nq;
    std::vector<double> soa(5 * nq);
    for (size_t j = 0; j < nq; ++j) for (int q = 0; q < 5; ++q) soa[q * nq + j] = inflow_aos[5 * j + q];
    // planarity (mh_cloud_desc.planar): a nozzle row with an azimuthal velocity puts azimuthal momentum into the field - general kernel from here on
    // (STRICT: the bit pattern of +0.0, as for the field - srhd_device.hpp)
    c->inflow_planar = true;
    for (size_t j = 0; j < nq; ++j) c->inflow_planar = c->inflow_planar && inflow_aos[5 * j + 3] == 0.0 && ! (c->cloud.arith == MH_ARITH_STRICT && std::signbit(inflow_aos[5 * j + 3]));
    if (! c->inflow_planar)
    {
        if (c->cloud.planar > 0) { set_error("set_inflow: `planar` was asserted, but the nozzle row has an azimuthal velocity"); return ctx_fail(c, MH_E_INVALID); }
        c->planar_now = false;
    }
    // stream-ordered after the stages already queued; the staging vector is consumed before return
    MH_HIP_TRY(hipStreamSynchronize(c->stream));
    MH_HIP_TRY(hipMemcpy(c->inflow, soa.data(), soa.size() * sizeof(double), hipMemcpyHostToDevice));
    return MH_OK;
}

// sedov fields are tiny: transpose AoS [nz][5] <-> SoA [5][nz] on the host
static int sedov_transfer(mh_ctx* c, double* host_aos, size_t ncell, bool to_device)
{
    const size_t n = (size_t) c->sedov.nz;
    if (ncell != n || ! host_aos) { set_error("sedov: expected %zu cells, got %zu", n, ncell); return MH_E_INVALID; }
    std::vector<double> soa(5 * n);
    if (to_device)
    {
        for (size_t i = 0; i < n; ++i) for (int q = 0; q < 5; ++q) soa[q * n + i] = host_aos[5 * i + q];
        MH_HIP_TRY(hipMemcpy(c->field[0], soa.data(), soa.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    else
    {
        MH_HIP_TRY(hipStreamSynchronize(c->stream));
        MH_HIP_TRY(hipMemcpy(soa.data(), c->field[0], soa.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) for (int q = 0; q < 5; ++q) host_aos[5 * i + q] = soa[q * n + i];
    }
    return MH_OK;
}

static int ensure_staging(mh_ctx* c, size_t doubles)
{
    if (c->staging_doubles >= doubles) return MH_OK;
    if (c->staging) hipFree(c->staging);
    c->staging = nullptr;
    c->staging_doubles = 0;
    if (hipMalloc((void**) &c->staging, doubles * sizeof(double)) != hipSuccess) { set_error("hipMalloc(staging) failed"); return MH_E_NOMEM; }
    c->staging_doubles = doubles;
    return MH_OK;
}

static bool ctx_can_fuse(const mh_ctx* c);

int mh_upload(mh_ctx* c, const double* u_aos_host, size_t ncell)
{
    if (! c || ! c->configured) { set_error("upload before configure"); return ctx_fail(c, MH_E_STATE); }
    if (c->kind == mh_ctx::KIND_SEDOV)
    {
        MH_HIP_TRY(hipSetDevice(c->device));
        if (int rc = sedov_transfer(c, const_cast<double*>(u_aos_host), ncell, true)) return ctx_fail(c, rc);
        c->uploaded = true;
        return MH_OK;
    }
    const mh_euler_cart_desc* d = &c->desc;
    const bool cloud = c->kind == mh_ctx::KIND_CLOUD;
    const int n0 = cloud ? c->cloud.nr : d->n[0];
    const size_t pitch = cloud ? (size_t) c->cloud.nq : row_pitch_of(d);
    const size_t expect = (size_t) n0 * pitch;
    if (ncell != expect || ! u_aos_host) { set_error("upload: expected %zu cells, got %zu", expect, ncell); return ctx_fail(c, MH_E_INVALID); }
    MH_HIP_TRY(hipSetDevice(c->device));
    if (int rc = ensure_staging(c, ncell * 5)) return ctx_fail(c, rc);
    MH_HIP_TRY(hipMemcpyAsync(c->staging, u_aos_host, ncell * 5 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    MH_HIP_TRY(aos_to_soa_launch(c->staging, c->field[0], 5, n0, pitch, c->stream));
    if (! cloud) MH_HIP_TRY(fill_ghost_rows_launch(c->field[0], 5, n0, pitch, d->bc_lo0, d->bc_hi0, c->stream));
    // planarity (mh_euler_cart_desc.planar): does this 2-D field carry a third momentum at all? One pass here, none per step - the planar
    // kernel writes that component as zero, so the property holds for as long as this solution is stepped
    c->planar_now = false;
    const int planar_request = cloud ? c->cloud.planar : d->planar;
    // (every PLM stage kernel of the 2-D Euler and `cloud` steppers has a planar form, STRICT on the exact bit pattern of +0.0)
    // A side of kind MH_BC_EXTERNAL holds ghost rows this context never sees (the caller's halo exchange writes them through mh_field_ptr), so
    // the look at rows [0, n0) decides nothing there: with such a side only planar > 0 - the caller's word for the WHOLE grid, as for slabs
    // that exchange with other processes - selects the planar kernels; planar = 0 means the general ones.
    const bool external_side = (cloud ? c->cloud.bc_lo0 : d->bc_lo0) == MH_BC_EXTERNAL || (cloud ? c->cloud.bc_hi0 : d->bc_hi0) == MH_BC_EXTERNAL;
    if ((cloud ? c->cloud.plm_theta >= 0.0 : (d->rank == 2 && d->plm_theta >= 0.0)) && planar_request >= 0 && ! (external_side && planar_request == 0))
    {
        int32_t nonzero = 0;
        MH_HIP_TRY(hipMemsetAsync(c->planar_flag, 0, sizeof(int32_t), c->stream));
        MH_HIP_TRY(plane_nonzero_launch(c->field[0], 5, 3, n0, pitch, c->planar_flag, c->stream, (cloud ? c->cloud.arith : d->arith) == MH_ARITH_STRICT));
        MH_HIP_TRY(hipMemcpyAsync(&nonzero, c->planar_flag, sizeof nonzero, hipMemcpyDeviceToHost, c->stream));
        MH_HIP_TRY(hipStreamSynchronize(c->stream));
        c->planar_now = nonzero == 0;
        if (planar_request > 0 && ! c->planar_now)
        {
            set_error(cloud ? "upload: `planar` was asserted, but the field has an azimuthal momentum" : "upload: `planar` was asserted, but the field has a third momentum");
            return ctx_fail(c, MH_E_INVALID);
        }
    }
    MH_HIP_TRY(hipStreamSynchronize(c->stream));
    c->uploaded = true;
    return MH_OK;
}

int mh_download(mh_ctx* c, double* u_aos_host, size_t ncell)
{
    if (! c || ! c->uploaded) { set_error("download before upload"); return ctx_fail(c, MH_E_STATE); }
    if (c->kind == mh_ctx::KIND_SEDOV)
    {
        MH_HIP_TRY(hipSetDevice(c->device));
        if (int rc = sedov_transfer(c, u_aos_host, ncell, false)) return ctx_fail(c, rc);
        return MH_OK;
    }
    const mh_euler_cart_desc* d = &c->desc;
    const bool cloud = c->kind == mh_ctx::KIND_CLOUD;
    const int n0 = cloud ? c->cloud.nr : d->n[0];
    const size_t pitch = cloud ? (size_t) c->cloud.nq : row_pitch_of(d);
    const size_t expect = (size_t) n0 * pitch;
    if (ncell != expect || ! u_aos_host) { set_error("download: expected %zu cells, got %zu", expect, ncell); return ctx_fail(c, MH_E_INVALID); }
    MH_HIP_TRY(hipSetDevice(c->device));
    if (int rc = ensure_staging(c, ncell * 5)) return ctx_fail(c, rc);
    MH_HIP_TRY(soa_to_aos_launch(c->field[0], c->staging, 5, n0, pitch, c->stream));
    MH_HIP_TRY(hipMemcpyAsync(u_aos_host, c->staging, ncell * 5 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    MH_HIP_TRY(hipStreamSynchronize(c->stream));
    return MH_OK;
}

// the stage launches of one API call between two events (mh_profile_read divides by the launches)
struct ProfileSpan
{
    mh_ctx* c;
    std::pair<hipEvent_t, hipEvent_t> ev = {nullptr, nullptr};
    explicit ProfileSpan(mh_ctx* ctx) : c(ctx)
    {
        if (! c->profile) return;
        c->span_launches = 0;
        if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess) { ev = {nullptr, nullptr}; return; }
        hipEventRecord(ev.first, c->stream);
    }
    ~ProfileSpan()
    {
        if (! ev.first) return;
        hipEventRecord(ev.second, c->stream);
        if (c->span_launches > 0) { c->events.push_back(ev); c->event_launches.push_back(c->span_launches); }
        else { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
    }
};

static hipError_t timed_stage(mh_ctx* c, const double* in, const double* base, double* out, double dt, double w)
{
    if (c->profile) ++c->span_launches;
    if (c->kind == mh_ctx::KIND_CLOUD)
    {
        mh_cloud_desc d = c->cloud;
        d.planar = c->planar_now && c->inflow_planar ? 1 : -1;          // what mh_upload and mh_cloud_set_inflow found
        return cloud_stage_launch(&d, c->geom, c->inflow, in, base, out, dt, w, 0, c->cloud.nr, c->status, c->stream);
    }
    mh_euler_cart_desc d = c->desc;
    d.planar = c->planar_now ? 1 : -1;          // what the upload found (mh_upload)
    return cart_stage_launch(&d, in, base, out, dt, w, 0, c->desc.n[0], c->status, c->stream);
}

// a whole-field RK2 step as one launch (the descriptors' fuse_stages; euler2d_fused.hip, cloud_fused.hip)
static bool ctx_can_fuse(const mh_ctx* c)
{
    if (c->kind == mh_ctx::KIND_CLOUD) return c->rk_order == 2 && c->cloud.fuse_stages >= 0 && cloud_fused_rk2_available(&c->cloud);
    return c->kind == mh_ctx::KIND_EULER_CART && c->rk_order == 2 && c->desc.fuse_stages >= 0 && euler2d_fused_rk2_available(&c->desc);
}

static hipError_t timed_fused_step(mh_ctx* c, const double* in, double* out, double dt)
{
    if (c->profile) ++c->span_launches;
    if (c->kind == mh_ctx::KIND_CLOUD)
    {
        mh_cloud_desc d = c->cloud;
        d.planar = c->planar_now && c->inflow_planar ? 1 : -1;          // what mh_upload and mh_cloud_set_inflow found
        return cloud_fused_rk2_launch(&d, c->geom, c->inflow, in, out, dt, c->status, c->stream);
    }
    mh_euler_cart_desc d = c->desc;
    d.planar = c->planar_now ? 1 : -1;          // what the upload found (mh_upload)
    return euler2d_fused_rk2_launch(&d, in, out, dt, c->status, c->stream, LaunchEvents());
}

int mh_step(mh_ctx* c, double dt, int nsteps)
{
    if (! c || ! c->uploaded) { set_error("step before upload"); return ctx_fail(c, MH_E_STATE); }
    MH_HIP_TRY(hipSetDevice(c->device));
    if (c->kind == mh_ctx::KIND_SEDOV)
    {
        // forward Euler only (src/subprog_sedov.cpp:414)
        const int n = c->sedov.nz;
        const double* dv = c->geom;
        const double* da = dv + n;
        const double* rc = da + n + 1;
        for (int s = 0; s < nsteps; ++s)
        {
            MH_HIP_TRY(sedov_stage_launch(c->sedov.system, c->field[0], c->field[1], dv, da, rc, n, c->sedov.gamma, dt, c->status, c->stream));
            std::swap(c->field[0], c->field[1]);
        }
        return MH_OK;
    }
    const ProfileSpan span(c);
    if (c->kind == mh_ctx::KIND_CLOUD)
    {
        const bool fused_cloud = ctx_can_fuse(c);
        for (int s = 0; s < nsteps; ++s)
        {
            if (c->rk_order == 1)
            {
                MH_HIP_TRY(timed_stage(c, c->field[0], nullptr, c->field[1], dt, 1.0));
                std::swap(c->field[0], c->field[1]);
            }
            else if (fused_cloud)
            {
                MH_HIP_TRY(timed_fused_step(c, c->field[0], c->field[1], dt));
                std::swap(c->field[0], c->field[1]);
            }
            else
            {
                MH_HIP_TRY(timed_stage(c, c->field[0], nullptr, c->field[1], dt, 1.0));
                MH_HIP_TRY(timed_stage(c, c->field[1], c->field[0], c->field[0], dt, 0.5));
            }
        }
        return MH_OK;
    }
    const bool fused = ctx_can_fuse(c);
    for (int s = 0; s < nsteps; ++s)
    {
        if (c->rk_order == 1)
        {
            MH_HIP_TRY(timed_stage(c, c->field[0], nullptr, c->field[1], dt, 1.0));
            std::swap(c->field[0], c->field[1]);
        }
        else if (fused)
        {
            MH_HIP_TRY(timed_fused_step(c, c->field[0], c->field[1], dt));
            std::swap(c->field[0], c->field[1]);
        }
        else
        {
            // u1 = advance(u0); u = u0*0.5 + advance(u1)*0.5 written in place over u0
            MH_HIP_TRY(timed_stage(c, c->field[0], nullptr, c->field[1], dt, 1.0));
            MH_HIP_TRY(timed_stage(c, c->field[1], c->field[0], c->field[0], dt, 0.5));
        }
    }
    return MH_OK;
}

int mh_synchronize(mh_ctx* c)
{
    if (! c) return MH_E_INVALID;
    MH_HIP_TRY(hipSetDevice(c->device));
    MH_HIP_TRY(hipStreamSynchronize(c->stream));
    return MH_OK;
}

int mh_status(mh_ctx* c, mh_step_result* result)
{
    if (! c || ! result) return MH_E_INVALID;
    MH_HIP_TRY(hipSetDevice(c->device));
    uint32_t h[2] = {0, 0};
    MH_HIP_TRY(hipMemcpyAsync(h, c->status, sizeof h, hipMemcpyDeviceToHost, c->stream));
    MH_HIP_TRY(hipMemsetAsync(c->status, 0, sizeof h, c->stream));
    MH_HIP_TRY(hipStreamSynchronize(c->stream));
    result->status = (int32_t) h[0];
    result->reserved = 0;
    result->first_bad_index = h[1] ? (uint64_t) (0xFFFFFFFFu - h[1]) : UINT64_MAX;      // device word: status_device.hpp
    return MH_OK;
}

int mh_status_word(mh_ctx* c, int32_t* status)
{
    if (! status) return MH_E_INVALID;
    mh_step_result r;
    if (int rc = mh_status(c, &r)) return rc;
    *status = r.status;
    return MH_OK;
}

int mh_step_checked(mh_ctx* c, double dt, mh_step_result* result)
{
    if (! c || ! c->uploaded) { set_error("step before upload"); return ctx_fail(c, MH_E_STATE); }
    if (! result) return MH_E_INVALID;
    MH_HIP_TRY(hipSetDevice(c->device));
    MH_HIP_TRY(hipMemsetAsync(c->status, 0, 2 * sizeof(int32_t), c->stream));       // the result speaks about THIS step only
    double* committed = nullptr;              // the buffer that holds the step's result
    {
    const ProfileSpan span(c);
    if (c->kind == mh_ctx::KIND_SEDOV)
    {
        const int n = c->sedov.nz;
        const double* dv = c->geom;
        const double* da = dv + n;
        const double* rc = da + n + 1;
        MH_HIP_TRY(sedov_stage_launch(c->sedov.system, c->field[0], c->field[1], dv, da, rc, n, c->sedov.gamma, dt, c->status, c->stream));
        committed = c->field[1];
    }
    else if (c->rk_order == 1)
    {
        MH_HIP_TRY(timed_stage(c, c->field[0], nullptr, c->field[1], dt, 1.0));
        committed = c->field[1];
    }
    else if (ctx_can_fuse(c))
    {
        MH_HIP_TRY(timed_fused_step(c, c->field[0], c->field[1], dt));       // never in place: field[0] survives a failed step
        committed = c->field[1];
    }
    else
    {
        if (! c->third)
        {
            if (hipMalloc((void**) &c->third, c->field_doubles * sizeof(double)) != hipSuccess) { set_error("hipMalloc of the third field (%zu bytes) failed", c->field_doubles * sizeof(double)); return ctx_fail(c, MH_E_NOMEM); }
            MH_HIP_TRY(hipMemsetAsync(c->third, 0, c->field_doubles * sizeof(double), c->stream));
        }
        MH_HIP_TRY(timed_stage(c, c->field[0], nullptr, c->field[1], dt, 1.0));
        MH_HIP_TRY(timed_stage(c, c->field[1], c->field[0], c->third, dt, 0.5));       // NOT in place: field[0] survives a failed step
        committed = c->third;
    }
    }
    if (int rc = mh_status(c, result)) return ctx_fail(c, rc);
    if (result->status != 0)
    {
        set_error("step rejected: status 0x%x, first failing cell %llu; the previous solution is unchanged", result->status, (unsigned long long) result->first_bad_index);
        return ctx_fail(c, MH_E_PHYSICS);
    }
    if (committed == c->third) std::swap(c->field[0], c->third);
    else                       std::swap(c->field[0], c->field[1]);
    return MH_OK;
}

int mh_field_is_planar(const mh_ctx* c) { return c && c->planar_now && (c->kind != mh_ctx::KIND_CLOUD || c->inflow_planar) ? 1 : 0; }

double* mh_field_ptr(mh_ctx* c, int which)
{
    if (! c || which < 0 || which > 1) return nullptr;
    return c->field[which];
}

int mh_profile_enable(mh_ctx* c, int on)
{
    if (! c) return MH_E_INVALID;
    for (auto& ev : c->events) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
    c->events.clear();
    c->event_launches.clear();
    c->profile = on != 0;
    return MH_OK;
}

int mh_profile_read(mh_ctx* c, double* avg_stage_ms, int* nlaunches)
{
    if (! c) return MH_E_INVALID;
    MH_HIP_TRY(hipSetDevice(c->device));
    MH_HIP_TRY(hipStreamSynchronize(c->stream));
    double total = 0.0;
    for (auto& ev : c->events)
    {
        float ms = 0.f;
        MH_HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
        total += ms;
    }
    int count = 0;
    for (int k : c->event_launches) count += k;
    if (avg_stage_ms) *avg_stage_ms = count == 0 ? 0.0 : total / count;
    if (nlaunches) *nlaunches = count;
    return MH_OK;
}

// ---- per-function entry points --------------------------------------------
static dim3 grid1(size_t n) { return dim3((unsigned) ((n + 255) / 256)); }

int mh_plm_gradient_n(size_t n, const double* yl, const double* y0, const double* yr, double theta, double* g, int arith, void* stream)
{
    if (arith != MH_ARITH_STRICT && arith != MH_ARITH_FAST) { set_error("unknown arith mode"); return MH_E_INVALID; }
    if (n == 0) return MH_OK;
    if (arith == MH_ARITH_FAST) hipLaunchKernelGGL(plm_kernel<FastArith>, grid1(n), dim3(256), 0, (hipStream_t) stream, n, yl, y0, yr, theta, g);
    else                        hipLaunchKernelGGL(plm_kernel<StrictArith>, grid1(n), dim3(256), 0, (hipStream_t) stream, n, yl, y0, yr, theta, g);
    MH_HIP_TRY(hipGetLastError());
    return MH_OK;
}

int mh_euler_recover_primitive_n(size_t n, const double* U, double gamma, double tfloor, double* P, int arith, void* stream)
{
    if (arith != MH_ARITH_STRICT && arith != MH_ARITH_FAST) { set_error("unknown arith mode"); return MH_E_INVALID; }
    if (n == 0) return MH_OK;
    if (arith == MH_ARITH_FAST) hipLaunchKernelGGL(c2p_kernel<FastArith>, grid1(n), dim3(256), 0, (hipStream_t) stream, n, U, gamma, tfloor, P);
    else                        hipLaunchKernelGGL(c2p_kernel<StrictArith>, grid1(n), dim3(256), 0, (hipStream_t) stream, n, U, gamma, tfloor, P);
    MH_HIP_TRY(hipGetLastError());
    return MH_OK;
}

int mh_euler_to_conserved_n(size_t n, const double* P, double gamma, double* U, int arith, void* stream)
{
    if (arith != MH_ARITH_STRICT && arith != MH_ARITH_FAST) { set_error("unknown arith mode"); return MH_E_INVALID; }
    if (n == 0) return MH_OK;
    if (arith == MH_ARITH_FAST) hipLaunchKernelGGL(p2c_kernel<FastArith>, grid1(n), dim3(256), 0, (hipStream_t) stream, n, P, gamma, U);
    else                        hipLaunchKernelGGL(p2c_kernel<StrictArith>, grid1(n), dim3(256), 0, (hipStream_t) stream, n, P, gamma, U);
    MH_HIP_TRY(hipGetLastError());
    return MH_OK;
}

int mh_euler_riemann_n(size_t n, const double* Pl, const double* Pr, int axis, double gamma, int riemann_kind, double* F, int arith, void* stream)
{
    if (arith != MH_ARITH_STRICT && arith != MH_ARITH_FAST) { set_error("unknown arith mode"); return MH_E_INVALID; }
    if (axis < 0 || axis > 2 || (riemann_kind != MH_RIEMANN_HLLE && riemann_kind != MH_RIEMANN_HLLC)) { set_error("bad axis/solver"); return MH_E_INVALID; }
    if (n == 0) return MH_OK;
    hipStream_t s = (hipStream_t) stream;
#define MH_LAUNCH_R(R, AX) do { if (arith == MH_ARITH_FAST) hipLaunchKernelGGL((riemann_kernel<FastArith, R, AX>), grid1(n), dim3(256), 0, s, n, Pl, Pr, gamma, F); \
                              else hipLaunchKernelGGL((riemann_kernel<StrictArith, R, AX>), grid1(n), dim3(256), 0, s, n, Pl, Pr, gamma, F); } while (0)
    switch (riemann_kind * 3 + axis)
    {
        case 0: MH_LAUNCH_R(0, 0); break;
        case 1: MH_LAUNCH_R(0, 1); break;
        case 2: MH_LAUNCH_R(0, 2); break;
        case 3: MH_LAUNCH_R(1, 0); break;
        case 4: MH_LAUNCH_R(1, 1); break;
        case 5: MH_LAUNCH_R(1, 2); break;
    }
#undef MH_LAUNCH_R
    MH_HIP_TRY(hipGetLastError());
    return MH_OK;
}

int mh_srhd_recover_primitive_n(size_t n, const double* U, double gamma, double tfloor, double* P, int32_t* status, void* stream)
{
    if (n == 0) return MH_OK;
    hipLaunchKernelGGL(srhd_c2p_kernel, grid1(n), dim3(256), 0, (hipStream_t) stream, n, U, gamma, tfloor, P, status);
    MH_HIP_TRY(hipGetLastError());
    return MH_OK;
}

int mh_srhd_to_conserved_n(size_t n, const double* P, double gamma, double* U, void* stream)
{
    if (n == 0) return MH_OK;
    hipLaunchKernelGGL(srhd_p2c_kernel, grid1(n), dim3(256), 0, (hipStream_t) stream, n, P, gamma, U);
    MH_HIP_TRY(hipGetLastError());
    return MH_OK;
}

int mh_srhd_riemann_hlle_n(size_t n, const double* Pl, const double* Pr, int axis, double gamma, double* F, void* stream)
{
    if (axis < 0 || axis > 2) { set_error("bad axis"); return MH_E_INVALID; }
    if (n == 0) return MH_OK;
    hipStream_t s = (hipStream_t) stream;
    if (axis == 0) hipLaunchKernelGGL(srhd_hlle_kernel<0>, grid1(n), dim3(256), 0, s, n, Pl, Pr, gamma, F);
    if (axis == 1) hipLaunchKernelGGL(srhd_hlle_kernel<1>, grid1(n), dim3(256), 0, s, n, Pl, Pr, gamma, F);
    if (axis == 2) hipLaunchKernelGGL(srhd_hlle_kernel<2>, grid1(n), dim3(256), 0, s, n, Pl, Pr, gamma, F);
    MH_HIP_TRY(hipGetLastError());
    return MH_OK;
}

int mh_srhd_source_terms_n(size_t n, const double* P, const double* r, const double* cot_theta, double gamma, double* S, void* stream)
{
    if (n == 0) return MH_OK;
    hipLaunchKernelGGL(srhd_src_kernel, grid1(n), dim3(256), 0, (hipStream_t) stream, n, P, r, cot_theta, gamma, S);
    MH_HIP_TRY(hipGetLastError());
    return MH_OK;
}

static int iso2d_launch(int mode, int axis, size_t n, const double* a, const double* b, const double* c, const double* d,
                        double* out, double* out2, int32_t* flag, void* stream)
{
    if (axis < 0 || axis > 1) { set_error("iso2d: axis must be 0 or 1"); return MH_E_INVALID; }
    if (n == 0) return MH_OK;
    if (axis == 0) hipLaunchKernelGGL(iso2d_kernel<0>, grid1(n), dim3(256), 0, (hipStream_t) stream, mode, n, a, b, c, d, out, out2, flag);
    else           hipLaunchKernelGGL(iso2d_kernel<1>, grid1(n), dim3(256), 0, (hipStream_t) stream, mode, n, a, b, c, d, out, out2, flag);
    MH_HIP_TRY(hipGetLastError());
    return MH_OK;
}

int mh_iso2d_to_conserved_n(size_t n, const double* P, double* U, void* stream)
{ return iso2d_launch(ISO_P2C, 0, n, P, nullptr, nullptr, nullptr, U, nullptr, nullptr, stream); }
int mh_iso2d_recover_primitive_n(size_t n, const double* U, double* P, int32_t* threw, void* stream)
{ return iso2d_launch(ISO_C2P, 0, n, U, nullptr, nullptr, nullptr, P, nullptr, threw, stream); }
int mh_iso2d_to_conserved_angmom_n(size_t n, const double* P, const double* x, double* Q, void* stream)
{ return iso2d_launch(ISO_P2Q, 0, n, P, x, nullptr, nullptr, Q, nullptr, nullptr, stream); }
int mh_iso2d_recover_primitive_angmom_n(size_t n, const double* Q, const double* x, double* P, int32_t* threw, void* stream)
{ return iso2d_launch(ISO_Q2P, 0, n, Q, x, nullptr, nullptr, P, nullptr, threw, stream); }
int mh_iso2d_flux_n(size_t n, const double* P, const double* cs2, int axis, double* F, void* stream)
{ return iso2d_launch(ISO_FLUX, axis, n, P, nullptr, cs2, nullptr, F, nullptr, nullptr, stream); }
int mh_iso2d_wavespeeds_n(size_t n, const double* P, const double* cs2, int axis, double* lam, void* stream)
{ return iso2d_launch(ISO_LAM, axis, n, P, nullptr, cs2, nullptr, lam, nullptr, nullptr, stream); }
int mh_iso2d_riemann_n(size_t n, const double* Pl, const double* Pr, const double* cs2l, const double* cs2r, int axis, int riemann_kind,
                       double* F, double* contact, int32_t* threw, void* stream)
{
    if (riemann_kind != MH_RIEMANN_HLLE && riemann_kind != MH_RIEMANN_HLLC) { set_error("iso2d: unknown riemann solver"); return MH_E_INVALID; }
    return iso2d_launch(riemann_kind == MH_RIEMANN_HLLC ? ISO_HLLC : ISO_HLLE, axis, n, Pl, Pr, cs2l, cs2r, F, contact, threw, stream);
}

// ---- integer work ----------------------------------------------------------
void mh_partition_rows(size_t count, size_t nparts, size_t part, size_t* start, size_t* final_)
{
    *start  = (part + 0) * count / nparts;
    *final_ = (part + 1) * count / nparts;
}

static void factorize(int num, std::vector<int>& out)
{
    // smallest-divisor-first recursion, the order of mara::parallel::detail::prime_factors
    int d = 2;
    for (; ; ++d)
    {
        if (num % d == 0) break;
        if (d * d > num) { d = num; break; }
    }
    if (d == num || num / d == 1) { out.push_back(d == num ? num : d); return; }
    factorize(d, out);
    factorize(num / d, out);
}

int mh_propose_block_decomposition(int rank, unsigned long nblocks, unsigned long* blocks_per_axis)
{
    if (rank < 1 || rank > 3 || nblocks < 1 || ! blocks_per_axis) return MH_E_INVALID;
    std::vector<int> f;
    factorize((int) nblocks, f);
    for (int g = 0; g < rank; ++g)
    {
        size_t a, b;
        mh_partition_rows(f.size(), (size_t) rank, (size_t) g, &a, &b);
        int prod = 1;
        for (size_t k = a; k < b; ++k) prod *= f[k];
        blocks_per_axis[g] = (unsigned long) prod;
    }
    return MH_OK;
}

// mara::create_access_pattern_array (src/app_parallel.hpp:148-179) for the blocks of propose_block_decomposition<3>(world): block
// (c0, c1, c2) = the row-major position `rank` in the array of access patterns; extents per axis by nd::divvy
int mh_block_layout(const int global_n[3], int world, int rank, int blocks_per_axis[3], int coords[3], int start[3], int count[3])
{
    if (! global_n || world < 1 || rank < 0 || rank >= world) { set_error("mh_block_layout: rank %d of %d", rank, world); return MH_E_INVALID; }
    unsigned long B[3];
    if (int rc = mh_propose_block_decomposition(3, (unsigned long) world, B)) return rc;
    const int c[3] = {rank / (int) (B[1] * B[2]), (rank / (int) B[2]) % (int) B[1], rank % (int) B[2]};
    for (int a = 0; a < 3; ++a)
    {
        size_t s0, s1;
        mh_partition_rows((size_t) global_n[a], (size_t) B[a], (size_t) c[a], &s0, &s1);
        if (s1 <= s0) { set_error("too many blocks for global domain size"); return MH_E_INVALID; }      // the reference's std::logic_error (:160-163)
        if (blocks_per_axis) blocks_per_axis[a] = (int) B[a];
        if (coords) coords[a] = c[a];
        if (start) start[a] = (int) s0;
        if (count) count[a] = (int) (s1 - s0);
    }
    return MH_OK;
}

// ---- device utilities -------------------------------------------------------
int mh_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }
int mh_device_cu_count(void) { return device_cu_count(); }          // compute units of the current device: what the launchers size their chunks for
int mh_malloc(void** ptr, size_t bytes) { MH_HIP_TRY(hipMalloc(ptr, bytes)); return MH_OK; }
int mh_free(void* ptr) { MH_HIP_TRY(hipFree(ptr)); return MH_OK; }
int mh_memcpy_h2d(void* dst, const void* src, size_t bytes) { MH_HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return MH_OK; }
int mh_memcpy_d2h(void* dst, const void* src, size_t bytes) { MH_HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return MH_OK; }
int mh_device_synchronize(void) { MH_HIP_TRY(hipDeviceSynchronize()); return MH_OK; }

// how the last one-launch RK2 step of this process cut its rows (family 1: 2-D Euler, 3: cloud - the row-range guard's numbering)
int mh_debug_last_fused_cut(int family, int32_t out[4])
{
    if (! out || (family != 1 && family != 3)) { set_error("mh_debug_last_fused_cut: family 1 (euler2d fused) or 3 (cloud fused)"); return MH_E_INVALID; }
    int v[4];
    if (family == 1) euler2d_fused_last_cut(v); else cloud_fused_last_cut(v);
    for (int k = 0; k < 4; ++k) out[k] = v[k];
    return MH_OK;
}

int mh_debug_row_range(int family, int32_t lo_hi[2], int reset)
{
    if (! lo_hi) { set_error("row range: null argument"); return MH_E_INVALID; }
    bool ok = false;
    switch (family)
    {
        case MH_ROWS_EULER2D:        ok = rows_requested_euler2d(lo_hi, reset); break;
        case MH_ROWS_EULER2D_FUSED:  ok = rows_requested_euler2d_fused(lo_hi, reset); break;
        case MH_ROWS_CLOUD:          ok = rows_requested_cloud(lo_hi, reset); break;
        case MH_ROWS_CLOUD_FUSED:    ok = rows_requested_cloud_fused(lo_hi, reset); break;
        case MH_ROWS_EULER3D_STRICT: ok = rows_requested_euler3d(lo_hi, reset); break;
        case MH_ROWS_EULER3D_FAST:   ok = rows_requested_euler3d_fast(lo_hi, reset); break;
        case MH_ROWS_BINARY_STRICT:  ok = rows_requested_binary(lo_hi, reset); break;
        case MH_ROWS_BINARY_FAST:    ok = rows_requested_binary_fast(lo_hi, reset); break;
        default: set_error("row range: unknown kernel family %d", family); return MH_E_INVALID;
    }
    if (! ok) { set_error("row range: this library was built without the row-range guard (-DMH_CHECK_ROWS; make -C mara3_amd/csrc check)"); return MH_E_STATE; }
    return MH_OK;
}

} // extern "C"

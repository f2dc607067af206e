// Shared helpers of the compiled hosts: wall-clock timing of a step (the reference's
// mara::time_execution, src/app_performance.hpp:76-82, which produces its `kzps`
// figure), a throwing wrapper around the C ABI, and a raw binary state dump (the
// HDF5 checkpoints of sedov and cloud live in h5_checkpoint.hpp; this raw dump is what the parity tests read).
#pragma once
#include <chrono>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>
#include <sys/stat.h>
#include "mara_hip.h"

namespace host {

inline void check(int rc, const mh_ctx* ctx, const char* what)
{
    if (rc != MH_OK) throw std::runtime_error(std::string(what) + ": " + mh_last_error(ctx));
}

template<typename F> double time_ms(F&& f)
{
    const auto t0 = std::chrono::high_resolution_clock::now();
    f();
    return std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
}

inline std::string hex_word(unsigned v)
{
    char buf[16];
    std::snprintf(buf, sizeof buf, "0x%x", v);
    return buf;
}

// A device stage cannot throw: it leaves status bits and the first failing cell (mh_step_result, include/mara_hip.h). Any bit ends the
// run here with the exception the reference would have thrown for it - mara::srhd::recover_primitive's std::invalid_argument and its
// four texts (src/physics_srhd.hpp:430-449) for the relativistic system; for mara::euler, whose recover_primitive never throws
// upstream, a std::runtime_error that names what the kernel saw.
inline void throw_on_result(const mh_step_result& r, int system = MH_SYSTEM_EULER)
{
    if (r.status == 0) return;
    const std::string where = " (first failing cell: flat index " + std::to_string((unsigned long long) r.first_bad_index) + ", device status " + hex_word((unsigned) r.status) + ")";
    if (system == MH_SYSTEM_SRHD)
    {
        const std::string head = "mara::srhd::recover_primitive failure: ";
        if (r.status & MH_STATUS_C2P_FAILED)   throw std::invalid_argument(head + "root finder not converging" + where);
        if (r.status & MH_STATUS_NEG_DENSITY)  throw std::invalid_argument(head + "negative density" + where);
        if (r.status & MH_STATUS_NEG_PRESSURE) throw std::invalid_argument(head + "negative pressure" + where);
        throw std::invalid_argument(head + "nan W" + where);
    }
    if (r.status & MH_STATUS_NEG_DENSITY)  throw std::runtime_error("negative density in updated state" + where);
    if (r.status & MH_STATUS_NEG_PRESSURE) throw std::runtime_error("negative pressure in recovered primitive state" + where);
    if (r.status & MH_STATUS_NAN)          throw std::runtime_error("nan in updated state" + where);
    throw std::runtime_error("device status word not clean" + where);
}

inline void throw_on_status(mh_ctx* ctx, int system = MH_SYSTEM_EULER)
{
    mh_step_result r;
    check(mh_status(ctx, &r), ctx, "mh_status");
    throw_on_result(r, system);
}

// <outdir>/<name>: [int64 rank][int64 shape...][int64 nq][f64 time][int64 iteration][f64 vertices (1-D only)...][f64 data...]
inline void dump_state(const std::string& outdir, const std::string& name, const std::vector<long>& shape, long nq,
                       double time, long iteration, const std::vector<double>& vertices, const std::vector<double>& data)
{
    if (! outdir.empty()) mkdir(outdir.c_str(), 0755);
    const std::string path = outdir.empty() ? name : outdir + "/" + name;
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (! f) throw std::runtime_error("cannot write " + path);
    long rank = (long) shape.size();
    std::fwrite(&rank, sizeof rank, 1, f);
    std::fwrite(shape.data(), sizeof(long), shape.size(), f);
    std::fwrite(&nq, sizeof nq, 1, f);
    std::fwrite(&time, sizeof time, 1, f);
    std::fwrite(&iteration, sizeof iteration, 1, f);
    long nv = (long) vertices.size();
    std::fwrite(&nv, sizeof nv, 1, f);
    std::fwrite(vertices.data(), sizeof(double), vertices.size(), f);
    std::fwrite(data.data(), sizeof(double), data.size(), f);
    std::fclose(f);
    std::printf("write %s\n", path.c_str());
}

} // namespace host

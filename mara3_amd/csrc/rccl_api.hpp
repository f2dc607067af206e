// RCCL bound at run time (dlopen "librccl.so.1": inside a torch process that is the copy torch already loaded), so that
// libmara_hip.so has no link-time dependency on it and single-GPU hosts never touch it. Shared by the slab and block steppers.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include "launch.hpp"

namespace mh {

struct RcclApi
{
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi* rccl();                                        // nullptr when librccl cannot be loaded (slab.hip)
int rccl_fail(ncclResult_t r, const char* what);        // sets the thread's error text, returns MH_E_HIP

} // namespace mh

// A communicator the host creates once per process and lends to the steppers it builds (mh_comm_create / mh_*_use_comm, slab.hip)
struct mh_comm
{
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

#define MH_RCCL_TRY(call) do { ncclResult_t _r = (call); if (_r != ncclSuccess) return mh::rccl_fail(_r, #call); } while (0)

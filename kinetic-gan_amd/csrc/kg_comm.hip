// Data-parallel gradient exchange of the WGAN-GP step over RCCL / xGMI: one in-place sum
// all-reduce of a flat fp32 gradient bucket per optimiser step (kinetic-gan.py:155,174 run the optimiser on every
// parameter; with one process per GPU the buckets of all ranks are summed first and kg_adam_step folds the 1 / world
// scaling in).  The reference has no distributed code at all; these entry points are what a data-parallel launcher of
// the reference's loop binds instead of torch.distributed / DistributedDataParallel.
//
// RCCL is bound LAZILY (dlopen at kg_comm_unique_id / kg_comm_init time, first the copy the process already holds -
// torch ships its own librccl.so.1 - then the ROCm one): libkgan_hip.so itself has no link-time dependency on it, so
// the single-GPU path and the CPU-side build / ABI checks never touch a communication library.
// No global mutable state besides the resolved function table (written once under a mutex, then read-only).
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "kg_common.h"

namespace {

// the slice of rccl.h this file needs (ABI-stable since NCCL 2.x: 128-byte opaque id, opaque communicator handle)
typedef struct { char internal[KG_COMM_ID_BYTES]; } NcclUniqueId;
typedef void* NcclComm;
enum { NCCL_SUCCESS = 0, NCCL_FLOAT32 = 7, NCCL_SUM = 0 };

struct Rccl {
    int (*GetUniqueId)(NcclUniqueId*);
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int);
    int (*CommDestroy)(NcclComm);
    int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
    int (*CommCount)(NcclComm, int*);
    const char* (*GetErrorString)(int);
    bool ok;
};

Rccl g_rccl;
std::once_flag g_once;
char g_dlerr[256] = "";          // dlerror() text of the LAST failed dlopen, captured once (a second dlerror() call returns NULL)

void resolve() {
    memset(&g_rccl, 0, sizeof(g_rccl));
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);      // the copy already mapped into this process
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        const char* e = dlerror();
        snprintf(g_dlerr, sizeof(g_dlerr), "%s", e ? e : "dlopen failed");
        return;
    }
    g_rccl.GetUniqueId = (int (*)(NcclUniqueId*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(NcclComm*, int, NcclUniqueId, int))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(NcclComm))dlsym(h, "ncclCommDestroy");
    g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, NcclComm, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.CommCount = (int (*)(NcclComm, int*))dlsym(h, "ncclCommCount");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce;
    if (!g_rccl.ok) snprintf(g_dlerr, sizeof(g_dlerr), "a required ncclXxx symbol is missing");
}

const Rccl* rccl() {
    std::call_once(g_once, resolve);
    return g_rccl.ok ? &g_rccl : nullptr;
}

int fail(const char* what, int rc) {
    const Rccl* r = rccl();
    kg_set_error("%s: RCCL error %d (%s)", what, rc, (r && r->GetErrorString) ? r->GetErrorString(rc) : "?");
    return rc > 0 ? rc : -1;
}

struct KgComm {
    NcclComm comm;
    int rank, world, device;
};

}  // namespace

extern "C" int kg_comm_unique_id(void* id) {
    KG_REQUIRE(id != nullptr, "kg_comm_unique_id: null id");
    const Rccl* r = rccl();
    KG_REQUIRE(r != nullptr, "kg_comm_unique_id: librccl.so.1 not found / incomplete: %s", g_dlerr);
    NcclUniqueId u;
    const int rc = r->GetUniqueId(&u);
    if (rc != NCCL_SUCCESS) return fail("kg_comm_unique_id", rc);
    memcpy(id, u.internal, KG_COMM_ID_BYTES);
    return 0;
}

extern "C" int kg_comm_init(void** comm, int32_t rank, int32_t world, const void* id, int32_t device) {
    KG_REQUIRE(comm != nullptr && id != nullptr, "kg_comm_init: null pointer");
    KG_REQUIRE(world >= 1 && rank >= 0 && rank < world, "kg_comm_init: rank %d of %d", rank, world);
    KG_REQUIRE(device >= 0, "kg_comm_init: device %d", device);
    const Rccl* r = rccl();
    KG_REQUIRE(r != nullptr, "kg_comm_init: librccl.so.1 not found / incomplete: %s", g_dlerr);
    int prev = -1;
    hipError_t e = hipGetDevice(&prev);
    if (e == hipSuccess) e = hipSetDevice(device);          // the communicator binds to the calling thread's device
    if (e != hipSuccess) {
        kg_set_error("kg_comm_init: hipSetDevice(%d): %s", device, hipGetErrorString(e));
        return (int)e;
    }
    NcclUniqueId u;
    memcpy(u.internal, id, KG_COMM_ID_BYTES);
    NcclComm c = nullptr;
    const int rc = r->CommInitRank(&c, world, u, rank);
    if (prev >= 0 && prev != device) {
        const hipError_t e2 = hipSetDevice(prev);       // back to the caller's device
        if (e2 != hipSuccess && rc == NCCL_SUCCESS) {
            kg_set_error("kg_comm_init: hipSetDevice(%d): %s", prev, hipGetErrorString(e2));
            r->CommDestroy(c);
            return (int)e2;
        }
    }
    if (rc != NCCL_SUCCESS) return fail("kg_comm_init", rc);
    KgComm* k = new KgComm{c, rank, world, device};
    *comm = k;
    return 0;
}

extern "C" int kg_comm_world(const void* comm) {
    KG_REQUIRE(comm != nullptr, "kg_comm_world: null communicator");
    return ((const KgComm*)comm)->world;
}

extern "C" int kg_allreduce_flat(void* comm, float* buf, int64_t n, void* stream) {
    KG_REQUIRE(comm != nullptr, "kg_allreduce_flat: null communicator");
    KG_REQUIRE(n >= 0 && (buf != nullptr || n == 0), "kg_allreduce_flat: null buffer");
    const Rccl* r = rccl();
    KG_REQUIRE(r != nullptr, "kg_allreduce_flat: RCCL not bound");
    if (n == 0) return 0;
    KgComm* k = (KgComm*)comm;
    {   // the communicator is bound to ONE device (kg_comm_init): a call from a thread whose current device is another
        // one would enqueue on a stream of the wrong device (round-3 ADVICE)
        int cur = -1;
        const hipError_t e = hipGetDevice(&cur);
        KG_REQUIRE(e == hipSuccess && cur == k->device, "kg_allreduce_flat: current device %d, communicator bound to device %d",
                   cur, k->device);
    }
    // in place, sum; enqueued on the caller's stream like every kernel of this library (no host synchronisation: the
    // call is legal inside a stream capture, RCCL records its kernels into the graph)
    const int rc = r->AllReduce(buf, buf, (size_t)n, NCCL_FLOAT32, NCCL_SUM, k->comm, (hipStream_t)stream);
    if (rc != NCCL_SUCCESS) return fail("kg_allreduce_flat", rc);
    return 0;
}

extern "C" int kg_comm_destroy(void* comm) {
    if (comm == nullptr) return 0;
    KgComm* k = (KgComm*)comm;
    const Rccl* r = rccl();
    int rc = NCCL_SUCCESS;
    if (r) rc = r->CommDestroy(k->comm);
    delete k;
    if (rc != NCCL_SUCCESS) return fail("kg_comm_destroy", rc);
    return 0;
}

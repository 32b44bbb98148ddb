// Gradient exchange behind the C ABI (cb_comm_init / cb_allreduce_bucket): one RCCL communicator per process
// (one process per GPU), in-place sum all-reduce of a bucket of the flat gradient buffer on a caller-given HIP stream.
//
// Reference: Horovod's NCCL all-reduce of every parameter gradient (src/tasks/run_video_retrieval.py:298-305, 432) -- one tensor at
// a time from the framework's background thread.  Here the host hands over whole buckets of ONE flat buffer (fp32, or the bf16
// wire image) and chooses the stream, so the exchange is ordered against the compute stream with HIP events only and can be
// captured into a hipGraph like any kernel launch.
//
// RCCL is resolved at run time (dlopen "librccl.so.1"): a process that already carries an RCCL (PyTorch-ROCm does) shares it,
// a single-GPU process that never calls cb_comm_* needs none, and the library loads on machines without RCCL.
// The five entry points used are declared below from RCCL's public API (rccl.h: ncclGetUniqueId, ncclCommInitRank,
// ncclAllReduce, ncclCommDestroy, ncclGetErrorString; ncclFloat32 = 7, ncclBfloat16 = 9, ncclSum = 0, 128-byte ncclUniqueId).
#include <dlfcn.h>
#include <string.h>

#include "common.h"

namespace {

struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*CommDestroyFn)(Comm);
typedef const char* (*ErrorStringFn)(int);

struct CommState {
    void* lib = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn init_rank = nullptr;
    AllReduceFn all_reduce = nullptr;
    CommDestroyFn destroy = nullptr;
    ErrorStringFn error_string = nullptr;
    Comm comm = nullptr;
    int rank = -1, world = 0;
};
CommState g_comm;

int load_rccl() {
    if (g_comm.lib) return 0;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return cb_fail("cb_comm: cannot load librccl.so.1 (%s)", dlerror());
    g_comm.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(h, "ncclGetUniqueId"));
    g_comm.init_rank = reinterpret_cast<CommInitRankFn>(dlsym(h, "ncclCommInitRank"));
    g_comm.all_reduce = reinterpret_cast<AllReduceFn>(dlsym(h, "ncclAllReduce"));
    g_comm.destroy = reinterpret_cast<CommDestroyFn>(dlsym(h, "ncclCommDestroy"));
    g_comm.error_string = reinterpret_cast<ErrorStringFn>(dlsym(h, "ncclGetErrorString"));
    if (!g_comm.get_unique_id || !g_comm.init_rank || !g_comm.all_reduce || !g_comm.destroy)
        return cb_fail("cb_comm: librccl.so.1 lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy");
    g_comm.lib = h;
    return 0;
}

int rccl_fail(const char* what, int rc) {
    return cb_fail("%s: RCCL error %d (%s)", what, rc, g_comm.error_string ? g_comm.error_string(rc) : "?");
}

}  // namespace

extern "C" int cb_comm_unique_id(void* id128) {
    CB_REQUIRE(id128, "cb_comm_unique_id: null output");
    if (load_rccl()) return -1;
    UniqueId id;
    const int rc = g_comm.get_unique_id(&id);
    if (rc != 0) return rccl_fail("cb_comm_unique_id", rc);
    memcpy(id128, id.internal, sizeof(id.internal));
    return 0;
}

extern "C" int cb_comm_init(int32_t rank, int32_t world, const void* id128) {
    CB_REQUIRE(id128 && world >= 1 && rank >= 0 && rank < world, "cb_comm_init: bad rank %d / world %d", rank, world);
    CB_REQUIRE(!g_comm.comm, "cb_comm_init: a communicator already exists (cb_comm_destroy first)");
    if (load_rccl()) return -1;
    UniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    Comm c = nullptr;
    const int rc = g_comm.init_rank(&c, world, id, rank);           // on the calling thread's current HIP device
    if (rc != 0) return rccl_fail("cb_comm_init", rc);
    g_comm.comm = c; g_comm.rank = rank; g_comm.world = world;
    return 0;
}

extern "C" int cb_comm_info(int32_t* rank, int32_t* world) {
    if (!g_comm.comm) return cb_fail("cb_comm_info: no communicator");
    if (rank) *rank = g_comm.rank;
    if (world) *world = g_comm.world;
    return 0;
}

extern "C" int cb_allreduce_bucket(void* buf, int64_t count, int32_t dtype, void* stream) {
    CB_REQUIRE(g_comm.comm, "cb_allreduce_bucket: cb_comm_init has not been called");
    CB_REQUIRE(buf && count >= 0 && (dtype == CB_F32 || dtype == CB_BF16), "cb_allreduce_bucket: bad arguments");
    if (count == 0) return 0;
    const int rc = g_comm.all_reduce(buf, buf, (size_t)count, dtype == CB_F32 ? 7 : 9, 0, g_comm.comm, cb_stream(stream));
    if (rc != 0) return rccl_fail("cb_allreduce_bucket", rc);
    return 0;
}

extern "C" int cb_comm_destroy(void) {
    if (!g_comm.comm) return 0;
    const int rc = g_comm.destroy(g_comm.comm);
    g_comm.comm = nullptr; g_comm.rank = -1; g_comm.world = 0;
    if (rc != 0) return rccl_fail("cb_comm_destroy", rc);
    return 0;
}

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
typedef int (*ReduceScatterFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, Comm, hipStream_t);
typedef int (*BroadcastFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*CommDestroyFn)(Comm);
typedef const char* (*ErrorStringFn)(int);

struct CommState {
    void* lib = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn init_rank = nullptr;
    AllReduceFn all_reduce = nullptr;
    ReduceScatterFn reduce_scatter = nullptr;
    AllGatherFn all_gather = nullptr;
    BroadcastFn broadcast = nullptr;
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
    g_comm.reduce_scatter = reinterpret_cast<ReduceScatterFn>(dlsym(h, "ncclReduceScatter"));
    g_comm.all_gather = reinterpret_cast<AllGatherFn>(dlsym(h, "ncclAllGather"));
    g_comm.broadcast = reinterpret_cast<BroadcastFn>(dlsym(h, "ncclBroadcast"));
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

// Reduce-scatter + all-gather: the two halves of an all-reduce with room for work in between (the optimizer on 1/world of the
// parameters).  On the point-to-point xGMI mesh both run as direct exchanges over all links.
// cb_reduce_scatter_bucket: rank r receives, in recv, the sum over ranks of elements [r*recv_count, (r+1)*recv_count) of every rank's
// send (send holds world * recv_count elements; recv may be send + rank * recv_count: in place).
// cb_allgather_bucket: every rank contributes send_count elements; recv (world * send_count elements) gets rank r's at offset
// r * send_count (send may be recv + rank * send_count: in place).
extern "C" int cb_reduce_scatter_bucket(const void* send, void* recv, int64_t recv_count, int32_t dtype, void* stream) {
    CB_REQUIRE(g_comm.comm, "cb_reduce_scatter_bucket: cb_comm_init has not been called");
    CB_REQUIRE(g_comm.reduce_scatter, "cb_reduce_scatter_bucket: this RCCL has no ncclReduceScatter");
    CB_REQUIRE(send && recv && recv_count >= 0 && (dtype == CB_F32 || dtype == CB_BF16), "cb_reduce_scatter_bucket: bad arguments");
    if (recv_count == 0) return 0;
    const int rc = g_comm.reduce_scatter(send, recv, (size_t)recv_count, dtype == CB_F32 ? 7 : 9, 0, g_comm.comm, cb_stream(stream));
    if (rc != 0) return rccl_fail("cb_reduce_scatter_bucket", rc);
    return 0;
}

extern "C" int cb_allgather_bucket(const void* send, void* recv, int64_t send_count, int32_t dtype, void* stream) {
    CB_REQUIRE(g_comm.comm, "cb_allgather_bucket: cb_comm_init has not been called");
    CB_REQUIRE(g_comm.all_gather, "cb_allgather_bucket: this RCCL has no ncclAllGather");
    CB_REQUIRE(send && recv && send_count >= 0 && (dtype == CB_F32 || dtype == CB_BF16), "cb_allgather_bucket: bad arguments");
    if (send_count == 0) return 0;
    const int rc = g_comm.all_gather(send, recv, (size_t)send_count, dtype == CB_F32 ? 7 : 9, g_comm.comm, cb_stream(stream));
    if (rc != 0) return rccl_fail("cb_allgather_bucket", rc);
    return 0;
}

// hvd.broadcast_parameters (run_video_retrieval.py:304): root's buffer to every rank, in place
extern "C" int cb_broadcast_bucket(void* buf, int64_t count, int32_t dtype, int32_t root, void* stream) {
    CB_REQUIRE(g_comm.comm, "cb_broadcast_bucket: cb_comm_init has not been called");
    CB_REQUIRE(g_comm.broadcast, "cb_broadcast_bucket: this RCCL has no ncclBroadcast");
    CB_REQUIRE(buf && count >= 0 && (dtype == CB_F32 || dtype == CB_BF16) && root >= 0 && root < g_comm.world, "cb_broadcast_bucket: bad arguments");
    if (count == 0) return 0;
    const int rc = g_comm.broadcast(buf, buf, (size_t)count, dtype == CB_F32 ? 7 : 9, root, g_comm.comm, cb_stream(stream));
    if (rc != 0) return rccl_fail("cb_broadcast_bucket", rc);
    return 0;
}

extern "C" int cb_comm_destroy(void) {
    if (!g_comm.comm) return 0;
    const int rc = g_comm.destroy(g_comm.comm);
    g_comm.comm = nullptr; g_comm.rank = -1; g_comm.world = 0;
    if (rc != 0) return rccl_fail("cb_comm_destroy", rc);
    return 0;
}

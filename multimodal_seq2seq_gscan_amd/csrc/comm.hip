// Data-parallel gradient exchange on the CALLER'S stream (SURVEY.md §8(b) `flat_allreduce`, §8(e)): one RCCL
// communicator per process, created once, and ncclAllReduce enqueued on the stream the backward kernels and the
// optimiser run on — no hop to a communication stream and back (torch.distributed's ProcessGroupNCCL runs every
// collective on its own stream: two cross-stream events, ~50 us of a 0.5 ms step).
//
// The reference has no distributed code (seq2seq/train.py:24,65: one process, one device); this replaces nothing in
// it and is what the data-parallel loop of this package adds around train.py:110-111.
//
// RCCL is resolved at RUN time from the copy the process already holds (PyTorch-ROCm loads its bundled librccl.so.1),
// not linked: the library keeps loading — and every single-GPU entry point keeps working — on a machine without RCCL,
// and a process never ends up with two RCCL instances.  Only five stable entry points of the NCCL API are used.
#include <dlfcn.h>
#include <string.h>

#include "step.h"

namespace gscan {

namespace {
struct UniqueId { char internal[GSCAN_COMM_ID_BYTES]; };     // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed by value
using GetUniqueIdFn = int (*)(UniqueId *);
using CommInitRankFn = int (*)(void **, int, UniqueId, int);
using AllReduceFn = int (*)(const void *, void *, size_t, int, int, void *, hipStream_t);
using CommDestroyFn = int (*)(void *);
using CommCountFn = int (*)(void *, int *);
using GetErrorStringFn = const char *(*)(int);
constexpr int kNcclFloat32 = 7, kNcclSum = 0;                // ncclDataType_t / ncclRedOp_t values of the NCCL 2 API

struct Rccl {
    void *handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    AllReduceFn all_reduce = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    CommCountFn comm_count = nullptr;
    GetErrorStringFn error_string = nullptr;
    bool tried = false;
};
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.all_reduce) return 0;
    GSCAN_CHECK(!g_rccl.tried, "comm: RCCL could not be loaded earlier in this process");
    g_rccl.tried = true;
    // the instance the process already has (torch's), else the system one
    const char *names[] = {"librccl.so.1", "librccl.so"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!h)
        for (const char *n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    GSCAN_CHECK(h, "comm: librccl.so.1 not found (%s): data-parallel steps need RCCL", dlerror());
    g_rccl.handle = h;
    g_rccl.get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
    g_rccl.comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
    g_rccl.error_string = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
    g_rccl.comm_count = (CommCountFn)dlsym(h, "ncclCommCount");
    AllReduceFn ar = (AllReduceFn)dlsym(h, "ncclAllReduce");
    GSCAN_CHECK(g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.error_string && ar,
                "comm: the RCCL library lacks one of ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / "
                "ncclCommDestroy / ncclGetErrorString");
    g_rccl.all_reduce = ar;
    return 0;
}

#define GSCAN_RCCL(call)                                                                                  \
    do {                                                                                                  \
        const int r_ = (call);                                                                            \
        if (r_ != 0) { set_error("%s failed: %s", #call, g_rccl.error_string(r_)); return 1; }            \
    } while (0)
}  // namespace

// Can this process reach RCCL at all?  Loads the library and resolves the entry points, nothing else: no device call, no
// communicator, nothing collective.  Every rank answers this BEFORE anyone enters ncclCommInitRank (which blocks until
// all ranks have joined), so that a rank without RCCL cannot leave the others waiting inside it (train.RcclCommunicator).
int comm_available() { return rccl_load(); }

int comm_unique_id(void *id_host) {
    GSCAN_CHECK(id_host, "comm_unique_id: NULL buffer");
    TRY_RC(rccl_load());
    GSCAN_RCCL(g_rccl.get_unique_id((UniqueId *)id_host));
    return 0;
}

int comm_init(void **comm, int nranks, int rank, const void *id_host) {
    GSCAN_CHECK(comm && id_host, "comm_init: NULL argument");
    GSCAN_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: rank %d of %d", rank, nranks);
    TRY_RC(rccl_load());
    UniqueId id;
    memcpy(&id, id_host, sizeof(id));
    *comm = nullptr;
    GSCAN_RCCL(g_rccl.comm_init_rank(comm, nranks, id, rank));
    return 0;
}

int comm_allreduce_f32(void *comm, float *buf, size_t n, hipStream_t stream) {
    GSCAN_CHECK(comm && buf, "allreduce_f32: NULL argument");
    if (n == 0) return 0;
    TRY_RC(rccl_load());
    GSCAN_RCCL(g_rccl.all_reduce(buf, buf, n, kNcclFloat32, kNcclSum, comm, stream));
    return 0;
}

int comm_count(void *comm, int *nranks) {
    GSCAN_CHECK(comm && nranks, "comm_count: NULL argument");
    TRY_RC(rccl_load());
    GSCAN_CHECK(g_rccl.comm_count, "comm_count: the RCCL library has no ncclCommCount");
    GSCAN_RCCL(g_rccl.comm_count(comm, nranks));
    return 0;
}

int comm_destroy(void *comm) {
    if (!comm) return 0;
    TRY_RC(rccl_load());
    GSCAN_RCCL(g_rccl.comm_destroy(comm));
    return 0;
}

}  // namespace gscan

// RCCL entry points of the C-ABI: the data-parallel gradient exchange for hosts that do not go through torch.distributed.
// one process per GPU, one exchange per step - the average of the flat fp32 gradient buffers, issued in
// buckets on a side HIP stream while backward is still running.  The reference has no collective on its live path (dead
// train_distill.py:48-64 shows the intended DistributedDataParallel recipe); the Python product drives the same exchange through
// torch.distributed (backend "nccl" = RCCL), host/ddp.py.
// RCCL is resolved with dlopen at the first call: a process that already carries an RCCL (PyTorch bundles one) keeps using that
// copy - two copies in one process would each run their own bootstrap / proxy threads.
#include "mi_common.h"
#include <dlfcn.h>
#include <string.h>

namespace {

typedef int (*fn_get_unique_id)(void*);
typedef int (*fn_comm_init_rank)(void**, int, ...);      // ncclUniqueId is passed BY VALUE (128 bytes): called through a typed thunk below
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*fn_error_string)(int);

struct UniqueId { char bytes[128]; };                    // == ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)
typedef int (*fn_comm_init_rank_typed)(void**, int, UniqueId, int);

struct Rccl {
    void* handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank_typed comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_error_string error_string = nullptr;
    bool tried = false;
};

Rccl& rccl() {
    static Rccl r;
    if (r.tried) return r;
    r.tried = true;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    // a copy that is already mapped (PyTorch's) wins: RTLD_NOLOAD first
    for (const char* n : names)
        if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    for (const char* n : names)
        if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) return r;
    r.get_unique_id = (fn_get_unique_id)dlsym(r.handle, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank_typed)dlsym(r.handle, "ncclCommInitRank");
    r.comm_destroy = (fn_comm_destroy)dlsym(r.handle, "ncclCommDestroy");
    r.all_reduce = (fn_all_reduce)dlsym(r.handle, "ncclAllReduce");
    r.error_string = (fn_error_string)dlsym(r.handle, "ncclGetErrorString");
    return r;
}

int rccl_fail(const char* what, int rc) {
    Rccl& r = rccl();
    return mi_set_error(MI_EHIP, "%s: RCCL error %d (%s)", what, rc, r.error_string ? r.error_string(rc) : "?");
}

}  // namespace

extern "C" int mi_comm_unique_id(void* id128) {
    MI_REQUIRE(id128, "mi_comm_unique_id: null");
    Rccl& r = rccl();
    MI_REQUIRE(r.get_unique_id, "mi_comm_unique_id: librccl.so not found");
    const int rc = r.get_unique_id(id128);
    return rc ? rccl_fail("mi_comm_unique_id", rc) : MI_OK;
}

extern "C" int mi_comm_init_rank(void** comm, int nranks, const void* id128, int rank) {
    MI_REQUIRE(comm && id128 && nranks > 0 && rank >= 0 && rank < nranks, "mi_comm_init_rank: bad argument");
    Rccl& r = rccl();
    MI_REQUIRE(r.comm_init_rank, "mi_comm_init_rank: librccl.so not found");
    UniqueId id;
    memcpy(id.bytes, id128, sizeof(id.bytes));
    const int rc = r.comm_init_rank(comm, nranks, id, rank);
    return rc ? rccl_fail("mi_comm_init_rank", rc) : MI_OK;
}

extern "C" int mi_comm_destroy(void* comm) {
    MI_REQUIRE(comm, "mi_comm_destroy: null");
    Rccl& r = rccl();
    MI_REQUIRE(r.comm_destroy, "mi_comm_destroy: librccl.so not found");
    const int rc = r.comm_destroy(comm);
    return rc ? rccl_fail("mi_comm_destroy", rc) : MI_OK;
}

// In-place all-reduce of one gradient bucket.  dtype: 0 = fp32, 1 = bf16.  average != 0: ncclAvg (sum / nranks), else ncclSum.
extern "C" int mi_allreduce_bucket(void* ptr, size_t count, int dtype, int average, void* comm, void* stream) {
    MI_REQUIRE(ptr && comm && count > 0, "mi_allreduce_bucket: bad argument");
    MI_REQUIRE(dtype == 0 || dtype == 1, "mi_allreduce_bucket: dtype %d (0 = fp32, 1 = bf16)", dtype);
    Rccl& r = rccl();
    MI_REQUIRE(r.all_reduce, "mi_allreduce_bucket: librccl.so not found");
    const int nccl_dtype = dtype == 0 ? 7 /* ncclFloat32 */ : 9 /* ncclBfloat16 */;
    const int rc = r.all_reduce(ptr, ptr, count, nccl_dtype, average ? 4 /* ncclAvg */ : 0 /* ncclSum */, comm, (hipStream_t)stream);
    return rc ? rccl_fail("mi_allreduce_bucket", rc) : MI_OK;
}

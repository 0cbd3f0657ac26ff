// comm.hip -- the one collective of the row-sharded index as a C entry point: an all-gather of every rank's top-k
// message (k x 8-byte keys [+ k best rows] + one count | overflow word) over RCCL, on the stream the selection ran on.
//
// SURVEY section 8(b) lists `ssw_topk_allgather(ssw_comm *, ...)` in the C-ABI: without it a caller that binds the
// library from the reference's side (option B) cannot use more than one GPU -- the Python package does
// the exchange through torch.distributed (seesaw_amd/sharded.py), which such a caller does not have.  RCCL is bound at
// run time (dlopen: the copy torch already mapped if there is one, so a process has a single RCCL), never at link time:
// a single-GPU user of libseesaw_hip.so does not load it.
// The reference has no counterpart (its index is one numpy array in host RAM).
#include <dlfcn.h>

#include <mutex>

#include "ssw_common.h"

using namespace ssw;

namespace {

struct NcclId {
    char internal[128];
};
typedef void *nccl_comm_t;
typedef int (*fn_get_unique_id)(NcclId *);
typedef int (*fn_comm_init_rank)(nccl_comm_t *, int, NcclId, int);
typedef int (*fn_all_gather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t);
typedef int (*fn_comm_destroy)(nccl_comm_t);
typedef const char *(*fn_error_string)(int);
constexpr int NCCL_UINT64 = 5;  // ncclUint64 (rccl.h)

struct Rccl {
    void *handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_error_string error_string = nullptr;
};

Rccl *rccl() {
    static Rccl lib;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (const char *n : names)  // the copy already in the process (torch's) first
            if (!lib.handle) lib.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char *n : names)
            if (!lib.handle) lib.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!lib.handle) return;
        lib.get_unique_id = (fn_get_unique_id)dlsym(lib.handle, "ncclGetUniqueId");
        lib.comm_init_rank = (fn_comm_init_rank)dlsym(lib.handle, "ncclCommInitRank");
        lib.all_gather = (fn_all_gather)dlsym(lib.handle, "ncclAllGather");
        lib.comm_destroy = (fn_comm_destroy)dlsym(lib.handle, "ncclCommDestroy");
        lib.error_string = (fn_error_string)dlsym(lib.handle, "ncclGetErrorString");
    });
    if (!lib.handle || !lib.get_unique_id || !lib.comm_init_rank || !lib.all_gather || !lib.comm_destroy) return nullptr;
    return &lib;
}

ssw_status nccl_fail(Rccl *r, const char *what, int rc) {
    set_error("%s failed: %s", what, r->error_string ? r->error_string(rc) : "RCCL error");
    return SSW_ERR_HIP;
}

}  // namespace

struct ssw_comm {
    int device = 0, rank = 0, world = 1;
    nccl_comm_t comm = nullptr;
};

extern "C" {

ssw_status ssw_comm_unique_id(void *out_id128) {
    SSW_REQUIRE(out_id128 != nullptr, "NULL argument");
    Rccl *r = rccl();
    SSW_REQUIRE(r != nullptr, "RCCL (librccl.so) is not loadable in this process");
    NcclId id;
    const int rc = r->get_unique_id(&id);
    if (rc != 0) return nccl_fail(r, "ncclGetUniqueId", rc);
    memcpy(out_id128, id.internal, sizeof(id.internal));
    return SSW_OK;
}

ssw_status ssw_comm_create(int32_t device, const void *unique_id128, int32_t rank, int32_t world, ssw_comm **out) {
    SSW_REQUIRE(out != nullptr && unique_id128 != nullptr, "NULL argument");
    *out = nullptr;
    SSW_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank %d outside [0, %d)", rank, world);
    Rccl *r = rccl();
    SSW_REQUIRE(r != nullptr, "RCCL (librccl.so) is not loadable in this process");
    DeviceGuard guard(device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", device);
        return SSW_ERR_HIP;
    }
    NcclId id;
    memcpy(id.internal, unique_id128, sizeof(id.internal));
    ssw_comm *c = new (std::nothrow) ssw_comm();
    if (!c) return SSW_ERR_NOMEM;
    c->device = device;
    c->rank = rank;
    c->world = world;
    const int rc = r->comm_init_rank(&c->comm, world, id, rank);
    if (rc != 0) {
        delete c;
        return nccl_fail(r, "ncclCommInitRank", rc);
    }
    *out = c;
    return SSW_OK;
}

ssw_status ssw_comm_destroy(ssw_comm *c) {
    if (!c) return SSW_OK;
    Rccl *r = rccl();
    if (r && c->comm) {
        DeviceGuard guard(c->device);
        (void)r->comm_destroy(c->comm);
    }
    delete c;
    return SSW_OK;
}

// every rank contributes msg_len u64 words from dev_send; dev_recv [world * msg_len] holds them in rank order.
// Enqueues on hip_stream (the stream the selection was enqueued on: no host synchronisation in between).
ssw_status ssw_topk_allgather(ssw_comm *c, void *hip_stream, const uint64_t *dev_send, uint64_t *dev_recv,
                              int32_t msg_len) {
    SSW_REQUIRE(c != nullptr && dev_send != nullptr && dev_recv != nullptr && msg_len > 0, "bad argument");
    Rccl *r = rccl();
    SSW_REQUIRE(r != nullptr, "RCCL (librccl.so) is not loadable in this process");
    DeviceGuard guard(c->device);
    const int rc = r->all_gather(dev_send, dev_recv, (size_t)msg_len, NCCL_UINT64, c->comm, (hipStream_t)hip_stream);
    if (rc != 0) return nccl_fail(r, "ncclAllGather", rc);
    return SSW_OK;
}

}  // extern "C"

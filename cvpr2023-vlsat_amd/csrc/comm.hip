// The one collective of the path: scenes are sharded over ranks with no data-path exchange, and a
// sharded evaluation ends with ONE all-reduce(sum) of a short fp64 vector of additive counts (what the reference's
// validation() derives its percentages from, src/model/model.py:214-242).  These entry points do that over RCCL without
// going through a Python framework: bootstrap as usual (rank 0 makes a unique id, the host shares its 128 bytes with the
// other ranks by any means -- a file, a pipe, torch.distributed -- and every rank joins with vlsat_comm_init on its own GPU).
//
// RCCL is resolved at run time, in this order: (1) symbols already visible to the process (RTLD_DEFAULT), (2) a copy that
// is already MAPPED but was loaded RTLD_LOCAL -- how Python extension modules load PyTorch's librccl.so -- picked up with
// dlopen(..., RTLD_NOLOAD), so that a process never ends up with two RCCL instances, (3) librccl.so.1 from the loader path.
// The few prototypes used are declared here (they are RCCL's stable C ABI): the library has neither a link-time nor a
// build-time dependency on RCCL, and single-GPU users never load it.
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <string>

#include "common.h"
#include "../../include/vlsat.h"

namespace {

// RCCL C ABI (rccl.h): opaque communicator, 128-byte unique id, result / datatype / reduction codes
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr int ncclDouble = 8;        // ncclFloat64
constexpr int ncclSum = 0;

struct Rccl {
    ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    const char* (*error_string)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void* lib = RTLD_DEFAULT;
        if (!dlsym(RTLD_DEFAULT, "ncclAllReduce")) {
            lib = nullptr;
            for (const char* name : {"librccl.so", "librccl.so.1"})          // already mapped (RTLD_LOCAL)? use that copy
                if (!lib) lib = dlopen(name, RTLD_NOLOAD | RTLD_NOW);
            for (const char* name : {"librccl.so.1", "librccl.so"})
                if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!lib) { r.why = std::string("RCCL not found: ") + dlerror(); return; }
        }
        r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(dlsym(lib, "ncclGetUniqueId"));
        r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(dlsym(lib, "ncclCommInitRank"));
        r.all_reduce = reinterpret_cast<decltype(r.all_reduce)>(dlsym(lib, "ncclAllReduce"));
        r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(lib, "ncclCommDestroy"));
        r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(lib, "ncclGetErrorString"));
        r.ok = r.get_unique_id && r.comm_init_rank && r.all_reduce && r.comm_destroy;
        if (!r.ok) r.why = "RCCL symbols missing";
    });
    return r;
}

int check(ncclResult_t e, const char* what) {
    if (e == ncclSuccess) return 0;
    const Rccl& r = rccl();
    return vlsat::fail(VLSAT_EHIP, std::string(what) + ": " + (r.error_string ? r.error_string(e) : "RCCL error"));
}

}  // namespace

extern "C" {

int vlsat_comm_unique_id(void* out128) {
    if (!out128) return vlsat::fail(VLSAT_EINVAL, "comm_unique_id: null argument");
    Rccl& r = rccl();
    if (!r.ok) return vlsat::fail(VLSAT_ESTATE, r.why);
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes in the C ABI");
    return check(r.get_unique_id(static_cast<ncclUniqueId*>(out128)), "ncclGetUniqueId");
}

int vlsat_comm_init(const void* id128, int32_t n_ranks, int32_t rank, void** comm) {
    if (!id128 || !comm || n_ranks < 1 || rank < 0 || rank >= n_ranks) return vlsat::fail(VLSAT_EINVAL, "comm_init: bad argument");
    Rccl& r = rccl();
    if (!r.ok) return vlsat::fail(VLSAT_ESTATE, r.why);
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    const int rc = check(r.comm_init_rank(&c, n_ranks, id, rank), "ncclCommInitRank");
    if (rc) return rc;
    *comm = c;
    return 0;
}

// buf (device, fp64[n]) <- sum over ranks, in place, asynchronous on `stream`.  A collective: EVERY rank of the
// communicator must call it with the same n (n == 0 returns without entering the collective -- on every rank or on none).
int vlsat_metrics_allreduce(void* comm, double* buf, int32_t n, void* stream) {
    if (!comm || !buf || n < 0) return vlsat::fail(VLSAT_EINVAL, "metrics_allreduce: bad argument");
    if (n == 0) return 0;
    Rccl& r = rccl();
    if (!r.ok) return vlsat::fail(VLSAT_ESTATE, r.why);
    return check(r.all_reduce(buf, buf, (size_t)n, ncclDouble, ncclSum, static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)),
                 "ncclAllReduce");
}

void vlsat_comm_destroy(void* comm) {
    if (!comm) return;
    Rccl& r = rccl();
    if (r.ok) r.comm_destroy(static_cast<ncclComm_t>(comm));
}

}  // extern "C"

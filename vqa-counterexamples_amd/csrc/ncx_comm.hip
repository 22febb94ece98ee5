// ncx_comm.hip -- the C ABI's opaque RCCL handle ("no global state except an opaque handle for RCCL").
//
// Data parallelism over xGMI is one sum all-reduce of fp32 gradient blocks per step.  The Python host uses
// torch.distributed's RCCL backend for it; a host that binds the C ABI without torch gets the same collective here.  RCCL
// is resolved at first use with dlopen("librccl.so.1") -- the copy a framework has already loaded (same SONAME) or
// /opt/rocm's -- so the library keeps no link-time dependency on it and single-GPU users never load it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/neuralcx.h"

namespace {

struct Rccl {
    void* so;
    ncclResult_t (*get_unique_id)(ncclUniqueId*);
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*comm_destroy)(ncclComm_t);
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    bool ok;
};

Rccl* rccl() {
    static Rccl r{};
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) { r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.so) break; }
        if (r.so) {
            r.get_unique_id = (decltype(r.get_unique_id))dlsym(r.so, "ncclGetUniqueId");
            r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(r.so, "ncclCommInitRank");
            r.comm_destroy = (decltype(r.comm_destroy))dlsym(r.so, "ncclCommDestroy");
            r.all_reduce = (decltype(r.all_reduce))dlsym(r.so, "ncclAllReduce");
            r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_reduce;
        }
    }
    return r.ok ? &r : nullptr;
}

}  // namespace

struct ncx_comm { ncclComm_t nccl; int nranks, rank; };

extern "C" {

int ncx_comm_unique_id(void* id128) {
    if (!id128) return NCX_E_NULL;
    static_assert(sizeof(ncclUniqueId) == NCX_COMM_ID_BYTES, "ncclUniqueId size");
    Rccl* r = rccl();
    if (!r) return NCX_E_UNSUPPORTED;
    ncclUniqueId id;
    if (r->get_unique_id(&id) != ncclSuccess) return NCX_E_COMM;
    memcpy(id128, &id, sizeof id);
    return NCX_OK;
}

int ncx_comm_create(const void* id128, int32_t nranks, int32_t rank, ncx_comm** out) {
    if (!id128 || !out) return NCX_E_NULL;
    if (nranks < 1 || rank < 0 || rank >= nranks) return NCX_E_DIMS;
    Rccl* r = rccl();
    if (!r) return NCX_E_UNSUPPORTED;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncx_comm* c = (ncx_comm*)calloc(1, sizeof(ncx_comm));
    if (!c) return NCX_E_WORKSPACE;
    if (r->comm_init_rank(&c->nccl, nranks, id, rank) != ncclSuccess) { free(c); return NCX_E_COMM; }   // binds the CURRENT HIP device
    c->nranks = nranks; c->rank = rank;
    *out = c;
    return NCX_OK;
}

int ncx_comm_destroy(ncx_comm* c) {
    if (!c) return NCX_E_NULL;
    Rccl* r = rccl();
    const bool ok = r && r->comm_destroy(c->nccl) == ncclSuccess;
    free(c);
    return ok ? NCX_OK : NCX_E_COMM;
}

int ncx_allreduce(ncx_comm* c, float* buf, size_t n, void* stream) {
    if (!c) return NCX_E_NULL;
    if (n == 0) return NCX_OK;
    if (!buf) return NCX_E_NULL;
    Rccl* r = rccl();
    if (!r) return NCX_E_UNSUPPORTED;
    return r->all_reduce(buf, buf, n, ncclFloat, ncclSum, c->nccl, (hipStream_t)stream) == ncclSuccess ? NCX_OK : NCX_E_COMM;
}

}  // extern "C"

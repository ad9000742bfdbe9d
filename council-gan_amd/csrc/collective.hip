// The data path's collectives behind the C-ABI (cg_allgather_images).
//
// One communicator per process group a rank belongs to (the slice group for the image exchange,
// the member group for the replica gradient average), one process per GPU, RCCL over xGMI.  RCCL is
// NOT a link-time dependency of this library: it is resolved on the first cg_comm_* call -- the copy
// already mapped into the process if there is one (torch ships its own librccl.so.1), otherwise the
// ROCm installation's -- so a single-GPU host never touches it and a host without RCCL gets a clear
// error from the cg_comm_* calls only.
//
// Reference: the exchange replaces the in-process reads of the other members' images at
// /root/reference/trainer_council.py:853-856, 872-874 (the reference is single-process; there is no
// reference collective to mirror).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <new>

#include "cg_common.h"

static_assert(CG_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "cg_comm_unique_id hands out RCCL's id verbatim");

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    char why[256] = {0};
    bool ok = false;
};

Rccl rccl;
std::once_flag rccl_once;

template <class F>
bool sym(void* h, const char* name, F& out) {
    out = reinterpret_cast<F>(dlsym(h, name));
    return out != nullptr;
}

void rccl_load() {
    // 1. whatever the process has already mapped under RCCL's soname (torch.distributed's backend "nccl")
    static const char* const names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names)
        if (!rccl.handle) rccl.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    // 2. the loader's search path, then the ROCm installation
    static const char* const paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : paths)
        if (!rccl.handle) rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!rccl.handle) {
        snprintf(rccl.why, sizeof(rccl.why), "librccl.so.1 not found (%s)", dlerror());
        return;
    }
    void* h = rccl.handle;
    rccl.ok = sym(h, "ncclGetUniqueId", rccl.get_unique_id) && sym(h, "ncclCommInitRank", rccl.comm_init_rank) &&
              sym(h, "ncclCommDestroy", rccl.comm_destroy) && sym(h, "ncclAllGather", rccl.all_gather) &&
              sym(h, "ncclAllReduce", rccl.all_reduce) && sym(h, "ncclGetErrorString", rccl.error_string);
    if (!rccl.ok) snprintf(rccl.why, sizeof(rccl.why), "librccl.so.1 lacks an expected symbol");
}

int need_rccl(const char* who) {
    std::call_once(rccl_once, rccl_load);
    if (!rccl.ok) return cg_set_error(CG_ERR_LAUNCH, "%s: RCCL unavailable: %s", who, rccl.why);
    return CG_OK;
}

int nccl_rc(ncclResult_t r, const char* who) {
    if (r == ncclSuccess) return CG_OK;
    return cg_set_error(CG_ERR_LAUNCH, "%s: RCCL: %s", who, rccl.error_string(r));
}

}  // namespace

struct cg_comm {
    ncclComm_t comm;
    int rank, nranks;
};

extern "C" int cg_comm_unique_id(unsigned char* id) {
    CG_CHECK_ARG(id != nullptr, "cg_comm_unique_id: null pointer");
    int rc = need_rccl("cg_comm_unique_id");
    if (rc) return rc;
    ncclUniqueId u;
    rc = nccl_rc(rccl.get_unique_id(&u), "cg_comm_unique_id");
    if (rc) return rc;
    memcpy(id, u.internal, CG_COMM_ID_BYTES);
    return CG_OK;
}

extern "C" int cg_comm_create(const unsigned char* id, int rank, int nranks, cg_comm** out) {
    CG_CHECK_ARG(id && out, "cg_comm_create: null pointer");
    CG_CHECK_ARG(nranks >= 1 && rank >= 0 && rank < nranks, "cg_comm_create: rank %d of %d", rank, nranks);
    int rc = need_rccl("cg_comm_create");
    if (rc) return rc;
    ncclUniqueId u;
    memcpy(u.internal, id, CG_COMM_ID_BYTES);
    cg_comm* c = new (std::nothrow) cg_comm{nullptr, rank, nranks};
    if (!c) return cg_set_error(CG_ERR_WORKSPACE, "cg_comm_create: out of host memory");
    rc = nccl_rc(rccl.comm_init_rank(&c->comm, nranks, u, rank), "cg_comm_create");   // on the calling thread's HIP device
    if (rc) {
        delete c;
        return rc;
    }
    *out = c;
    return CG_OK;
}

extern "C" int cg_comm_destroy(cg_comm* c) {
    if (!c) return CG_OK;
    int rc = need_rccl("cg_comm_destroy");
    if (!rc) rc = nccl_rc(rccl.comm_destroy(c->comm), "cg_comm_destroy");
    delete c;
    return rc;
}

extern "C" int cg_comm_info(const cg_comm* c, int* rank, int* nranks) {
    CG_CHECK_ARG(c != nullptr, "cg_comm_info: null communicator");
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    return CG_OK;
}

extern "C" int cg_allgather_images(cg_comm* c, const float* send, float* recv, size_t elems_per_rank, cg_stream_t stream) {
    CG_CHECK_ARG(c && send && recv, "cg_allgather_images: null pointer");
    if (elems_per_rank == 0) return CG_OK;
    int rc = need_rccl("cg_allgather_images");
    if (rc) return rc;
    return nccl_rc(rccl.all_gather(send, recv, elems_per_rank, ncclFloat32, c->comm, cg_s(stream)), "cg_allgather_images");
}

extern "C" int cg_allreduce_sum(cg_comm* c, float* buf, size_t elems, cg_stream_t stream) {
    CG_CHECK_ARG(c && buf, "cg_allreduce_sum: null pointer");
    if (elems == 0) return CG_OK;
    int rc = need_rccl("cg_allreduce_sum");
    if (rc) return rc;
    return nccl_rc(rccl.all_reduce(buf, buf, elems, ncclFloat32, ncclSum, c->comm, cg_s(stream)), "cg_allreduce_sum");
}

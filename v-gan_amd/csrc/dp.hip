// Data-parallel exchange behind the C ABI: thin wrappers over RCCL for callers of the library that do not
// go through torch.distributed.  The step needs ONE collective: all-reduce(SUM) of the generator gradient (M_4, [d, L+1]
// with the collapsed chain) on the caller's stream, between the weight-gradient contraction and the optimiser -- the point
// of the reference's `backward(); step()` (src/vgan.py:618-619).  The reference itself has no collective.
//
// With the SHARDED front of the step (large batches: v-gan_amd/trainer.py) there is a second exchange before the Gram: the
// all-gather of the Y rows of the operand, their norms and the column keys (vgan_dp_allgather, in place, byte counts).
//
// RCCL is opened lazily (dlopen of librccl.so) so that the library loads -- and every other entry point works -- on hosts
// without it; one process per GPU, communicators created from an id that rank 0 obtains and the caller distributes.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <string>

#include "vgan_common.hpp"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*error_string)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;  // dlerror() text captured at the failing dlopen / dlsym (a later dl* call would have wiped it)
};

Rccl& rccl() {
    static Rccl r = [] {
        Rccl q;
        q.handle = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (q.handle == nullptr) q.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (q.handle == nullptr) {
            const char* e = dlerror();
            q.why = e ? e : "dlopen failed";
            return q;
        }
        auto sym = [&q](const char* name) -> void* {
            void* p = dlsym(q.handle, name);
            if (p == nullptr && q.why.empty()) {
                const char* e = dlerror();
                q.why = e ? e : (std::string("missing symbol ") + name);
            }
            return p;
        };
        q.get_unique_id = reinterpret_cast<decltype(q.get_unique_id)>(sym("ncclGetUniqueId"));
        q.comm_init_rank = reinterpret_cast<decltype(q.comm_init_rank)>(sym("ncclCommInitRank"));
        q.comm_destroy = reinterpret_cast<decltype(q.comm_destroy)>(sym("ncclCommDestroy"));
        q.all_reduce = reinterpret_cast<decltype(q.all_reduce)>(sym("ncclAllReduce"));
        q.all_gather = reinterpret_cast<decltype(q.all_gather)>(sym("ncclAllGather"));
        q.error_string = reinterpret_cast<decltype(q.error_string)>(sym("ncclGetErrorString"));
        q.ok = q.get_unique_id && q.comm_init_rank && q.comm_destroy && q.all_reduce && q.all_gather && q.error_string;
        return q;
    }();
    return r;
}

int fail(const char* what, ncclResult_t rc) {
    vgan::set_error("%s: %s", what, rccl().error_string ? rccl().error_string(rc) : "RCCL error");
    return VGAN_ERR_HIP;
}

}  // namespace

struct vgan_dp_comm {
    ncclComm_t comm;
    int nranks, rank;
};

#define VGAN_NEED_RCCL()                                                                      \
    do {                                                                                      \
        if (!rccl().ok) {                                                                     \
            ::vgan::set_error("RCCL is not available (librccl.so): %s", rccl().why.c_str());    \
            return VGAN_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

extern "C" int vgan_dp_unique_id(uint8_t* id) {
    VGAN_CHECK_ARG(id != nullptr);
    VGAN_NEED_RCCL();
    ncclUniqueId u;
    const ncclResult_t rc = rccl().get_unique_id(&u);
    if (rc != ncclSuccess) return fail("ncclGetUniqueId", rc);
    static_assert(sizeof(u) == VGAN_DP_ID_BYTES, "RCCL unique id size");
    memcpy(id, &u, sizeof(u));
    return VGAN_OK;
}

extern "C" int vgan_dp_comm_create(vgan_dp_comm** comm, int nranks, const uint8_t* id, int rank) {
    VGAN_CHECK_ARG(comm && id && nranks >= 1 && rank >= 0 && rank < nranks);
    VGAN_NEED_RCCL();
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    const ncclResult_t rc = rccl().comm_init_rank(&c, nranks, u, rank);
    if (rc != ncclSuccess) return fail("ncclCommInitRank", rc);
    *comm = new vgan_dp_comm{c, nranks, rank};
    return VGAN_OK;
}

extern "C" int vgan_dp_allreduce_sum(vgan_dp_comm* comm, float* buf, int64_t count, vgan_stream_t stream) {
    VGAN_CHECK_ARG(comm && buf && count > 0);
    VGAN_NEED_RCCL();
    const ncclResult_t rc = rccl().all_reduce(buf, buf, (size_t)count, ncclFloat32, ncclSum, comm->comm, (hipStream_t)stream);
    if (rc != ncclSuccess) return fail("ncclAllReduce", rc);
    return VGAN_OK;
}

extern "C" int vgan_dp_allgather(vgan_dp_comm* comm, void* buf, int64_t bytes_per_rank, vgan_stream_t stream) {
    VGAN_CHECK_ARG(comm && buf && bytes_per_rank > 0);
    VGAN_NEED_RCCL();
    // in place: rank r's contribution already sits at buf + r * bytes_per_rank (RCCL's in-place all-gather convention)
    const char* mine = static_cast<const char*>(buf) + (int64_t)comm->rank * bytes_per_rank;
    const ncclResult_t rc = rccl().all_gather(mine, buf, (size_t)bytes_per_rank, ncclChar, comm->comm, (hipStream_t)stream);
    if (rc != ncclSuccess) return fail("ncclAllGather", rc);
    return VGAN_OK;
}

extern "C" int vgan_dp_comm_destroy(vgan_dp_comm* comm) {
    VGAN_CHECK_ARG(comm != nullptr);
    VGAN_NEED_RCCL();
    const ncclResult_t rc = rccl().comm_destroy(comm->comm);
    delete comm;
    if (rc != ncclSuccess) return fail("ncclCommDestroy", rc);
    return VGAN_OK;
}

// sg_comm.hip -- the one exchange step of the multi-GPU path: all-gather of the ranks' packed feature sets.
//
// Frames are independent, so ranks (one process or thread per GPU) extract their own contiguous block of frames
// without talking to each other; afterwards every rank may need every frame's features.  The sets differ in size, so
// this is an all-gather-v: the per-frame counts are all-gathered first (RCCL ncclAllGather), every rank derives the
// same layout from them -- rank r's features, frames back to back without padding, start at feature rank_first[r] --
// compacts its own features into its slot of the output, and the slots are filled by one grouped set of
// ncclBroadcast calls, rank r the root of slot r.  Over a fully connected xGMI node every slot then crosses each link
// once.
//
// RCCL is bound at run time (dlopen of the librccl that ships beside the HIP runtime in use): libsiftgpu.so itself has
// no link-time dependency on it, and a single-GPU caller never touches it.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "sg_internal.h"

namespace sg {

// ---- compaction: [n_img][cap] padded feature slots -> frames back to back ------------------------------------------
// One column of workgroups per frame; a frame's first output feature is the sum of the counts before it.
__global__ __launch_bounds__(256) void compact_feats_k(const sg_packed_feature *__restrict__ in,
                                                       const unsigned int *__restrict__ counts, unsigned int n_img,
                                                       size_t cap, sg_packed_feature *__restrict__ out)
{
    __shared__ unsigned long long first_s;
    const unsigned int b = blockIdx.y;
    if (threadIdx.x < 64) {
        unsigned long long sum = 0;
        for (unsigned int i = threadIdx.x; i < b; i += 64)
            sum += min((size_t)counts[i], cap);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
            sum += __shfl_xor(sum, d);
        if (threadIdx.x == 0)
            first_s = sum;
    }
    __syncthreads();
    const size_t n16 = min((size_t)counts[b], cap) * (sizeof(sg_packed_feature) / 16);
    const uint4 *src = reinterpret_cast<const uint4 *>(in + (size_t)b * cap);
    uint4 *dst = reinterpret_cast<uint4 *>(out + first_s);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        dst[i] = src[i];
}

void launch_compact_feats(const sg_packed_feature *in, const unsigned int *counts, int n_img, size_t cap,
                          sg_packed_feature *out, hipStream_t s)
{
    if (n_img < 1)
        return;
    dim3 g(64, n_img);
    compact_feats_k<<<g, 256, 0, s>>>(in, counts, (unsigned int)n_img, cap, out);
}

// ---- RCCL, bound at run time ------------------------------------------------------------------------------------------
namespace {

typedef void *comm_t;
struct unique_id {
    char internal[128];
};
enum { kNcclSuccess = 0, kNcclChar = 0, kNcclUint32 = 3 }; // ncclResult_t / ncclDataType_t values of rccl.h

struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(unique_id *) = nullptr;
    int (*CommInitRank)(comm_t *, int, unique_id, int) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*CommCount)(comm_t, int *) = nullptr;
    int (*CommUserRank)(comm_t, int *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, comm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, comm_t, hipStream_t) = nullptr; // optional: the point-to-point form
    int (*Recv)(void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;
};

const Rccl &rccl()
{
    static const Rccl r = [] {
        Rccl v;
        // The copy that belongs to the HIP runtime this library is running on: a process may hold more than one ROCm
        // installation (a host framework ships its own), and an RCCL from one does not work on the runtime of another.
        // So: the directory of the libamdhip64 that resolved hipGetDeviceCount, by path (DEEPBIND: should another RCCL copy be
        // in the process already, this one still binds to its own functions); then whatever the loader finds by name.
        Dl_info info;
        if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                for (const char *n : {"librccl.so.1", "librccl.so"})
                    if (!v.lib)
                        v.lib = dlopen((dir + n).c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
            }
        }
        for (const char *n : {"librccl.so.1", "librccl.so"})
            if (!v.lib)
                v.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
        if (!v.lib) {
            const char *e = dlerror();
            v.why = std::string("RCCL is not available: ") + (e ? e : "dlopen failed");
            return v;
        }
        bool ok = true;
        auto sym = [&](const char *name) {
            void *p = dlsym(v.lib, name);
            if (!p) {
                ok = false;
                v.why = std::string("RCCL lacks ") + name;
            }
            return p;
        };
        v.GetUniqueId = (int (*)(unique_id *))sym("ncclGetUniqueId");
        v.CommInitRank = (int (*)(comm_t *, int, unique_id, int))sym("ncclCommInitRank");
        v.CommDestroy = (int (*)(comm_t))sym("ncclCommDestroy");
        v.CommCount = (int (*)(comm_t, int *))sym("ncclCommCount");
        v.CommUserRank = (int (*)(comm_t, int *))sym("ncclCommUserRank");
        v.AllGather = (int (*)(const void *, void *, size_t, int, comm_t, hipStream_t))sym("ncclAllGather");
        v.Broadcast = (int (*)(const void *, void *, size_t, int, int, comm_t, hipStream_t))sym("ncclBroadcast");
        v.GroupStart = (int (*)())sym("ncclGroupStart");
        v.GroupEnd = (int (*)())sym("ncclGroupEnd");
        v.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
        if (!ok)
            v.lib = nullptr;
        if (v.lib) { // not required: without them the exchange keeps to its broadcast form
            v.Send = (int (*)(const void *, size_t, int, int, comm_t, hipStream_t))dlsym(v.lib, "ncclSend");
            v.Recv = (int (*)(void *, size_t, int, int, comm_t, hipStream_t))dlsym(v.lib, "ncclRecv");
        }
        return v;
    }();
    return r;
}

} // namespace

const char *rccl_unavailable_reason()
{
    const Rccl &r = rccl();
    return r.lib ? nullptr : r.why.c_str();
}

int rccl_unique_id(void *id128)
{
    const Rccl &r = rccl();
    if (!r.lib)
        return -1;
    return r.GetUniqueId(static_cast<unique_id *>(id128));
}

int rccl_comm_create(const void *id128, int rank, int n_ranks, void **comm)
{
    const Rccl &r = rccl();
    if (!r.lib)
        return -1;
    unique_id id;
    memcpy(&id, id128, sizeof id);
    return r.CommInitRank(comm, n_ranks, id, rank);
}

int rccl_comm_destroy(void *comm)
{
    const Rccl &r = rccl();
    return r.lib ? r.CommDestroy(comm) : -1;
}

int rccl_comm_shape(void *comm, int *rank, int *n_ranks)
{
    const Rccl &r = rccl();
    if (!r.lib)
        return -1;
    int rc = r.CommCount(comm, n_ranks);
    return rc != kNcclSuccess ? rc : r.CommUserRank(comm, rank);
}

const char *rccl_error_string(int rc)
{
    const Rccl &r = rccl();
    return r.lib && rc >= 0 ? r.GetErrorString(rc) : "RCCL is not available";
}

int rccl_allgather_u32(void *comm, const unsigned int *send, unsigned int *recv, size_t count, hipStream_t s)
{
    return rccl().AllGather(send, recv, count, kNcclUint32, comm, s);
}

// Which form the all-gather-v takes: 0 (default) one broadcast per slot, 1 point-to-point -- every rank sends its
// slot to every peer and receives every peer's slot, all in one group: over a fully connected xGMI node a slot then
// crosses each link once and no link carries another pair's data.  SIFTGPU_ALLGATHERV=p2p|bcast, read at every call
// (the caller of a multi-rank run decides after trying both on the node at hand); every rank must use the same form.
static int allgatherv_mode()
{
    const char *e = getenv("SIFTGPU_ALLGATHERV");
    return e && strcmp(e, "p2p") == 0 ? 1 : 0;
}

// slot r of `base` (bytes first[r] .. first[r+1]) travels from rank r to every other rank
int rccl_allgatherv_bytes(void *comm, char *base, const size_t *first_byte, int rank, int n_ranks, hipStream_t s)
{
    const Rccl &r = rccl();
    int rc = r.GroupStart();
    if (rc != kNcclSuccess)
        return rc;
    if (allgatherv_mode() == 1 && r.Send && r.Recv) {
        const size_t mine = first_byte[rank + 1] - first_byte[rank];
        for (int k = 0; k < n_ranks && rc == kNcclSuccess; k++) {
            if (k == rank)
                continue;
            const size_t theirs = first_byte[k + 1] - first_byte[k];
            if (mine) // (every rank sees the same counts: the peer posts the matching receive exactly when mine != 0)
                rc = r.Send(base + first_byte[rank], mine, kNcclChar, k, comm, s);
            if (theirs && rc == kNcclSuccess)
                rc = r.Recv(base + first_byte[k], theirs, kNcclChar, k, comm, s);
        }
        const int rc_end = r.GroupEnd();
        return rc != kNcclSuccess ? rc : rc_end;
    }
    for (int k = 0; k < n_ranks; k++) {
        const size_t bytes = first_byte[k + 1] - first_byte[k];
        if (bytes == 0)
            continue; // every rank sees the same counts, so every rank skips the same slots
        rc = r.Broadcast(base + first_byte[k], base + first_byte[k], bytes, kNcclChar, k, comm, s);
        if (rc != kNcclSuccess) {
            (void)r.GroupEnd();
            return rc;
        }
    }
    return r.GroupEnd();
}

} // namespace sg

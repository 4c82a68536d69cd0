// C ABI, multi-GPU part (include/lf_mkd.h): the all-gather of descriptor shards over RCCL -- the one collective of the path
// (BASELINE configs[3]).  librccl is bound at run time (dlopen + dlsym with the header's own prototypes), so
// liblf_mkd.so loads on a machine without it and every other entry point works there.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "lf_mkd_internal.h"

struct lf_mkd_comm {
    ncclComm_t comm = nullptr;
    int n_ranks = 0, rank = 0, device = 0;
    int last_form = -1;   // LF_MKD_GATHER_* the latest lf_mkd_allgather_descriptors took (RING falls back to DIRECT on unequal shards)
};

namespace {

struct Rccl {
    void *lib = nullptr;
    std::string error;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};

// the loader's state: bound once, on first use
Rccl &state() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            const char *e = dlerror();
            r.error = std::string("cannot load librccl: ") + (e ? e : "?");
            return;
        }
#define LF_SYM(field, name)                                                 \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name));      \
    if (!r.field) {                                                         \
        r.error = std::string("librccl lacks ") + name;                     \
        r.lib = nullptr;                                                    \
        return;                                                             \
    }
        LF_SYM(GetUniqueId, "ncclGetUniqueId")
        LF_SYM(CommInitRank, "ncclCommInitRank")
        LF_SYM(CommDestroy, "ncclCommDestroy")
        LF_SYM(AllGather, "ncclAllGather")
        LF_SYM(Send, "ncclSend")
        LF_SYM(Recv, "ncclRecv")
        LF_SYM(GroupStart, "ncclGroupStart")
        LF_SYM(GroupEnd, "ncclGroupEnd")
        LF_SYM(GetErrorString, "ncclGetErrorString")
        LF_SYM(GetVersion, "ncclGetVersion")
#undef LF_SYM
    });
    return r;
}
// nullptr when RCCL cannot be used (the reason: rccl_error())
const Rccl *rccl() { return state().lib ? &state() : nullptr; }
const std::string &rccl_error() { return state().error; }

// One group of point-to-point transfers, rows of 128 f32: ncclGroupStart, a Send and / or a Recv per entry (empty ones are not
// posted), ncclGroupEnd -- always closed, also after a failed post.  The direct form of the gather and the loopback
// diagnostic both go through here.
struct Transfer {
    const float *send;
    uint64_t send_rows;
    int dst;
    float *recv;
    uint64_t recv_rows;
    int src;
};
int post_group(lf_mkd *h, const Rccl *r, lf_mkd_comm *c, const std::vector<Transfer> &ts, hipStream_t s) {
    constexpr uint64_t kW = LF_MKD_DESC_LEN;
    auto check = [&](ncclResult_t rc, const char *what) -> int {
        if (rc == ncclSuccess) return LF_MKD_OK;
        return lf_mkd_internal_fail(h, LF_MKD_ERR_COMM, std::string(what) + ": " + r->GetErrorString(rc));
    };
    if (int rc = check(r->GroupStart(), "ncclGroupStart")) return rc;
    int first = LF_MKD_OK;
    for (const Transfer &t : ts) {
        if (t.send_rows && first == LF_MKD_OK)
            first = check(r->Send(t.send, t.send_rows * kW, ncclFloat, t.dst, c->comm, s), "ncclSend");
        if (t.recv_rows && first == LF_MKD_OK)
            first = check(r->Recv(t.recv, t.recv_rows * kW, ncclFloat, t.src, c->comm, s), "ncclRecv");
    }
    const int end = check(r->GroupEnd(), "ncclGroupEnd");
    return first != LF_MKD_OK ? first : end;
}

}  // namespace

extern "C" {

int lf_mkd_comm_unique_id(uint8_t *id) {
    if (!id) return LF_MKD_ERR_BAD_ARG;
    const Rccl *r = rccl();
    if (!r) return LF_MKD_ERR_COMM;
    static_assert(sizeof(ncclUniqueId) == LF_MKD_COMM_ID_BYTES, "identifier size");
    ncclUniqueId u;
    if (r->GetUniqueId(&u) != ncclSuccess) return LF_MKD_ERR_COMM;
    for (int i = 0; i < LF_MKD_COMM_ID_BYTES; ++i) id[i] = static_cast<uint8_t>(u.internal[i]);
    return LF_MKD_OK;
}

int lf_mkd_comm_create(lf_mkd *h, const uint8_t *id, int32_t n_ranks, int32_t rank, lf_mkd_comm **out) {
    if (!h || !out) return LF_MKD_ERR_BAD_ARG;
    *out = nullptr;
    if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return lf_mkd_internal_fail(h, LF_MKD_ERR_BAD_ARG, "comm_create: bad identifier / rank / n_ranks");
    const Rccl *r = rccl();
    if (!r) return lf_mkd_internal_fail(h, LF_MKD_ERR_COMM, rccl_error());
    lfmkd::DeviceScope scope;
    if (scope.enter(lf_mkd_internal_device(h)) != hipSuccess) return lf_mkd_internal_fail(h, LF_MKD_ERR_HIP, "hipSetDevice");
    lf_mkd_comm *c = new (std::nothrow) lf_mkd_comm;
    if (!c) return LF_MKD_ERR_BAD_ARG;
    ncclUniqueId u;
    for (int i = 0; i < LF_MKD_COMM_ID_BYTES; ++i) u.internal[i] = static_cast<char>(id[i]);
    const ncclResult_t rc = r->CommInitRank(&c->comm, n_ranks, u, rank);
    if (rc != ncclSuccess) {
        delete c;
        return lf_mkd_internal_fail(h, LF_MKD_ERR_COMM, std::string("ncclCommInitRank: ") + r->GetErrorString(rc));
    }
    c->n_ranks = n_ranks;
    c->rank = rank;
    c->device = lf_mkd_internal_device(h);
    *out = c;
    return LF_MKD_OK;
}

int lf_mkd_comm_destroy(lf_mkd_comm *c) {
    if (!c) return LF_MKD_OK;
    const Rccl *r = rccl();
    int rc = LF_MKD_OK;
    if (r && c->comm) {
        lfmkd::DeviceScope scope;
        (void)scope.enter(c->device);
        if (r->CommDestroy(c->comm) != ncclSuccess) rc = LF_MKD_ERR_COMM;
    }
    delete c;
    return rc;
}

int lf_mkd_comm_info(const lf_mkd_comm *c, int32_t *rccl_version, int32_t *n_ranks, int32_t *rank) {
    if (rccl_version) {
        *rccl_version = 0;
        const Rccl *r = rccl();
        if (!r) return LF_MKD_ERR_COMM;
        int v = 0;
        if (r->GetVersion(&v) != ncclSuccess) return LF_MKD_ERR_COMM;
        *rccl_version = v;
    }
    if (n_ranks) *n_ranks = c ? c->n_ranks : 0;
    if (rank) *rank = c ? c->rank : 0;
    return LF_MKD_OK;
}

int lf_mkd_allgather_descriptors(lf_mkd *h, lf_mkd_comm *c, const uint64_t *counts, float *d_buf, int32_t mode,
                                 void *stream) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!c || !counts) return lf_mkd_internal_fail(h, LF_MKD_ERR_BAD_ARG, "allgather_descriptors: null communicator / counts");
    if (mode != LF_MKD_GATHER_DIRECT && mode != LF_MKD_GATHER_RING)
        return lf_mkd_internal_fail(h, LF_MKD_ERR_BAD_ARG, "allgather_descriptors: unknown mode");
    const Rccl *r = rccl();
    if (!r) return lf_mkd_internal_fail(h, LF_MKD_ERR_COMM, rccl_error());
    uint64_t total = 0;
    std::vector<uint64_t> off(size_t(c->n_ranks) + 1, 0);
    bool equal = true;
    for (int p = 0; p < c->n_ranks; ++p) {
        off[p + 1] = off[p] + counts[p];
        equal = equal && counts[p] == counts[0];
    }
    total = off[c->n_ranks];
    if (total == 0) return LF_MKD_OK;
    if (!d_buf) return lf_mkd_internal_fail(h, LF_MKD_ERR_BAD_ARG, "allgather_descriptors: null buffer");
    lfmkd::DeviceScope scope;
    if (scope.enter(lf_mkd_internal_device(h)) != hipSuccess) return lf_mkd_internal_fail(h, LF_MKD_ERR_HIP, "hipSetDevice");
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : lf_mkd_internal_stream(h);
    constexpr uint64_t kW = LF_MKD_DESC_LEN;
    if (mode == LF_MKD_GATHER_RING && equal && counts[0] > 0) {   // in place: the shard already sits at its offset
        const ncclResult_t rc = r->AllGather(d_buf + off[c->rank] * kW, d_buf, counts[0] * kW, ncclFloat, c->comm, s);
        if (rc != ncclSuccess)
            return lf_mkd_internal_fail(h, LF_MKD_ERR_COMM, std::string("ncclAllGather: ") + r->GetErrorString(rc));
        c->last_form = LF_MKD_GATHER_RING;
        return LF_MKD_OK;
    }
    // direct: this rank's shard to every peer, every peer's shard from it, as one group (peer order staggered by rank so
    // that at any position of the group the pairs are disjoint)
    std::vector<Transfer> ts;
    for (int step = 1; step < c->n_ranks; ++step) {
        const int dst = (c->rank + step) % c->n_ranks, src = (c->rank - step + c->n_ranks) % c->n_ranks;
        ts.push_back(Transfer{d_buf + off[c->rank] * kW, counts[c->rank], dst, d_buf + off[src] * kW, counts[src], src});
    }
    c->last_form = LF_MKD_GATHER_DIRECT;
    return post_group(h, r, c, ts, s);
}

int lf_mkd_comm_loopback(lf_mkd *h, lf_mkd_comm *c, const float *d_src, float *d_dst, uint64_t n_rows, void *stream) {
    if (!h) return LF_MKD_ERR_BAD_ARG;
    if (!c) return lf_mkd_internal_fail(h, LF_MKD_ERR_BAD_ARG, "comm_loopback: null communicator");
    if (n_rows && (!d_src || !d_dst)) return lf_mkd_internal_fail(h, LF_MKD_ERR_BAD_ARG, "comm_loopback: null buffer");
    if (n_rows && d_src == d_dst) return lf_mkd_internal_fail(h, LF_MKD_ERR_BAD_ARG, "comm_loopback: source and destination must differ");
    const Rccl *r = rccl();
    if (!r) return lf_mkd_internal_fail(h, LF_MKD_ERR_COMM, rccl_error());
    lfmkd::DeviceScope scope;
    if (scope.enter(lf_mkd_internal_device(h)) != hipSuccess) return lf_mkd_internal_fail(h, LF_MKD_ERR_HIP, "hipSetDevice");
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : lf_mkd_internal_stream(h);
    // (n_rows == 0: an empty group, as the gather posts for a rank whose peers all hold nothing)
    return post_group(h, r, c, {Transfer{d_src, n_rows, c->rank, d_dst, n_rows, c->rank}}, s);
}

int lf_mkd_comm_last_form(const lf_mkd_comm *c) { return c ? c->last_form : -1; }

}  // extern "C"

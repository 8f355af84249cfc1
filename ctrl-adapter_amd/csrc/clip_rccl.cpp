// Native RCCL transport of the frame-sharded clip (include/ctrl_hip.h: ctrl_clip_comm): the three exchanges of the adapter --
// all-to-all (or K|V all-gather) around the temporal transformer, the +-1-frame halo of the Conv3d, the all-reduce of the temporal
// GroupNorm sums -- enqueued straight on the forward's HIP stream as RCCL calls over xGMI, from C++.  No host callback into Python
// sits between two launches, so the sharded step is recordable into a hipGraph like the unsharded one (round 3 ran it eagerly through
// torch.distributed: +15 % at world 1 before a byte crossed a link).
//
// RCCL is resolved at run time (dlopen: the process usually has torch's copy loaded already, and a link-time dependency would pull
// a second one in).  The communicator is the library's own: rank 0 draws a unique id (ctrl_rccl_unique_id), the caller broadcasts
// its 128 bytes over whatever channel it has (clip_parallel.RcclTransport uses the torch.distributed group it is given, any backend),
// and every rank calls ctrl_rccl_comm_create with it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

#include "common.h"
#include "../../include/ctrl_hip.h"

namespace {
struct Api {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllToAll)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    std::string err;
};

Api* api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        // RTLD_NOLOAD first: the copy torch (or anything else) already brought into the process
        for (const char* n : {"librccl.so.1", "librccl.so"}) {
            a.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (a.h) break;
        }
        if (!a.h)
            for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                a.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
                if (a.h) break;
            }
        if (!a.h) { a.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return; }
#define SYM(f, name) a.f = (decltype(a.f))dlsym(a.h, name); if (!a.f) { a.err = std::string("librccl lacks ") + name; return; }
        SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
        SYM(GetErrorString, "ncclGetErrorString") SYM(AllReduce, "ncclAllReduce") SYM(AllGather, "ncclAllGather")
        SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
#undef SYM
        a.AllToAll = (decltype(a.AllToAll))dlsym(a.h, "ncclAllToAll");      // RCCL extension; pairwise send / recv without it
    });
    return &a;
}
}  // namespace

struct ctrl_rccl_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    char* ws = nullptr;
    int64_t ws_bytes = 0;
    int64_t bytes_sent = 0;
};

#define NCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess) CTRL_FAIL(std::string(#expr) + " -> " + api()->GetErrorString(r_));       \
    } while (0)

namespace {
int cb_all_gather(void* user, int64_t send_off, int64_t recv_off, int64_t bytes, void* stream) {
    ctrl_rccl_comm* c = (ctrl_rccl_comm*)user;
    c->bytes_sent += bytes * (c->world - 1);
    NCCL_TRY(api()->AllGather(c->ws + send_off, c->ws + recv_off, (size_t)bytes, ncclChar, c->comm, (hipStream_t)stream));
    return 0;
}
int cb_all_reduce(void* user, int64_t off, int64_t count, void* stream) {
    ctrl_rccl_comm* c = (ctrl_rccl_comm*)user;
    c->bytes_sent += 4 * count;
    NCCL_TRY(api()->AllReduce(c->ws + off, c->ws + off, (size_t)count, ncclFloat, ncclSum, c->comm, (hipStream_t)stream));
    return 0;
}
int cb_halo(void* user, int64_t sp, int64_t sn, int64_t rp, int64_t rn, int64_t bytes, void* stream) {
    ctrl_rccl_comm* c = (ctrl_rccl_comm*)user;
    hipStream_t s = (hipStream_t)stream;
    if (c->world == 1) return 0;                       // no neighbour on either side: the halo slots stay as the caller left them
    NCCL_TRY(api()->GroupStart());
    if (c->rank > 0) {
        NCCL_TRY(api()->Send(c->ws + sp, (size_t)bytes, ncclChar, c->rank - 1, c->comm, s));
        NCCL_TRY(api()->Recv(c->ws + rp, (size_t)bytes, ncclChar, c->rank - 1, c->comm, s));
        c->bytes_sent += bytes;
    }
    if (c->rank < c->world - 1) {
        NCCL_TRY(api()->Send(c->ws + sn, (size_t)bytes, ncclChar, c->rank + 1, c->comm, s));
        NCCL_TRY(api()->Recv(c->ws + rn, (size_t)bytes, ncclChar, c->rank + 1, c->comm, s));
        c->bytes_sent += bytes;
    }
    NCCL_TRY(api()->GroupEnd());
    return 0;
}
int cb_all_to_all(void* user, int64_t send_off, int64_t recv_off, int64_t bytes, void* stream) {
    ctrl_rccl_comm* c = (ctrl_rccl_comm*)user;
    hipStream_t s = (hipStream_t)stream;
    c->bytes_sent += bytes * (c->world - 1);
    if (api()->AllToAll) {
        NCCL_TRY(api()->AllToAll(c->ws + send_off, c->ws + recv_off, (size_t)bytes, ncclChar, c->comm, s));
        return 0;
    }
    NCCL_TRY(api()->GroupStart());
    for (int r = 0; r < c->world; ++r) {               // every peer pair over its own xGMI link
        NCCL_TRY(api()->Send(c->ws + send_off + (int64_t)r * bytes, (size_t)bytes, ncclChar, r, c->comm, s));
        NCCL_TRY(api()->Recv(c->ws + recv_off + (int64_t)r * bytes, (size_t)bytes, ncclChar, r, c->comm, s));
    }
    NCCL_TRY(api()->GroupEnd());
    return 0;
}
}  // namespace

extern "C" {
int ctrl_rccl_unique_id(void* out128) {
    CTRL_CHECK(out128, "rccl_unique_id: null buffer");
    CTRL_CHECK(api()->err.empty(), api()->err);
    ncclUniqueId id;
    NCCL_TRY(api()->GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(out128, &id, sizeof id);
    return 0;
}

int ctrl_rccl_comm_create(const void* id128, int rank, int world, ctrl_rccl_comm** out) {
    CTRL_CHECK(id128 && out && world >= 1 && rank >= 0 && rank < world, "rccl_comm_create: bad arguments");
    CTRL_CHECK(api()->err.empty(), api()->err);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ctrl_rccl_comm* c = new ctrl_rccl_comm;
    c->rank = rank; c->world = world; c->device = cur_device();
    ncclResult_t r = api()->CommInitRank(&c->comm, world, id, rank);        // collective over the ranks of the clip
    if (r != ncclSuccess) { delete c; CTRL_FAIL(std::string("ncclCommInitRank -> ") + api()->GetErrorString(r)); }
    *out = c;
    return 0;
}

void ctrl_rccl_comm_destroy(ctrl_rccl_comm* c) {
    if (!c) return;
    if (c->comm) (void)api()->CommDestroy(c->comm);
    delete c;
}

// fills the callback table for ctrl_adapter_forward_clip_sharded; `ws` = the exchange workspace (device memory of this rank)
int ctrl_rccl_comm_bind(ctrl_rccl_comm* c, void* ws, int64_t ws_bytes, int use_all_to_all, ctrl_clip_comm* cs) {
    CTRL_CHECK(c && cs && ws && ws_bytes > 0, "rccl_comm_bind: bad arguments");
    c->ws = (char*)ws; c->ws_bytes = ws_bytes;
    cs->rank = c->rank; cs->world = c->world;
    cs->ws = ws; cs->ws_bytes = ws_bytes;
    cs->all_gather = cb_all_gather; cs->all_reduce_sum_f32 = cb_all_reduce; cs->halo_exchange = cb_halo;
    cs->all_to_all = use_all_to_all ? cb_all_to_all : nullptr;
    cs->user = c; cs->ws_needed = 0; cs->next_lane = nullptr;
    return 0;
}

int64_t ctrl_rccl_comm_bytes_sent(const ctrl_rccl_comm* c) { return c ? c->bytes_sent : 0; }
}

// sr_comm.cpp -- the multi-GPU entry points of the C ABI (sr_comm_init and the sharded exchange):
// one process per GPU, RCCL point-to-point transfers of exactly the tile rows a strip needs, one grouped batch per
// image, on the context's stream.  The reference has no counterpart (it is a single-process NumPy program).
//
// RCCL is bound at run time (dlopen), never at link time: a Python host already carries PyTorch's own librccl.so and a
// process must not host two; a C / C++ host gets the ROCm one.  SR_RCCL_LIB names a specific file.  RTLD_LOCAL: RCCL's
// dependency librocm_smi64.so must not export its statics to a later libamd_smi.so (double free at exit otherwise).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "sr_ctx.h"
#include "sr_hip.h"

namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

RcclApi g_api;
std::once_flag g_once;

void bind_rccl()
{
    std::vector<std::string> names;
    if (const char *env = std::getenv("SR_RCCL_LIB")) names.push_back(env);
    for (const char *n : {"librccl.so", "librccl.so.1"}) names.push_back(n);
    // a copy the process already holds (PyTorch's) wins over loading a second one
    for (int pass = 0; pass < 2 && !g_api.lib; ++pass)
        for (const std::string &n : names) {
            g_api.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (g_api.lib) break;
        }
    if (!g_api.lib) {
        for (const char *n : {"/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            g_api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (g_api.lib) break;
        }
    }
    if (!g_api.lib) {
        const char *e = dlerror();
        g_api.why = std::string("librccl.so not loadable: ") + (e ? e : "?");
        return;
    }
    bool ok = true;
    auto sym = [&](const char *name) {
        void *p = dlsym(g_api.lib, name);
        if (!p) {
            ok = false;
            g_api.why = std::string("librccl.so lacks ") + name;
        }
        return p;
    };
    g_api.GetUniqueId = (decltype(g_api.GetUniqueId))sym("ncclGetUniqueId");
    g_api.CommInitRank = (decltype(g_api.CommInitRank))sym("ncclCommInitRank");
    g_api.CommDestroy = (decltype(g_api.CommDestroy))sym("ncclCommDestroy");
    g_api.CommCount = (decltype(g_api.CommCount))sym("ncclCommCount");
    g_api.CommUserRank = (decltype(g_api.CommUserRank))sym("ncclCommUserRank");
    g_api.GroupStart = (decltype(g_api.GroupStart))sym("ncclGroupStart");
    g_api.GroupEnd = (decltype(g_api.GroupEnd))sym("ncclGroupEnd");
    g_api.Send = (decltype(g_api.Send))sym("ncclSend");
    g_api.Recv = (decltype(g_api.Recv))sym("ncclRecv");
    g_api.AllReduce = (decltype(g_api.AllReduce))sym("ncclAllReduce");
    g_api.GetErrorString = (decltype(g_api.GetErrorString))sym("ncclGetErrorString");
    if (!ok) {
        dlclose(g_api.lib);
        g_api.lib = nullptr;
    }
}

int need_rccl(const char *who)
{
    std::call_once(g_once, bind_rccl);
    if (!g_api.lib) return sr_set_error(SR_ERR_UNSUPPORTED, "%s: %s", who, g_api.why.c_str());
    return SR_OK;
}

#define RCCLCHK(expr)                                                                                                  \
    do {                                                                                                               \
        ncclResult_t r_ = (expr);                                                                                      \
        if (r_ != ncclSuccess)                                                                                         \
            return sr_set_error(SR_ERR_COMM, "%s: %s (%s:%d)", #expr, g_api.GetErrorString(r_), __FILE__, __LINE__);    \
    } while (0)

}  // namespace

struct sr_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    bool own = false;
    bool broken = false;    // a post inside a group failed: nothing further is enqueued on this communicator
};

extern "C" {

int sr_comm_unique_id(void *id128)
{
    if (!id128) return sr_set_error(SR_ERR_INVALID_ARG, "sr_comm_unique_id: null buffer");
    int rc = need_rccl("sr_comm_unique_id");
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == SR_COMM_ID_BYTES, "SR_COMM_ID_BYTES must match ncclUniqueId");
    ncclUniqueId id;
    RCCLCHK(g_api.GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof id);
    return SR_OK;
}

int sr_comm_init(sr_ctx *ctx, const void *id128, int world, int rank, sr_comm **out)
{
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_comm_init: bad arguments (world %d, rank %d)", world, rank);
    int rc = need_rccl("sr_comm_init");
    if (rc) return rc;
    CTX_ENTER(ctx);                                   // the communicator binds to the context's device
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    sr_comm *c = new sr_comm;
    ncclResult_t r = g_api.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return sr_set_error(SR_ERR_COMM, "ncclCommInitRank(world %d, rank %d): %s", world, rank, g_api.GetErrorString(r));
    }
    c->world = world;
    c->rank = rank;
    c->own = true;
    *out = c;
    return SR_OK;
}

int sr_comm_wrap(void *nccl_comm, sr_comm **out)
{
    if (!nccl_comm || !out) return sr_set_error(SR_ERR_INVALID_ARG, "sr_comm_wrap: null argument");
    int rc = need_rccl("sr_comm_wrap");
    if (rc) return rc;
    sr_comm *c = new sr_comm;
    c->comm = (ncclComm_t)nccl_comm;
    ncclResult_t r = g_api.CommCount(c->comm, &c->world);
    if (r == ncclSuccess) r = g_api.CommUserRank(c->comm, &c->rank);
    if (r != ncclSuccess) {
        delete c;
        return sr_set_error(SR_ERR_COMM, "sr_comm_wrap: %s", g_api.GetErrorString(r));
    }
    *out = c;
    return SR_OK;
}

int sr_comm_info(const sr_comm *comm, int *world, int *rank)
{
    if (!comm) return sr_set_error(SR_ERR_INVALID_ARG, "sr_comm_info: null communicator");
    if (world) *world = comm->world;
    if (rank) *rank = comm->rank;
    return SR_OK;
}

int sr_comm_destroy(sr_comm *comm)
{
    if (!comm) return SR_OK;
    int rc = SR_OK;
    if (comm->own && comm->comm) {
        ncclResult_t r = g_api.CommDestroy(comm->comm);
        if (r != ncclSuccess) rc = sr_set_error(SR_ERR_COMM, "ncclCommDestroy: %s", g_api.GetErrorString(r));
    }
    delete comm;
    return rc;
}

int sr_comm_exchange(sr_ctx *ctx, sr_comm *comm, const sr_xfer *sends, int n_send, const sr_xfer *recvs, int n_recv)
{
    if (!comm || n_send < 0 || n_recv < 0 || (n_send && !sends) || (n_recv && !recvs))
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_comm_exchange: bad arguments");
    for (int pass = 0; pass < 2; ++pass) {
        const sr_xfer *x = pass ? recvs : sends;
        for (int i = 0; i < (pass ? n_recv : n_send); ++i)
            if (x[i].peer < 0 || x[i].peer >= comm->world || (x[i].bytes && !x[i].d_ptr))
                return sr_set_error(SR_ERR_INVALID_ARG, "sr_comm_exchange: %s %d: peer %d of %d, %zu bytes at %p", pass ? "recv" : "send",
                                    i, x[i].peer, comm->world, (size_t)x[i].bytes, x[i].d_ptr);
    }
    if (comm->broken) return sr_set_error(SR_ERR_COMM, "sr_comm_exchange: an earlier batch on this communicator failed; destroy it");
    CTX_ENTER(ctx);
    ProfScope prof(ctx, "comm_exchange");
    // one group: inside it the transfers of a pair match whatever their posting order, and two strips that send to
    // each other cannot deadlock
    RCCLCHK(g_api.GroupStart());
    ncclResult_t first = ncclSuccess;
    for (int i = 0; i < n_send && first == ncclSuccess; ++i)
        if (sends[i].bytes) first = g_api.Send(sends[i].d_ptr, (size_t)sends[i].bytes, ncclUint8, sends[i].peer, comm->comm, ctx->stream);
    for (int i = 0; i < n_recv && first == ncclSuccess; ++i)
        if (recvs[i].bytes) first = g_api.Recv(recvs[i].d_ptr, (size_t)recvs[i].bytes, ncclUint8, recvs[i].peer, comm->comm, ctx->stream);
    // A group must be closed; after a failed post the batch it launches is partial (peers will wait on the transfers that
    // were never posted) and the communicator is unusable: the caller must destroy it (documented in sr_hip.h).
    ncclResult_t end = g_api.GroupEnd();
    if (first != ncclSuccess) {
        comm->broken = true;
        return sr_set_error(SR_ERR_COMM, "sr_comm_exchange: ncclSend/ncclRecv: %s (partial batch; the communicator must be destroyed)",
                            g_api.GetErrorString(first));
    }
    if (end != ncclSuccess) return sr_set_error(SR_ERR_COMM, "sr_comm_exchange: ncclGroupEnd: %s", g_api.GetErrorString(end));
    return SR_OK;
}

int sr_comm_exchange_tile_rows(sr_ctx *ctx, sr_comm *comm, const sr_tile_rect *h_tiles, int n, int cn, const int *h_need,
                               const int *h_owner, const void *const *d_owned, const int64_t *owned_stride, void *const *d_recv)
{
    if (!comm || n < 1) return sr_set_error(SR_ERR_INVALID_ARG, "sr_comm_exchange_tile_rows: bad arguments");
    std::vector<sr_xfer> sends((size_t)n * comm->world), recvs((size_t)n);
    int ns = 0, nr = 0;
    int rc = sr_exchange_xfers(h_tiles, n, cn, comm->world, comm->rank, h_need, h_owner, d_owned, owned_stride, d_recv, sends.data(),
                               (int)sends.size(), &ns, recvs.data(), (int)recvs.size(), &nr);
    if (rc) return rc;
    return sr_comm_exchange(ctx, comm, sends.data(), ns, recvs.data(), nr);
}

int sr_laplacian_blend_sharded(sr_ctx *ctx, sr_comm *comm, sr_blend_plan *plan, const sr_tile_rect *h_tiles, int n, int cn,
                               const int *h_need, const int *h_owner, const void *const *d_owned, const int64_t *strides,
                               void *const *d_recv, uint8_t *d_canvas, int64_t canvas_stride)
{
    if (!comm || !plan || !h_tiles || !h_need || !h_owner || !d_owned || !strides || !d_recv || n < 1)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_laplacian_blend_sharded: bad arguments");
    // The exchange is enqueued on ctx's stream and the blend on the plan's: they must be the same context, or the blend
    // could read received rows before RCCL has written them.  The plan indexes n tiles of cn channels: the caller's must
    // be those.
    sr_ctx *plan_ctx = nullptr;
    int plan_n = 0, plan_cn = 0;
    if (!plan_describe(plan, &plan_ctx, &plan_n, &plan_cn))
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_laplacian_blend_sharded: null or destroyed plan");
    if (plan_ctx != ctx)
        return sr_set_error(SR_ERR_INVALID_ARG, "sr_laplacian_blend_sharded: the plan belongs to another context (the exchange "
                                                "and the blend must share one stream)");
    if (n != plan_n || cn != plan_cn)
        return sr_set_error(SR_ERR_SHAPE, "sr_laplacian_blend_sharded: %d tiles x %d channels given, the plan has %d x %d", n, cn,
                            plan_n, plan_cn);
    // tile base pointers as the blend wants them (validated on the host before anything is posted)
    std::vector<void *> base((size_t)n, nullptr);
    int rc = sr_sharded_tile_bases(h_tiles, n, cn, comm->world, comm->rank, h_need, h_owner, d_owned, strides, d_recv, base.data());
    if (rc) return rc;
    rc = sr_comm_exchange_tile_rows(ctx, comm, h_tiles, n, cn, h_need, h_owner, d_owned, strides, d_recv);
    if (rc) return rc;
    // same stream as the exchange: the blend starts when the rows have arrived
    return sr_laplacian_blend(plan, SR_U8, base.data(), strides, d_canvas, canvas_stride, nullptr);
}

int sr_comm_allreduce_f64(sr_ctx *ctx, sr_comm *comm, double *d_buf, int count)
{
    if (!comm || !d_buf || count < 1) return sr_set_error(SR_ERR_INVALID_ARG, "sr_comm_allreduce_f64: bad arguments");
    CTX_ENTER(ctx);
    RCCLCHK(g_api.AllReduce(d_buf, d_buf, (size_t)count, ncclFloat64, ncclSum, comm->comm, ctx->stream));
    return SR_OK;
}

}  // extern "C"

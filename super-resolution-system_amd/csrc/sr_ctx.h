// sr_ctx.h -- the device context shared by the .hip translation units of libsrhip.so (sr_engine.hip owns the
// definitions; sr_lpips.hip and sr_adjust.hip use them).  Internal: nothing here is part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

#include "sr_internal.h"

#define HIPCHK(expr)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return sr_set_error(SR_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                                __LINE__);                                                         \
    } while (0)

struct ProfPair {
    int name_id;
    hipEvent_t a, b;
};

// A small device table with a host shadow of what was last uploaded into it: per-step descriptor tables (tile
// pointers, strides, rectangles) rarely change between calls, and a host->device copy between two kernels costs a
// stream bubble of several microseconds -- identical content is not uploaded again.
struct CachedTable {
    void *d = nullptr;
    size_t cap = 0;
    std::vector<char> shadow;
};

struct sr_ctx {
    int device = 0;
    int num_cu = 256;               // compute units of the device (launch-shape heuristics)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::recursive_mutex mu;
    bool prof = false;
    std::string prof_only;          // when not empty: the one kernel family that is timed
    std::vector<std::string> prof_names;
    std::vector<ProfPair> prof_pairs;
    std::vector<hipEvent_t> ev_pool;
    // small reusable device scratch (resize tables, reduction partials, result words)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    // host copies of small tables handed to hipMemcpyAsync; released at the next stream sync
    std::vector<std::vector<char>> pending_host;
    size_t pending_bytes = 0;
    CachedTable extract_tab;                            // tile-extract descriptors
    CachedTable resize_tab;                             // cubic tables of the resized assessment
    CachedTable cubic_tab;                              // cubic tables of sr_resize_cubic_u8
    void *gray_planes = nullptr;                        // resized gray planes + SSE partials of the resized assessment
    size_t gray_planes_bytes = 0;
    // side streams of the final gather (its marched zones and the block kernel write disjoint parts of the canvas and
    // run beside each other) with the events that fork them off the context's stream and join them back
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t side_fork = nullptr, side_join[2] = {nullptr, nullptr};
};

// Enqueue a small host->device table upload whose source stays alive until the next sync.
hipError_t upload_small(sr_ctx *c, void *d_dst, const void *h_src, size_t bytes);
hipError_t stream_sync(sr_ctx *c);
// Upload into a fixed destination unless `shadow` shows the same bytes are already there.
hipError_t upload_if_changed(sr_ctx *c, void *d_dst, const void *h_src, size_t bytes, std::vector<char> &shadow);
// Same with a table that owns (and grows) its device buffer.
hipError_t upload_cached(sr_ctx *c, CachedTable &t, const void *h_src, size_t bytes);
bool ctx_is_live(const sr_ctx *c);
// What a blend plan was made for (sr_comm.cpp cross-checks the arguments of the sharded blend); false when the plan is
// null or destroyed.
struct sr_blend_plan;
bool plan_describe(const sr_blend_plan *p, sr_ctx **ctx, int *n, int *cn);
int ctx_scratch(sr_ctx *c, size_t bytes, void **out);
hipEvent_t prof_event(sr_ctx *c);
int check_launch(const char *what);
// Deterministic tree sum of part[n][ncomp] -> pointer (inside the two ping-pong buffers) to the ncomp results.
const double *reduce_partials(sr_ctx *ctx, const double *part, long long n, int ncomp, double *buf0, double *buf1);

struct Guard {
    sr_ctx *c;
    int prev = -1;
    bool ok = true;
    explicit Guard(sr_ctx *ctx) : c(ctx)
    {
        c->mu.lock();
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != c->device && hipSetDevice(c->device) != hipSuccess) ok = false;
    }
    ~Guard()
    {
        if (prev >= 0 && prev != c->device) (void)hipSetDevice(prev);
        c->mu.unlock();
    }
};

#define CTX_ENTER(ctx)                                                               \
    if (!ctx_is_live(ctx)) return sr_set_error(SR_ERR_INVALID_ARG, "%s: null or destroyed context", __func__); \
    Guard guard_(ctx);                                                               \
    if (!guard_.ok) return sr_set_error(SR_ERR_HIP, "%s: hipSetDevice(%d) failed", __func__, (ctx)->device)

struct ProfScope {
    sr_ctx *c;
    ProfPair p{};
    bool on;
    ProfScope(sr_ctx *ctx, const char *name) : c(ctx), on(ctx->prof)
    {
        if (on && !c->prof_only.empty() && c->prof_only != name) on = false;
        if (!on) return;
        int id = -1;
        for (size_t i = 0; i < c->prof_names.size(); ++i)
            if (c->prof_names[i] == name) id = (int)i;
        if (id < 0) {
            c->prof_names.push_back(name);
            id = (int)c->prof_names.size() - 1;
        }
        p.name_id = id;
        p.a = prof_event(c);
        p.b = prof_event(c);
        (void)hipEventRecord(p.a, c->stream);
    }
    ~ProfScope()
    {
        if (!on) return;
        (void)hipEventRecord(p.b, c->stream);
        c->prof_pairs.push_back(p);
    }
};

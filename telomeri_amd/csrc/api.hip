// telomeri_amd/csrc/api.hip -- the extern "C" surface of include/telomeri_hip.h: context, graph
// adoption / download, walk orchestration (work queue, overflow re-runs, packing), path sets.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>

#include "tlm_common.h"
#include "walk_kernels.h"

#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void tlm_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *tlm_last_error(void) { return g_err; }

extern "C" int tlm_abi_version(void) { return TLM_ABI_VERSION; }

void tlm_ctx_defer(tlm_ctx *ctx, std::function<void()> fn) {
    std::lock_guard<std::mutex> lk(ctx->deferred_mu);
    ctx->deferred.push_back(std::move(fn));
}

void tlm_ctx_start_deferred(tlm_ctx *ctx) {
    std::lock_guard<std::mutex> lk(ctx->deferred_mu);
    for (auto &fn : ctx->deferred) ctx->deferred_running.emplace_back(std::move(fn));
    ctx->deferred.clear();
}

void tlm_ctx_join_deferred(tlm_ctx *ctx) {
    tlm_ctx_start_deferred(ctx);
    std::vector<std::thread> running;
    {
        std::lock_guard<std::mutex> lk(ctx->deferred_mu);
        running.swap(ctx->deferred_running);
    }
    for (auto &t : running)
        if (t.joinable()) t.join();
}

extern "C" uint32_t tlm_walk_seed(int metric, uint32_t anchor_node, uint32_t r) {
    return 5489u + 0x9E3779B9u * (anchor_node + 1u) + 0x85EBCA6Bu * r + 0xC2B2AE35u * (uint32_t) metric;
}

// ---------------------------------------------------------------------------------------------
// context

struct Workspace {   // walk scratch kept across calls (bench loops); bitmaps stay all-zero between walks
    uint32_t waves = 0, cap = 0, bm_words = 0;
    uint32_t *scratch = nullptr, *bitmaps = nullptr;
    uint64_t ar_cap = 0;
    uint2 *arena = nullptr;
    uint32_t *cursor = nullptr;              // arena cursor (u64) at [2], filter count (u64) at [4], per-XCD work cursors
    // pinned host copy of the unit records of a call (38 MB for the whole A. thaliana config: a pageable, zero-filled
    // std::vector cost 10 ms per call)
    void *h_units = nullptr;
    size_t h_units_bytes = 0;
    int ensure_h_units(size_t bytes) {
        if (bytes <= h_units_bytes) return TLM_OK;
        if (h_units) (void) hipHostFree(h_units);
        h_units = nullptr;
        h_units_bytes = 0;
        TLM_HIP(hipHostMalloc(&h_units, bytes, hipHostMallocDefault));
        h_units_bytes = bytes;
        return TLM_OK;
    }
    void release_host() {
        if (h_units) (void) hipHostFree(h_units);
        h_units = nullptr;
        h_units_bytes = 0;
    }
    // (the owning stream has been synchronised, or the buffers are handed back in stream order)
    void release_scratch() {
        tlm_pool_free(scratch); tlm_pool_free(bitmaps);
        scratch = bitmaps = nullptr;
        waves = cap = bm_words = 0;
    }
    void release_arena() {
        tlm_pool_free(arena);
        arena = nullptr;
        ar_cap = 0;
    }
};

// A walk launch needs a stream, two events and a workspace.  Lane 0 is the context's own stream; lane 1 is a second,
// low-priority stream on which tlm_walk_builders runs the deterministic builders into the tail of the Monte-Carlo launch.
struct WalkLane {
    Workspace ws;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

struct CtxImpl : tlm_ctx {
    Workspace ws;           // lane 0 (with ctx->stream, ev0, ev1)
    WalkLane side;          // lane 1
    // "every workgroup of the Monte-Carlo launch has started": a host-mapped word the kernel sets (tlm_walk_builders)
    uint32_t *resident_host = nullptr, *resident_dev = nullptr;
    std::atomic<int> gate_released{0};
    bool force_full_kernel = false;     // TLM_WALK_FULL_KERNEL, read once when the context is made (tests, diagnostics)
};

extern "C" int tlm_ctx_create(int device, tlm_ctx **out) {
    TLM_CHECK(out, TLM_EINVAL, "tlm_ctx_create: null out");
    int n = 0;
    TLM_HIP(hipGetDeviceCount(&n));
    TLM_CHECK(device >= 0 && device < n, TLM_EINVAL, "tlm_ctx_create: device %d of %d", device, n);
    TLM_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    TLM_HIP(hipGetDeviceProperties(&prop, device));
    TLM_CHECK(std::strncmp(prop.gcnArchName, "gfx950", 6) == 0, TLM_EHIP,
              "tlm_ctx_create: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    CtxImpl *c = new CtxImpl();
    c->device = device;
    c->n_cus = prop.multiProcessorCount;
    c->force_full_kernel = getenv("TLM_WALK_FULL_KERNEL") != nullptr;
    int prio_lo = 0, prio_hi = 0;
    (void) hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);        // numerically lower = higher priority
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess
        || hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess
        || hipMalloc((void **) &c->ws.cursor, WALK_CURSOR_WORDS * 4) != hipSuccess
        || hipStreamCreateWithPriority(&c->side.stream, hipStreamNonBlocking, prio_lo) != hipSuccess
        || hipEventCreate(&c->side.ev0) != hipSuccess || hipEventCreate(&c->side.ev1) != hipSuccess
        || hipMalloc((void **) &c->side.ws.cursor, WALK_CURSOR_WORDS * 4) != hipSuccess
        || hipHostMalloc((void **) &c->pinned_small, TLM_PINNED_SMALL_BYTES, hipHostMallocDefault) != hipSuccess
        || hipHostMalloc((void **) &c->resident_host, 64, hipHostMallocMapped) != hipSuccess
        || hipHostGetDevicePointer((void **) &c->resident_dev, c->resident_host, 0) != hipSuccess) {
        tlm_set_error("tlm_ctx_create: stream/event creation failed: %s", hipGetErrorString(hipGetLastError()));
        tlm_ctx_destroy(c);         // releases whatever was created (every handle starts out null)
        return TLM_EHIP;
    }
    *out = c;
    return TLM_OK;
}

extern "C" void tlm_ctx_destroy(tlm_ctx *ctx) {
    if (!ctx) return;
    CtxImpl *c = static_cast<CtxImpl *>(ctx);
    (void) hipSetDevice(c->device);
    tlm_ctx_join_deferred(ctx);
    tlm_dist_finalize(ctx);
    if (c->stream) (void) hipStreamSynchronize(c->stream);
    if (c->side.stream) (void) hipStreamSynchronize(c->side.stream);
    c->ws.release_scratch();
    c->ws.release_arena();
    c->ws.release_host();
    (void) hipFree(c->ws.cursor);
    c->side.ws.release_scratch();
    c->side.ws.release_arena();
    c->side.ws.release_host();
    (void) hipFree(c->side.ws.cursor);
    if (c->resident_host) (void) hipHostFree(c->resident_host);
    if (c->pinned_small) (void) hipHostFree(c->pinned_small);
    if (c->xchg_pinned) (void) hipHostFree(c->xchg_pinned);
    if (c->peer_window) (void) hipIpcCloseMemHandle(c->peer_window);
    if (c->upload_window) (void) hipFree(c->upload_window);
    for (hipEvent_t e : c->phase_ev) (void) hipEventDestroy(e);
    if (c->side.ev0) (void) hipEventDestroy(c->side.ev0);
    if (c->side.ev1) (void) hipEventDestroy(c->side.ev1);
    if (c->side.stream) {
        tlm_pool_forget_stream(c->device, c->side.stream);      // graphs and path sets may outlive the context
        (void) hipStreamDestroy(c->side.stream);
    }
    if (c->ev0) (void) hipEventDestroy(c->ev0);
    if (c->ev1) (void) hipEventDestroy(c->ev1);
    if (c->stream) {
        tlm_pool_forget_stream(c->device, c->stream);
        (void) hipStreamDestroy(c->stream);
    }
    delete c;
}

extern "C" void *tlm_ctx_stream(tlm_ctx *ctx) { return ctx ? (void *) ctx->stream : nullptr; }

extern "C" int tlm_ctx_synchronize(tlm_ctx *ctx) {
    TLM_CHECK(ctx, TLM_EINVAL, "tlm_ctx_synchronize: null ctx");
    TLM_HIP(hipStreamSynchronize(ctx->stream));
    return TLM_OK;
}

extern "C" int tlm_ctx_set_walk_accounting(tlm_ctx *ctx, int on) {
    TLM_CHECK(ctx, TLM_EINVAL, "tlm_ctx_set_walk_accounting: null ctx");
    ctx->walk_accounting = on != 0;
    return TLM_OK;
}

extern "C" int tlm_ctx_set_arena_limit(tlm_ctx *ctx, uint64_t bytes) {
    TLM_CHECK(ctx, TLM_EINVAL, "tlm_ctx_set_arena_limit: null ctx");
    ctx->arena_limit = bytes;
    return TLM_OK;
}

extern "C" int tlm_ctx_memory_info(tlm_ctx *ctx, tlm_memory_info *out) {
    TLM_CHECK(ctx && out, TLM_EINVAL, "tlm_ctx_memory_info: null argument");
    tlm_pool_info(ctx->device, out);
    return TLM_OK;
}

extern "C" int tlm_ctx_trim(tlm_ctx *ctx) {
    TLM_CHECK(ctx, TLM_EINVAL, "tlm_ctx_trim: null ctx");
    CtxImpl *c = static_cast<CtxImpl *>(ctx);
    TLM_HIP(hipSetDevice(c->device));
    TLM_HIP(hipDeviceSynchronize());
    c->ws.release_scratch();
    c->ws.release_arena();
    c->side.ws.release_scratch();
    c->side.ws.release_arena();
    tlm_pool_trim(c->device);
    return TLM_OK;
}

extern "C" int tlm_ctx_release_workspaces(tlm_ctx *ctx) {
    TLM_CHECK(ctx, TLM_EINVAL, "tlm_ctx_release_workspaces: null ctx");
    CtxImpl *c = static_cast<CtxImpl *>(ctx);
    TLM_HIP(hipSetDevice(c->device));
    TLM_HIP(hipStreamSynchronize(c->stream));
    if (c->side.stream) TLM_HIP(hipStreamSynchronize(c->side.stream));
    c->ws.release_scratch();
    c->ws.release_arena();
    c->side.ws.release_scratch();
    c->side.ws.release_arena();
    return TLM_OK;
}

extern "C" int tlm_ctx_set_walk_tuning(tlm_ctx *ctx, int waves_per_cu, uint32_t path_capacity) {
    TLM_CHECK(ctx, TLM_EINVAL, "tlm_ctx_set_walk_tuning: null ctx");
    TLM_CHECK(waves_per_cu >= 0 && waves_per_cu <= 32, TLM_EINVAL, "waves_per_cu must be 0..32");
    ctx->waves_per_cu = waves_per_cu;
    ctx->path_capacity = path_capacity;
    return TLM_OK;
}

// ---------------------------------------------------------------------------------------------
// filter_score

extern "C" int tlm_filter_score(tlm_ctx *ctx, const tlm_filter_params *p, const tlm_records *rec, uint8_t *keep_dev,
                                float *si_dev, float *os_dev, float *es_dev, uint64_t *n_kept) {
    TLM_CHECK(ctx && p && rec, TLM_EINVAL, "tlm_filter_score: null argument");
    TLM_CHECK(rec->n == 0 || (rec->q_id && rec->t_id && rec->q_len && rec->q_start && rec->q_end && rec->t_len
                              && rec->t_start && rec->t_end && rec->matches), TLM_EINVAL,
              "tlm_filter_score: null column");
    TLM_HIP(hipSetDevice(ctx->device));
    CtxImpl *c = static_cast<CtxImpl *>(ctx);
    unsigned long long *cnt = (unsigned long long *) (c->ws.cursor + 4);
    TLM_HIP(hipMemsetAsync(cnt, 0, 8, ctx->stream));
    tlm_launch_filter_score(ctx->stream, *p, *rec, keep_dev, si_dev, os_dev, es_dev, cnt);
    TLM_HIP(hipGetLastError());
    unsigned long long h = 0;
    TLM_HIP(hipMemcpyAsync(&h, cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
    TLM_HIP(hipStreamSynchronize(ctx->stream));
    if (n_kept) *n_kept = h;
    return TLM_OK;
}

// ---------------------------------------------------------------------------------------------
// graph adoption / download

extern "C" void tlm_graph_destroy(tlm_graph *g) {
    if (!g) return;
    (void) hipSetDevice(g->device);
    (void) hipDeviceSynchronize();      // kernels of either walk lane may still read the graph
    void *bufs[] = {g->row_ptr, g->ent_es, g->ent_os, g->q_start, g->t_start, g->q_end, g->t_end, g->si, g->strand,
                    g->node_len, g->node_anchor, g->node_flags, g->node_name_id, g->anchors, g->det_off, g->w_row,
                    g->w_ent_es, g->w_ent_os, g->w_qt, g->kept_rec, g->w_old_of_new, g->w_anchors, g->w_m_es, g->w_m_os,
                    g->w_d16};
    for (void *b : bufs) tlm_pool_free(b, true);
    delete g;
}

template <class T> static int upload(T **dst, const T *src, size_t n, hipStream_t s) {
    if (int rc = tlm_pool_alloc((void **) dst, (n ? n : 1) * sizeof(T), s)) return rc;
    if (n && src) TLM_HIP(hipMemcpyAsync(*dst, src, n * sizeof(T), hipMemcpyHostToDevice, s));
    else if (n) TLM_HIP(hipMemsetAsync(*dst, 0, n * sizeof(T), s));
    return TLM_OK;
}

extern "C" int tlm_graph_from_csr(tlm_ctx *ctx, uint32_t n_nodes, uint64_t n_adj, const uint64_t *row_ptr,
                                  const uint32_t *nbr, const uint32_t *q_start, const uint32_t *q_end,
                                  const uint32_t *t_start, const uint32_t *t_end, const float *os, const float *si,
                                  const float *es, const uint8_t *strand, const uint32_t *node_len,
                                  const uint8_t *node_anchor, tlm_graph **out) {
    TLM_CHECK(ctx && out && row_ptr && node_len && node_anchor, TLM_EINVAL, "tlm_graph_from_csr: null argument");
    TLM_CHECK(n_adj == 0 || (nbr && q_start && t_start && os && es), TLM_EINVAL, "tlm_graph_from_csr: null column");
    TLM_CHECK(n_nodes < (1u << 31) && n_adj < (1ull << 32) - 64, TLM_ELIMIT, "tlm_graph_from_csr: graph too large");
    TLM_CHECK(row_ptr[0] == 0 && row_ptr[n_nodes] == n_adj, TLM_EINVAL, "tlm_graph_from_csr: row_ptr ends do not match");
    for (uint32_t i = 0; i < n_nodes; i++)
        TLM_CHECK(row_ptr[i] <= row_ptr[i + 1], TLM_EINVAL, "tlm_graph_from_csr: row_ptr not monotone at %u", i);
    for (uint64_t k = 0; k < n_adj; k++)
        TLM_CHECK(nbr[k] < n_nodes, TLM_EINVAL, "tlm_graph_from_csr: neighbour out of range at %llu", (unsigned long long) k);
    TLM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    tlm_graph *g = new tlm_graph();
    g->ctx = ctx;
    g->device = ctx->device;
    g->n_nodes = n_nodes;
    g->n_adj = n_adj;
    struct Guard { tlm_graph *g; ~Guard() { if (g) tlm_graph_destroy(g); } } guard{g};
    std::vector<uint32_t> rp((size_t) n_nodes + 1);
    for (uint32_t i = 0; i <= n_nodes; i++) rp[i] = (uint32_t) row_ptr[i];
    std::vector<uint2> ees(n_adj), eos(n_adj);
    for (uint64_t k = 0; k < n_adj; k++) {
        const uint32_t w = nbr[k] | ((uint32_t) (node_anchor[nbr[k]] != 0) << 31);
        uint32_t eb, ob;
        std::memcpy(&eb, &es[k], 4);
        std::memcpy(&ob, &os[k], 4);
        ees[k] = make_uint2(w, eb);
        eos[k] = make_uint2(w, ob);
    }
    std::vector<int32_t> name_id(n_nodes, -1);
    if (int rc = upload(&g->row_ptr, rp.data(), rp.size(), s)) return rc;
    if (int rc = upload(&g->ent_es, ees.data(), (size_t) n_adj, s)) return rc;
    if (int rc = upload(&g->ent_os, eos.data(), (size_t) n_adj, s)) return rc;
    if (int rc = upload(&g->q_start, q_start, (size_t) n_adj, s)) return rc;
    if (int rc = upload(&g->t_start, t_start, (size_t) n_adj, s)) return rc;
    if (int rc = upload(&g->q_end, q_end, (size_t) n_adj, s)) return rc;
    if (int rc = upload(&g->t_end, t_end, (size_t) n_adj, s)) return rc;
    if (int rc = upload(&g->si, si, (size_t) n_adj, s)) return rc;
    if (int rc = upload(&g->strand, strand, (size_t) n_adj, s)) return rc;
    if (int rc = upload(&g->node_len, node_len, (size_t) n_nodes, s)) return rc;
    if (int rc = upload(&g->node_anchor, node_anchor, (size_t) n_nodes, s)) return rc;
    if (int rc = upload(&g->node_name_id, name_id.data(), (size_t) n_nodes, s)) return rc;
    TLM_HIP(hipStreamSynchronize(s));
    g->info.n_kept = n_adj / 2;
    if (int rc = tlm_graph_finish(ctx, g)) return rc;
    guard.g = nullptr;
    *out = g;
    return TLM_OK;
}

extern "C" int tlm_graph_get_info(const tlm_graph *g, tlm_graph_info *out) {
    TLM_CHECK(g && out, TLM_EINVAL, "tlm_graph_get_info: null argument");
    *out = g->info;
    return TLM_OK;
}

extern "C" int tlm_graph_download(const tlm_graph *g, uint64_t *row_ptr, uint32_t *nbr, uint32_t *q_start,
                                  uint32_t *q_end, uint32_t *t_start, uint32_t *t_end, float *os, float *si, float *es,
                                  uint8_t *strand, uint32_t *node_len, uint8_t *node_anchor, int32_t *node_name_id) {
    TLM_CHECK(g, TLM_EINVAL, "tlm_graph_download: null graph");
    TLM_HIP(hipSetDevice(g->ctx->device));
    hipStream_t s = g->ctx->stream;
    const size_t N = g->n_nodes, A = (size_t) g->n_adj;
#define DL(dst, src, n) do { if ((dst) && (n)) TLM_HIP(hipMemcpyAsync((dst), (src), (n) * sizeof(*(dst)), hipMemcpyDeviceToHost, s)); } while (0)
    std::vector<uint32_t> rp;
    if (row_ptr) {
        rp.resize(N + 1);
        TLM_HIP(hipMemcpyAsync(rp.data(), g->row_ptr, (N + 1) * 4, hipMemcpyDeviceToHost, s));
    }
    std::vector<uint2> ees, eos;
    if ((nbr || es) && A) { ees.resize(A); TLM_HIP(hipMemcpyAsync(ees.data(), g->ent_es, A * 8, hipMemcpyDeviceToHost, s)); }
    if (os && A) { eos.resize(A); TLM_HIP(hipMemcpyAsync(eos.data(), g->ent_os, A * 8, hipMemcpyDeviceToHost, s)); }
    DL(q_start, g->q_start, A); DL(q_end, g->q_end, A); DL(t_start, g->t_start, A); DL(t_end, g->t_end, A);
    DL(si, g->si, A); DL(strand, g->strand, A); DL(node_len, g->node_len, N); DL(node_anchor, g->node_anchor, N);
    DL(node_name_id, g->node_name_id, N);
#undef DL
    TLM_HIP(hipStreamSynchronize(s));
    if (row_ptr) for (size_t i = 0; i <= N; i++) row_ptr[i] = rp[i];
    for (size_t k = 0; k < A; k++) {
        if (nbr) nbr[k] = ees[k].x & 0x7FFFFFFFu;
        if (es) std::memcpy(&es[k], &ees[k].y, 4);
        if (os) std::memcpy(&os[k], &eos[k].y, 4);
    }
    return TLM_OK;
}

extern "C" uint64_t tlm_mc_units(const tlm_graph *g, const tlm_walk_params *p) {
    if (!g || !p || p->rebuild_attempts <= 0) return 0;
    return (uint64_t) g->h_anchors.size() * (uint64_t) p->rebuild_attempts;
}

extern "C" uint64_t tlm_det_units(const tlm_graph *g) { return g ? g->h_det_off.back() : 0; }

extern "C" int tlm_graph_anchor_table(const tlm_graph *g, uint32_t *anchor_nodes, uint64_t *det_off) {
    TLM_CHECK(g, TLM_EINVAL, "tlm_graph_anchor_table: null graph");
    if (anchor_nodes && !g->h_anchors.empty()) std::memcpy(anchor_nodes, g->h_anchors.data(), g->h_anchors.size() * 4);
    if (det_off) std::memcpy(det_off, g->h_det_off.data(), g->h_det_off.size() * 8);
    return TLM_OK;
}

// ---------------------------------------------------------------------------------------------
// result packing kernels

// Exclusive scans over the units of (accepted ? 1 : 0) and (accepted ? n_nodes : 0): where each accepted path goes in the
// packed set.  rocPRIM device scans over transform iterators (a one-workgroup scan took 35 ms for 400 000 units, and as a
// single workgroup it could not even start while a persistent walk launch of the other lane filled the device).
struct AcceptedCount {
    __host__ __device__ uint32_t operator()(const UnitOut &u) const { return u.status == ST_ACCEPTED ? 1u : 0u; }
};
struct AcceptedNodes {
    __host__ __device__ uint64_t operator()(const UnitOut &u) const { return u.status == ST_ACCEPTED ? (uint64_t) u.nnodes : 0ull; }
};

__global__ void scan_totals_kernel(uint32_t n_units, const UnitOut *__restrict__ units, const uint32_t *__restrict__ p_index,
                                   const uint64_t *__restrict__ n_off, uint64_t *__restrict__ totals) {
    const UnitOut last = units[n_units - 1];
    const bool acc = last.status == ST_ACCEPTED;
    totals[0] = (uint64_t) p_index[n_units - 1] + (acc ? 1u : 0u);
    totals[1] = n_off[n_units - 1] + (acc ? (uint64_t) last.nnodes : 0ull);
}

__global__ __launch_bounds__(256) void pack_paths_kernel(uint32_t n_units, uint64_t unit_begin,
                                                         const uint64_t *__restrict__ unit_ids,
                                                         const UnitOut *__restrict__ units,
                                                         const uint32_t *__restrict__ p_index,
                                                         const uint64_t *__restrict__ n_off,
                                                         const uint2 *__restrict__ arena, uint64_t *__restrict__ o_unit,
                                                         int64_t *__restrict__ o_len, uint64_t *__restrict__ o_off,
                                                         uint32_t *__restrict__ o_nodes, uint32_t *__restrict__ o_slots) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t ui = wave; ui < n_units; ui += n_waves) {
        const UnitOut u = units[ui];
        if (u.status != ST_ACCEPTED) continue;
        const uint32_t pi = p_index[ui], n = u.nnodes;
        const uint64_t src = u.off, dst = n_off[ui];
        if (lane == 0) {
            o_unit[pi] = unit_ids ? unit_ids[ui] : unit_begin + ui;
            o_len[pi] = u.len;
            o_off[pi] = dst;
        }
        for (uint32_t i = lane; i < n; i += 64) {
            const uint2 e = arena[src + i];
            o_nodes[dst + i] = e.x;
            o_slots[dst + i] = e.y;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// walk orchestration

namespace {

struct UnitBufs {
    UnitOut *units = nullptr;
    uint32_t *p_index = nullptr;
    uint64_t *n_off = nullptr, *totals = nullptr;
    uint32_t *work_list = nullptr;
    uint64_t *unit_ids = nullptr;
    void *scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
    ~UnitBufs() {
        tlm_pool_free(units); tlm_pool_free(p_index); tlm_pool_free(n_off); tlm_pool_free(totals);
        tlm_pool_free(work_list); tlm_pool_free(unit_ids); tlm_pool_free(scan_tmp);
    }
};

template <class T> int pmalloc(T **p, size_t n, hipStream_t s) { return tlm_pool_alloc((void **) p, (n ? n : 1) * sizeof(T), s); }

int ensure_scratch(Workspace &w, hipStream_t stream, uint32_t waves, uint32_t cap, uint32_t bm_words) {
    if (w.waves >= waves && w.cap == cap && w.bm_words == bm_words) return TLM_OK;
    w.release_scratch();
    if (int rc = pmalloc(&w.scratch, (size_t) waves * cap * 3, stream)) return rc;
    if (int rc = pmalloc(&w.bitmaps, (size_t) waves * bm_words, stream)) return rc;
    TLM_HIP(hipMemsetAsync(w.bitmaps, 0, (size_t) waves * bm_words * 4, stream));
    w.waves = waves;
    w.cap = cap;
    w.bm_words = bm_words;
    return TLM_OK;
}

// Grow the arena to `need` entries, keeping the first `keep` entries.
int ensure_arena(Workspace &w, hipStream_t stream, uint64_t need, uint64_t keep) {
    if (w.ar_cap >= need) return TLM_OK;
    uint2 *nn = nullptr;
    if (int rc = pmalloc(&nn, (size_t) need, stream)) return rc;
    if (keep) {
        TLM_HIP(hipMemcpyAsync(nn, w.arena, keep * sizeof(uint2), hipMemcpyDeviceToDevice, stream));
        TLM_HIP(hipStreamSynchronize(stream));
    }
    w.release_arena();
    w.arena = nn;
    w.ar_cap = need;
    return TLM_OK;
}

// One launch over `n_slots` (1 or 2) metrics of the same units: work item = slot * n_units + unit.  Two metrics in one
// persistent launch share the work queue, which halves the share of the tail (walk lengths differ by 2.5x).
// `lane` 1 = the side stream and its workspace; `launched` (optional) is set once the first kernel of the call is in its
// stream, or when the call gives up earlier; `wait_for` (optional) is waited for before that first kernel is launched.
// `unit_list` (optional, host, strictly ascending, `unit_end` entries with unit_begin = 0): the units of the call as a list
// instead of a range -- what a rank of a sliced partition walks (tlm_dist_unit_share).
int run_walks(tlm_ctx *ctx, const tlm_graph *g, bool is_mc, int n_slots, const int *metrics, const tlm_walk_params *p,
              uint64_t unit_begin, uint64_t unit_end, tlm_pathset **out, tlm_counters *ctr, int lane = 0,
              std::atomic<int> *launched = nullptr, std::atomic<int> *wait_for = nullptr, bool wait_resident = true,
              const uint64_t *unit_list = nullptr) {
    struct Signal {         // whoever waits for this call's launch is let go when the call ends, whatever happened
        std::atomic<int> *f;
        std::atomic<int> *gate;
        ~Signal() {
            if (f) {
                f->store(1);
                gate->store(1);
            }
        }
    } signal{launched, &static_cast<CtxImpl *>(ctx)->gate_released};
    if (launched) static_cast<CtxImpl *>(ctx)->gate_released.store(0);
    TLM_CHECK(ctx && g && out && metrics, TLM_EINVAL, "walk: null argument");
    TLM_CHECK(g->ctx == ctx, TLM_EINVAL, "walk: graph belongs to another context");
    TLM_CHECK(n_slots == 1 || n_slots == 2, TLM_EINVAL, "walk: one or two metrics per launch");
    for (int k = 0; k < n_slots; k++) {
        TLM_CHECK(metrics[k] >= 0 && metrics[k] <= 3, TLM_EINVAL, "walk: metric %d", metrics[k]);
        TLM_CHECK((metrics[k] >= 2) == (metrics[0] >= 2), TLM_EINVAL, "walk: sqrt and plain metrics cannot share a launch");
    }
    TLM_CHECK(!is_mc || (p && p->rebuild_attempts >= 0 && p->backtrack_attempts >= 0), TLM_EINVAL, "walk: bad parameters");
    CtxImpl *c = static_cast<CtxImpl *>(ctx);
    TLM_HIP(hipSetDevice(ctx->device));
    (void) hipGetLastError();     // a stale error of an unrelated earlier call must not be blamed on this one
    Workspace &W = lane ? c->side.ws : c->ws;
    hipStream_t s = lane ? c->side.stream : ctx->stream;
    hipEvent_t ev0 = lane ? c->side.ev0 : ctx->ev0, ev1 = lane ? c->side.ev1 : ctx->ev1;
    const uint64_t total = is_mc ? tlm_mc_units(g, p) : tlm_det_units(g);
    if (unit_list) {
        TLM_CHECK(unit_begin == 0, TLM_EINVAL, "walk: a unit list starts at local index 0");
        for (uint64_t i = 0; i < unit_end; i++)
            TLM_CHECK(unit_list[i] < total && (i == 0 || unit_list[i - 1] < unit_list[i]), TLM_EINVAL,
                      "walk: unit list entry %llu (%llu) is out of range or not ascending", (unsigned long long) i,
                      (unsigned long long) unit_list[i]);
    } else {
        if (unit_end > total) unit_end = total;
        if (unit_begin > unit_end) unit_begin = unit_end;
    }
    const uint64_t n_units64 = unit_end - unit_begin;
    TLM_CHECK(n_units64 * (uint64_t) n_slots < (1ull << 31), TLM_ELIMIT, "walk: %llu units in one call (limit 2^31)",
              (unsigned long long) n_units64);
    const uint32_t n_units = (uint32_t) n_units64;
    const uint32_t n_all = n_units * (uint32_t) n_slots;

    tlm_pathset *pss[2] = {nullptr, nullptr};
    struct Guard {
        tlm_pathset **p;
        ~Guard() { if (p) { tlm_pathset_destroy(p[0]); tlm_pathset_destroy(p[1]); } }
    } guard{pss};
    for (int k = 0; k < n_slots; k++) {
        pss[k] = new tlm_pathset();
        pss[k]->ctx = ctx;
        pss[k]->device = ctx->device;
        pss[k]->n_units = n_units;
    }
    tlm_counters cnt[2];
    std::memset(cnt, 0, sizeof(cnt));

    UnitBufs u;
    if (int rc = pmalloc(&u.units, n_all, s)) return rc;
    if (int rc = pmalloc(&u.p_index, n_units, s)) return rc;
    if (int rc = pmalloc(&u.n_off, n_units, s)) return rc;
    if (int rc = pmalloc(&u.totals, 2, s)) return rc;
    if (int rc = pmalloc(&u.work_list, n_all, s)) return rc;
    if (unit_list && n_units) {
        if (int rc = pmalloc(&u.unit_ids, n_units, s)) return rc;
        TLM_HIP(hipMemcpyAsync(u.unit_ids, unit_list, (size_t) n_units * 8, hipMemcpyHostToDevice, s));
        TLM_HIP(hipStreamSynchronize(s));           // (the caller's array is read now, not when the stream gets there)
    }
    TLM_HIP(hipMemsetAsync(u.units, 0xFF, (size_t) (n_all ? n_all : 1) * sizeof(UnitOut), s));   // status = not run

    if (int rc = W.ensure_h_units(((size_t) n_all + 1) * sizeof(UnitOut))) return rc;
    UnitOut *const h_units = static_cast<UnitOut *>(W.h_units);

    // a call that gives up between a launch and its completion leaves the per-wave visited maps in an unknown state:
    // the workspace is dropped then (after the stream has drained), the next call starts from zeroed maps
    struct Dirty {
        Workspace &w;
        hipStream_t s;
        bool armed = false;
        ~Dirty() {
            if (armed) {
                (void) hipStreamSynchronize(s);
                w.release_scratch();
            }
        }
    } dirty{W, s};

    if (n_units) {
        const uint32_t N = g->n_nodes;
        // a walk pushes at most N nodes (a node is entered once) and logs, per push, at most one far mark AND one band move
        // (the band follows the walk), plus one band move per go-back -- and there are at most as many go-backs as pushes:
        // 3 N log entries bound every walk, whatever the node order and however large backtrack_attempts is
        const uint32_t full_cap = (uint32_t) std::min<uint64_t>(3ull * N + 64u, 0xFFFFFFF0ull);
        uint32_t cap = ctx->path_capacity ? ctx->path_capacity : 16384u;
        cap = std::max<uint32_t>(8u, std::min(cap, full_cap));
        const uint32_t bm_words = (N + 31) / 32 + 1 + WALK_RNG_WORDS;
        // persistent grid = what is resident at once: blocks/CU from the occupancy query (a block is 4
        // waves), optionally capped by the tuning knob; never more than 8 blocks (32 waves) per CU
        const bool sq = metrics[0] >= 2;
        // the kernels without the selections for long lists (and, Monte-Carlo, for order-unsafe rows) where the graph has
        // neither: the same walks; the smaller kernels leave the register allocator more room on the paths that run (the
        // deterministic one then fits 8 waves per SIMD instead of 6)
        bool lite = g->info.max_degree <= 512u && !c->force_full_kernel && !ctx->walk_accounting;
        for (int k = 0; is_mc && k < n_slots; k++) lite = lite && g->info.unsafe_nodes[metrics[k]] == 0u;
        // (the instantiation that is LAUNCHED: the accounting builds hold more LDS and registers than the product's)
        int bpc = std::min(tlm_walk_blocks_per_cu(is_mc, sq, lite, ctx->walk_accounting != 0), 32 / WALK_WAVES_PER_BLOCK);
        if (ctx->waves_per_cu) bpc = std::max(1, std::min(bpc, ctx->waves_per_cu / WALK_WAVES_PER_BLOCK));
        uint32_t waves = (uint32_t) ctx->n_cus * (uint32_t) bpc * WALK_WAVES_PER_BLOCK;
        constexpr uint32_t WPB = WALK_WAVES_PER_BLOCK;
        waves = std::min<uint32_t>(waves, ((n_all + WPB - 1) / WPB) * WPB);
        waves = std::max<uint32_t>(WPB, (waves / WPB) * WPB);
        cnt[0].waves = waves;
        if (int rc = ensure_scratch(W, s, waves, cap, bm_words)) return rc;
        dirty.armed = true;
        // arena for the accepted paths: the worst case (every unit accepted at full buffer length) when that is small,
        // else 2048 nodes per unit (the A. thaliana-shaped walks leave 1630 accepted path nodes per unit: 3500 per path,
        // 46 % accepted) -- a guess that the overflow re-run below corrects, and that tlm_ctx_set_arena_limit can cap
        // (tests force the re-run with it).  Fresh device memory costs ~30 ms per GB to obtain: guessing high is not free.
        {
            const uint64_t worst = (uint64_t) n_all * cap;
            uint64_t want = std::min<uint64_t>(worst, std::max<uint64_t>((uint64_t) n_all * 2048u, 1u << 20));
            if (ctx->arena_limit) want = std::max<uint64_t>(64u, std::min<uint64_t>(want, ctx->arena_limit / sizeof(uint2)));
            if (int rc = ensure_arena(W, s, want, 0)) return rc;
        }
        unsigned long long *ar_cursor = (unsigned long long *) (W.cursor + 2);
        TLM_HIP(hipMemsetAsync(W.cursor, 0, 16, s));

        auto ents_of = [&](int m) { return (m == TLM_METRIC_ES || m == TLM_METRIC_ES_SQRT) ? g->w_ent_es : g->w_ent_os; };
        WalkArgs a;
        std::memset(&a, 0, sizeof(a));
        a.rows = g->w_row;
        a.ents = ents_of(metrics[0]);
        a.flag_bit = 1u << metrics[0];
        a.metric = metrics[0];
        auto mets_of = [&](int m) { return (m == TLM_METRIC_ES || m == TLM_METRIC_ES_SQRT) ? g->w_m_es : g->w_m_os; };
        a.met = mets_of(metrics[0]);
        a.met1 = mets_of(metrics[n_slots - 1]);
        a.d16 = g->w_d16;
        a.ents1 = ents_of(metrics[n_slots - 1]);
        a.flag_bit1 = 1u << metrics[n_slots - 1];
        a.metric1 = metrics[n_slots - 1];
        a.n_units = n_units;
        a.qt = g->w_qt;
        a.node_len = g->node_len;
        a.n_nodes = N;
        a.anchors = g->anchors;
        a.w_anchors = g->w_anchors;
        a.old_of_new = g->w_old_of_new;
        a.n_anchors = (uint32_t) g->h_anchors.size();
        a.det_off = g->det_off;
        a.unit_begin = unit_begin;
        a.rebuild_attempts = is_mc ? p->rebuild_attempts : 1;
        a.backtrack_attempts = is_mc ? p->backtrack_attempts : 0;
        a.len_threshold = is_mc ? p->len_threshold : 0;
        a.cursor = W.cursor;
        a.units = u.units;
        a.n_ent = g->n_adj_pad;
        a.n_work_all = n_all;
        a.unit_ids = u.unit_ids;
        // a unit list: the longest run of units that share a start anchor stands in for "an anchor's trials" below
        uint32_t list_run = 0;
        if (unit_list && is_mc && p->rebuild_attempts > 0) {
            uint32_t run = 0;
            for (uint32_t i = 0; i < n_units; i++) {
                const bool same = i && unit_list[i] / (uint64_t) p->rebuild_attempts == unit_list[i - 1] / (uint64_t) p->rebuild_attempts;
                run = same ? run + 1 : 1;
                list_run = std::max(list_run, run);
            }
        }

        std::vector<uint32_t> work;     // empty = identity
        uint32_t n_work = n_all;
        uint32_t cur_cap = cap, cur_waves = waves;
        for (int round = 0; round < 64; round++) {
            a.n_work = n_work;
            a.work_list = work.empty() ? nullptr : u.work_list;
            // chunk of work items that go to one XCD: the trials of one anchor under one metric, or half of them (MC), a
            // run of first edges (DET); re-run lists are short and scattered.  The waves of a die then walk the same
            // neighbourhood and its L2 holds the rows they share -- which pays where the launch is bound by bytes: chr1-shaped
            // graph (mean list 468 entries, 8.0 TB/s of fabric traffic) 1345 -> 1241 ms with whole anchors, yeast-shaped (365)
            // 94.7 -> 92.1 ms; the A. thaliana-shaped graph (234; bound by instruction issue) is indifferent between a
            // quarter, a half and a whole anchor (738.4 / 737.6 / 742.6 ms).  Round 2 dealt quarters everywhere.
            const double mean_list = g->n_nodes ? (double) g->n_adj / (double) g->n_nodes : 0.0;
            const int per_anchor = unit_list ? (int) list_run : (is_mc ? p->rebuild_attempts : 0);
            a.chunk = !work.empty() ? 8u
                      : is_mc       ? (uint32_t) std::max(1, mean_list >= 300.0 || unit_list ? per_anchor : per_anchor / 2)
                                    : 32u;
            if (const char *e = getenv("TLM_WALK_CHUNK")) a.chunk = (uint32_t) std::max(1, atoi(e));   // tuning experiments
            a.scratch = W.scratch;
            a.cap = W.cap;
            a.bitmaps = W.bitmaps;
            a.bm_words = W.bm_words;
            a.arena = W.arena;
            a.ar_cap = W.ar_cap;
            uint32_t launch_waves = std::min<uint32_t>(cur_waves, ((n_work + WPB - 1) / WPB) * WPB);
            launch_waves = std::max<uint32_t>(WPB, launch_waves);
            TLM_HIP(hipMemsetAsync(W.cursor + WALK_QUEUE_BASE, 0, WALK_QUEUE_STRIDE * WALK_XCDS * 4, s));
            TLM_HIP(hipMemsetAsync(W.cursor + 6, 0xFF, 8, s));      // first workgroup in (atomicMin)
            TLM_HIP(hipMemsetAsync(W.cursor + 8, 0, 96, s));        // last workgroup out (atomicMax); accounting [10..31]
            if (wait_for) {
                // The other lane's launch goes first, and this one must not reach the dispatcher before ALL workgroups of
                // that launch are resident: two launches dispatched together share the slots, and deterministic waves that
                // got in early sit beside the Monte-Carlo waves for the whole launch (+74 ms on the A. thaliana config).
                // The Monte-Carlo kernel reports "every workgroup has started" through a host-mapped word; queued after
                // that, this launch's workgroups start where Monte-Carlo workgroups retire: in the tail.
                while (!wait_for->load()) std::this_thread::yield();
                wait_for = nullptr;
                if (c->resident_host && wait_resident) {
                    const auto t0 = std::chrono::steady_clock::now();
                    while (*(volatile uint32_t *) c->resident_host == 0 && !c->gate_released.load()
                           && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(5))
                        std::this_thread::sleep_for(std::chrono::microseconds(50));
                }
            }
            a.resident_flag = nullptr;
            if (launched && round == 0 && c->resident_host && is_mc) {      // someone waits for this launch to be resident
                *(volatile uint32_t *) c->resident_host = 0;
                a.resident_flag = c->resident_dev;
            }
            TLM_HIP(hipEventRecord(ev0, s));
            const int out_of_bounds = is_mc ? tlm_launch_walk_mc(s, launch_waves / WPB, sq, a, ctx->walk_accounting, lite)
                                            : tlm_launch_walk_det(s, launch_waves / WPB, sq, a, ctx->walk_accounting, lite);
            if (out_of_bounds) return TLM_EINTERNAL;         // checked build only (the message is set)
            TLM_HIP(hipGetLastError());
            TLM_HIP(hipEventRecord(ev1, s));
            if (launched) launched->store(1);
            static const bool tidy_after = [] { const char *e = std::getenv("TLM_TIDY"); return e && !std::strcmp(e, "after"); }();
            if (lane == 0 && round == 0 && !tidy_after) tlm_ctx_start_deferred(ctx);      // the device is busy for a while: the host tidies up
            unsigned long long span[13] = {0};     // first workgroup in, last out; accounting builds: layout bytes, step classes
            TLM_HIP(hipMemcpyAsync(span, W.cursor + 6, sizeof(span), hipMemcpyDeviceToHost, s));
            TLM_HIP(hipMemcpyAsync(h_units, u.units, (size_t) n_all * sizeof(UnitOut), hipMemcpyDeviceToHost, s));
            TLM_HIP(hipEventSynchronize(ev1));
            TLM_HIP(hipStreamSynchronize(s));
            if (lane == 0 && round == 0 && tidy_after) tlm_ctx_start_deferred(ctx);
            float ms = 0;
            TLM_HIP(hipEventElapsedTime(&ms, ev0, ev1));
            cnt[0].kernel_ms += ms;          // the launch is shared by the slots: it is booked on the first one
            cnt[0].kernel_launches++;
            cnt[0].layout_bytes += span[2];
            for (int k = 0; k < TLM_STEP_CLASSES; k++) cnt[0].step_class[k] += span[3 + k];
            // (only the deterministic kernel stamps its span; a Monte-Carlo launch starts at once: its span is `ms`)
            cnt[0].kernel_busy_ms += (!is_mc && span[1] > span[0]) ? (double) (span[1] - span[0]) * 1e-5 : ms;     // 100 MHz ticks
            // overflow handling: re-run the affected units with a full-size path buffer / a larger arena
            std::vector<uint32_t> sc_ovf, ar_ovf;
            for (uint32_t i = 0; i < n_all; i++) {
                const uint32_t st = h_units[i].status;
                if (st == ST_SCRATCH_OVF) sc_ovf.push_back(i);
                else if (st == ST_ARENA_OVF) ar_ovf.push_back(i);
                else TLM_CHECK(st <= ST_NEGATIVE, TLM_EINTERNAL, "walk: unit %u was not run (status %u)", i, st);
            }
            if (sc_ovf.empty() && ar_ovf.empty()) break;
            TLM_CHECK(round < 63, TLM_EINTERNAL, "walk: overflow re-runs did not converge");
            if (!ar_ovf.empty()) {
                uint64_t keep = ~0ull, extra = 0;
                for (uint32_t i : ar_ovf) { keep = std::min<uint64_t>(keep, h_units[i].off); extra += h_units[i].nnodes; }
                // every reservation after the first failed one failed too, so [0, keep) is all valid paths
                if (int rc = ensure_arena(W, s, (keep + extra) + (keep + extra) / 4 + (1u << 20), keep)) return rc;
                unsigned long long k = keep;
                TLM_HIP(hipMemcpyAsync(ar_cursor, &k, 8, hipMemcpyHostToDevice, s));
                TLM_HIP(hipStreamSynchronize(s));
            }
            if (!sc_ovf.empty()) {
                TLM_CHECK(cur_cap < full_cap, TLM_EINTERNAL, "walk: path buffer overflow at full capacity");
                cur_cap = full_cap;
                // bound the re-run's scratch to ~8 GiB
                const uint64_t per_wave = (uint64_t) full_cap * 12u + (uint64_t) bm_words * 4u;
                uint64_t w = (8ull << 30) / per_wave;
                cur_waves = (uint32_t) std::max<uint64_t>(WPB, std::min<uint64_t>(cur_waves, (w / WPB) * WPB));
                if (int rc = ensure_scratch(W, s, cur_waves, cur_cap, bm_words)) return rc;
            }
            work = sc_ovf;
            work.insert(work.end(), ar_ovf.begin(), ar_ovf.end());
            std::sort(work.begin(), work.end());
            n_work = (uint32_t) work.size();
            TLM_HIP(hipMemcpyAsync(u.work_list, work.data(), (size_t) n_work * 4, hipMemcpyHostToDevice, s));
            TLM_HIP(hipStreamSynchronize(s));
        }
    }

    // per slot: pack the accepted paths in unit order, collect the per-unit records
    for (int k = 0; k < n_slots; k++) {
        tlm_pathset *ps = pss[k];
        const UnitOut *du = u.units + (size_t) k * n_units;
        if (n_units) {
            auto cnt_in = rocprim::make_transform_iterator(du, AcceptedCount());
            auto nod_in = rocprim::make_transform_iterator(du, AcceptedNodes());
            size_t tb1 = 0, tb2 = 0;
            TLM_HIP(rocprim::exclusive_scan(nullptr, tb1, cnt_in, u.p_index, 0u, (size_t) n_units, rocprim::plus<uint32_t>(), s));
            TLM_HIP(rocprim::exclusive_scan(nullptr, tb2, nod_in, u.n_off, (uint64_t) 0, (size_t) n_units, rocprim::plus<uint64_t>(), s));
            if (std::max(tb1, tb2) > u.scan_tmp_bytes) {
                tlm_pool_free(u.scan_tmp);
                u.scan_tmp = nullptr;
                u.scan_tmp_bytes = 0;
                if (int rc = tlm_pool_alloc(&u.scan_tmp, std::max(tb1, tb2), s)) return rc;
                u.scan_tmp_bytes = std::max(tb1, tb2);
            }
            TLM_HIP(rocprim::exclusive_scan(u.scan_tmp, tb1, cnt_in, u.p_index, 0u, (size_t) n_units, rocprim::plus<uint32_t>(), s));
            TLM_HIP(rocprim::exclusive_scan(u.scan_tmp, tb2, nod_in, u.n_off, (uint64_t) 0, (size_t) n_units, rocprim::plus<uint64_t>(), s));
            hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(1), 0, s, n_units, du, u.p_index, u.n_off, u.totals);
            uint64_t h_tot[2] = {0, 0};
            TLM_HIP(hipMemcpyAsync(h_tot, u.totals, 16, hipMemcpyDeviceToHost, s));
            TLM_HIP(hipStreamSynchronize(s));
            ps->n_paths = h_tot[0];
            ps->n_nodes_total = h_tot[1];
        }
        if (int rc = pmalloc(&ps->unit, ps->n_paths, s)) return rc;
        if (int rc = pmalloc(&ps->length, ps->n_paths, s)) return rc;
        if (int rc = pmalloc(&ps->off, ps->n_paths + 1, s)) return rc;
        if (int rc = pmalloc(&ps->nodes, ps->n_nodes_total, s)) return rc;
        if (int rc = pmalloc(&ps->slots, ps->n_nodes_total, s)) return rc;
        if (n_units) {
            uint32_t blocks = std::min<uint32_t>((n_units + 3) / 4, 256u * 8u);
            hipLaunchKernelGGL(pack_paths_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, n_units, unit_begin, u.unit_ids, du,
                               u.p_index, u.n_off, W.arena, ps->unit, ps->length, ps->off, ps->nodes, ps->slots);
            TLM_HIP(hipGetLastError());
        }
        TLM_HIP(hipMemcpyAsync(ps->off + ps->n_paths, &ps->n_nodes_total, 8, hipMemcpyHostToDevice, s));
        TLM_HIP(hipStreamSynchronize(s));      // p_index / n_off are reused by the next slot
        ps->stats.resize(n_units);
        for (uint32_t i = 0; i < n_units; i++) {
            tlm_unit_stat &st = ps->stats[i];
            const UnitOut &o = h_units[(size_t) k * n_units + i];
            st.status = (uint8_t) o.status;
            st.n_nodes = o.nnodes;
            st.steps = o.steps;
            st.evals = o.evals;
            st.draws = o.draws;
            st.length = o.len;
            cnt[k].steps += o.steps;
            cnt[k].evals += o.evals;
            cnt[k].draws += o.draws;
        }
        cnt[k].units = n_units;
        cnt[k].accepted = ps->n_paths;
    }
    for (int k = 0; k < n_slots; k++) {
        if (ctr) ctr[k] = cnt[k];
        out[k] = pss[k];
    }
    guard.p = nullptr;
    dirty.armed = false;
    return TLM_OK;
}

}  // namespace

extern "C" int tlm_walk_mc(tlm_ctx *ctx, const tlm_graph *g, int metric, const tlm_walk_params *p, uint64_t unit_begin,
                           uint64_t unit_end, tlm_pathset **out, tlm_counters *ctr) {
    TLM_CHECK(p, TLM_EINVAL, "tlm_walk_mc: null parameters");
    return run_walks(ctx, g, true, 1, &metric, p, unit_begin, unit_end, out, ctr);
}

extern "C" int tlm_walk_det(tlm_ctx *ctx, const tlm_graph *g, int metric, uint64_t unit_begin, uint64_t unit_end,
                            tlm_pathset **out, tlm_counters *ctr) {
    return run_walks(ctx, g, false, 1, &metric, nullptr, unit_begin, unit_end, out, ctr);
}

extern "C" int tlm_walk_mc2(tlm_ctx *ctx, const tlm_graph *g, int metric_a, int metric_b, const tlm_walk_params *p,
                            uint64_t unit_begin, uint64_t unit_end, tlm_pathset **out2, tlm_counters *ctr2) {
    TLM_CHECK(p, TLM_EINVAL, "tlm_walk_mc2: null parameters");
    const int m[2] = {metric_a, metric_b};
    return run_walks(ctx, g, true, 2, m, p, unit_begin, unit_end, out2, ctr2);
}

extern "C" int tlm_walk_det2(tlm_ctx *ctx, const tlm_graph *g, int metric_a, int metric_b, uint64_t unit_begin,
                             uint64_t unit_end, tlm_pathset **out2, tlm_counters *ctr2) {
    const int m[2] = {metric_a, metric_b};
    return run_walks(ctx, g, false, 2, m, nullptr, unit_begin, unit_end, out2, ctr2);
}

/* The reference's four builder calls (Main.cpp:237-243) as one call.  The deterministic walks are few and long: their
 * launch lasts as long as its longest walk while most of the chip idles, and the Monte-Carlo launch ends in a tail in
 * which ever fewer waves are still walking.  Here the deterministic launch goes to a second, low-priority stream right
 * after the Monte-Carlo launch: its workgroups start wherever a Monte-Carlo workgroup has finished, i.e. in that tail.
 * out4 / ctr4: MC metric_a, MC metric_b, DET metric_a, DET metric_b -- the Path sets of the four separate calls. */
extern "C" int tlm_walk_builders(tlm_ctx *ctx, const tlm_graph *g, int metric_a, int metric_b, const tlm_walk_params *p,
                                 uint64_t mc_begin, uint64_t mc_end, uint64_t det_begin, uint64_t det_end,
                                 tlm_pathset **out4, tlm_counters *ctr4) {
    TLM_CHECK(ctx && g && out4 && p, TLM_EINVAL, "tlm_walk_builders: null argument");
    for (int k = 0; k < 4; k++) out4[k] = nullptr;
    const int m[2] = {metric_a, metric_b};
    tlm_counters cd[2], cm[2];
    std::memset(cd, 0, sizeof(cd));
    std::memset(cm, 0, sizeof(cm));
    std::atomic<int> mc_launched{0}, det_launched{0};
    int rc_det = TLM_OK;
    char det_err[512] = "";
    // Which launch goes first.  "tail" (default): Monte-Carlo first, the deterministic launch once every Monte-Carlo
    // workgroup is resident -- its workgroups start in the Monte-Carlo launch's tail.  "det_first": the deterministic launch
    // first -- most of its units are short and its waves leave within milliseconds, the few long walks (no length cap:
    // they bound the launch) then run BESIDE the Monte-Carlo launch instead of after it.
    static const bool det_first = [] { const char *e = std::getenv("TLM_BUILDERS_ORDER"); return e && !std::strcmp(e, "det_first"); }();
    std::thread side([&] {
        rc_det = det_first ? run_walks(ctx, g, false, 2, m, nullptr, det_begin, det_end, out4 + 2, cd, 1, &det_launched, nullptr)
                           : run_walks(ctx, g, false, 2, m, nullptr, det_begin, det_end, out4 + 2, cd, 1, nullptr, &mc_launched);
        if (rc_det != TLM_OK) std::snprintf(det_err, sizeof(det_err), "%s", tlm_last_error());
    });
    const int rc_mc = det_first ? run_walks(ctx, g, true, 2, m, p, mc_begin, mc_end, out4, cm, 0, nullptr, &det_launched, false)
                                : run_walks(ctx, g, true, 2, m, p, mc_begin, mc_end, out4, cm, 0, &mc_launched);
    side.join();
    if (rc_mc != TLM_OK || rc_det != TLM_OK) {
        if (rc_mc == TLM_OK) tlm_set_error("%s", det_err);
        for (int k = 0; k < 4; k++) {
            tlm_pathset_destroy(out4[k]);
            out4[k] = nullptr;
        }
        return rc_mc != TLM_OK ? rc_mc : rc_det;
    }
    if (ctr4) {
        ctr4[0] = cm[0]; ctr4[1] = cm[1]; ctr4[2] = cd[0]; ctr4[3] = cd[1];
    }
    return TLM_OK;
}

/* (ABI 4) The same calls over a LIST of units (host array, strictly ascending) instead of a range: what a rank of a sliced
 * partition walks (tlm_dist_unit_share).  Results: the Path set of exactly those units, ordered by unit; unit_stats in
 * list order. */
extern "C" int tlm_walk_list(tlm_ctx *ctx, const tlm_graph *g, int deterministic, int metric_a, int metric_b,
                             const tlm_walk_params *p, uint64_t n_units, const uint64_t *units, tlm_pathset **out2,
                             tlm_counters *ctr2) {
    TLM_CHECK(deterministic || p, TLM_EINVAL, "tlm_walk_list: null parameters");
    TLM_CHECK(units || !n_units, TLM_EINVAL, "tlm_walk_list: null unit list");
    const int m[2] = {metric_a, metric_b};
    static const uint64_t none = 0;
    return run_walks(ctx, g, !deterministic, metric_b < 0 ? 1 : 2, m, deterministic ? nullptr : p, 0, n_units, out2, ctr2, 0, nullptr,
                     nullptr, true, units ? units : &none);
}

extern "C" int tlm_walk_builders_list(tlm_ctx *ctx, const tlm_graph *g, int metric_a, int metric_b, const tlm_walk_params *p,
                                      uint64_t n_mc, const uint64_t *mc_units, uint64_t n_det, const uint64_t *det_units,
                                      tlm_pathset **out4, tlm_counters *ctr4) {
    TLM_CHECK(ctx && g && out4 && p && (mc_units || !n_mc) && (det_units || !n_det), TLM_EINVAL, "tlm_walk_builders_list: null argument");
    for (int k = 0; k < 4; k++) out4[k] = nullptr;
    static const uint64_t none = 0;
    if (!mc_units) mc_units = &none;
    if (!det_units) det_units = &none;
    const int m[2] = {metric_a, metric_b};
    tlm_counters cd[2], cm[2];
    std::memset(cd, 0, sizeof(cd));
    std::memset(cm, 0, sizeof(cm));
    std::atomic<int> mc_launched{0};
    int rc_det = TLM_OK;
    char det_err[512] = "";
    std::thread side([&] {          // as tlm_walk_builders: the deterministic launch starts in the Monte-Carlo launch's tail
        rc_det = run_walks(ctx, g, false, 2, m, nullptr, 0, n_det, out4 + 2, cd, 1, nullptr, &mc_launched, true, det_units);
        if (rc_det != TLM_OK) std::snprintf(det_err, sizeof(det_err), "%s", tlm_last_error());
    });
    const int rc_mc = run_walks(ctx, g, true, 2, m, p, 0, n_mc, out4, cm, 0, &mc_launched, nullptr, true, mc_units);
    side.join();
    if (rc_mc != TLM_OK || rc_det != TLM_OK) {
        if (rc_mc == TLM_OK) tlm_set_error("%s", det_err);
        for (int k = 0; k < 4; k++) {
            tlm_pathset_destroy(out4[k]);
            out4[k] = nullptr;
        }
        return rc_mc != TLM_OK ? rc_mc : rc_det;
    }
    if (ctr4) {
        ctr4[0] = cm[0]; ctr4[1] = cm[1]; ctr4[2] = cd[0]; ctr4[3] = cd[1];
    }
    return TLM_OK;
}

// ---------------------------------------------------------------------------------------------
// path sets

extern "C" int tlm_pathset_sizes(const tlm_pathset *ps, uint64_t *n_paths, uint64_t *n_nodes_total, uint64_t *n_units) {
    TLM_CHECK(ps, TLM_EINVAL, "tlm_pathset_sizes: null path set");
    if (n_paths) *n_paths = ps->n_paths;
    if (n_nodes_total) *n_nodes_total = ps->n_nodes_total;
    if (n_units) *n_units = ps->n_units;
    return TLM_OK;
}

extern "C" int tlm_pathset_export(const tlm_pathset *ps, int dst_is_device, uint64_t *unit, int64_t *length,
                                  uint64_t *off, uint32_t *nodes, uint32_t *slots) {
    TLM_CHECK(ps, TLM_EINVAL, "tlm_pathset_export: null path set");
    TLM_HIP(hipSetDevice(ps->device));
    hipStream_t s = ps->ctx->stream;
    const hipMemcpyKind k = dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (unit && ps->n_paths) TLM_HIP(hipMemcpyAsync(unit, ps->unit, ps->n_paths * 8, k, s));
    if (length && ps->n_paths) TLM_HIP(hipMemcpyAsync(length, ps->length, ps->n_paths * 8, k, s));
    if (off) TLM_HIP(hipMemcpyAsync(off, ps->off, (ps->n_paths + 1) * 8, k, s));
    if (nodes && ps->n_nodes_total) TLM_HIP(hipMemcpyAsync(nodes, ps->nodes, ps->n_nodes_total * 4, k, s));
    if (slots && ps->n_nodes_total) TLM_HIP(hipMemcpyAsync(slots, ps->slots, ps->n_nodes_total * 4, k, s));
    TLM_HIP(hipStreamSynchronize(s));
    return TLM_OK;
}

extern "C" int tlm_pathset_unit_stats(const tlm_pathset *ps, tlm_unit_stat *stats_host) {
    TLM_CHECK(ps && stats_host, TLM_EINVAL, "tlm_pathset_unit_stats: null argument");
    if (!ps->stats.empty()) std::memcpy(stats_host, ps->stats.data(), ps->stats.size() * sizeof(tlm_unit_stat));
    return TLM_OK;
}

extern "C" void tlm_pathset_destroy(tlm_pathset *ps) {
    if (!ps) return;
    (void) hipSetDevice(ps->device);
    // hipFree used to wait for the device; the pool does not, so make sure no kernel is still reading the set (a
    // no-op when the device is idle, which it is whenever a caller lets go of a set after using it)
    (void) hipDeviceSynchronize();
    tlm_pool_free(ps->unit, true); tlm_pool_free(ps->length, true); tlm_pool_free(ps->off, true);
    tlm_pool_free(ps->nodes, true); tlm_pool_free(ps->slots, true);
    delete ps;
}

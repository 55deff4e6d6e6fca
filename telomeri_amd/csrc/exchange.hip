// telomeri_amd/csrc/exchange.hip -- the ONE exchange of the multi-GPU path build.
//
// One process per GPU; rank r walks the units of its contiguous anchor bundle on a replicated graph; no collective
// while walking.  Afterwards every rank needs paths_ as ONE PathManager would hold it after the reference's builder
// calls (Main.cpp:237-243): builder by builder, inside a builder in unit order = rank order.  tlm_pathset_allgather does
// that in a single grouped RCCL exchange over the packed device buffers:
//   1. sizes: one ncclAllGather of (paths, nodes) per builder set + a "slots need 32 bits" flag;
//   2. every block (rank r, builder b) has a fixed place in the result; each rank ncclSend's its blocks to every peer
//      and ncclRecv's every peer's blocks STRAIGHT INTO those places -- all inside one ncclGroupStart/End, so on the
//      xGMI full mesh every pair of GPUs talks over its own link at once (no ring, no staging copy, no padding);
//   3. one fix-up kernel rebases the received `off` prefixes, tags `unit` with the builder and widens the slots.
// On the wire a path node costs 6 bytes (u32 node + u16 edge slot; a slot is a position inside a neighbour list, and
// only graphs with lists of 65 535 entries or more need 32 bits) instead of the 8 it occupies in HBM.
// RCCL is loaded on first use (dlopen of librccl.so.1, the copy the process already holds if it has one): the library
// itself does not depend on it, and single-GPU hosts never load its half gigabyte.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>

#include <rccl/rccl.h>

#include "tlm_common.h"

namespace {

struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
};

RcclApi *rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.handle ? &api : nullptr;
    tried = true;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);         // the copy already in the process (PyTorch's)
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        tlm_set_error("RCCL is not available: %s", dlerror());
        return nullptr;
    }
    bool ok = true;
    auto sym = [&](const char *name) {
        void *p = dlsym(h, name);
        if (!p) ok = false;
        return p;
    };
    api.GetUniqueId = (decltype(api.GetUniqueId)) sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank)) sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy)) sym("ncclCommDestroy");
    api.GetErrorString = (decltype(api.GetErrorString)) sym("ncclGetErrorString");
    api.AllGather = (decltype(api.AllGather)) sym("ncclAllGather");
    api.Send = (decltype(api.Send)) sym("ncclSend");
    api.Recv = (decltype(api.Recv)) sym("ncclRecv");
    api.GroupStart = (decltype(api.GroupStart)) sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd)) sym("ncclGroupEnd");
    if (!ok) {
        tlm_set_error("librccl.so.1 lacks an expected symbol");
        return nullptr;
    }
    api.handle = h;
    return &api;
}

#define TLM_NCCL(api, call)                                                                     \
    do {                                                                                        \
        ncclResult_t r_ = (call);                                                               \
        if (r_ != ncclSuccess) {                                                                \
            tlm_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, (api)->GetErrorString(r_)); \
            return TLM_EHIP;                                                                    \
        }                                                                                       \
    } while (0)

struct Tmp {
    void *p = nullptr;
    ~Tmp() { tlm_pool_free(p); }
};

// slots -> 16 bits (0xFFFF = the end-of-path sentinel 0xFFFFFFFF); *wide is set if a real slot does not fit
__global__ void narrow_slots_kernel(uint64_t n, const uint32_t *__restrict__ slots, uint16_t *__restrict__ out,
                                    uint32_t *__restrict__ wide) {
    bool w = false;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint32_t s = slots[i];
        w |= s != 0xFFFFFFFFu && s >= 0xFFFFu;
        out[i] = (uint16_t) (s == 0xFFFFFFFFu ? 0xFFFFu : s);
    }
    if (w) atomicOr(wide, 1u);
}

// One block of the result = one builder set of one rank.
struct BlockTable {
    uint32_t n_blocks;
    uint64_t path_begin[TLM_EXCHANGE_MAX_BLOCKS + 1];     // in result order; the last entry is the total
    uint64_t node_begin[TLM_EXCHANGE_MAX_BLOCKS + 1];
    uint8_t builder[TLM_EXCHANGE_MAX_BLOCKS];
};

__device__ inline uint32_t block_of(const uint64_t *begin, uint32_t n_blocks, uint64_t i) {
    uint32_t lo = 0, hi = n_blocks;       // largest k with begin[k] <= i
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (begin[mid] <= i) lo = mid;
        else hi = mid;
    }
    return lo;
}

// received `off` values are relative to their own set: rebase; tag `unit` with the builder as tlm_pathset_concat does
__global__ void fixup_paths_kernel(BlockTable t, uint64_t *__restrict__ unit, uint64_t *__restrict__ off) {
    const uint64_t n = t.path_begin[t.n_blocks];
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint32_t k = block_of(t.path_begin, t.n_blocks, i);
        unit[i] = (unit[i] & 0x00FFFFFFFFFFFFFFull) | ((uint64_t) t.builder[k] << 56);
        off[i] += t.node_begin[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) off[n] = t.node_begin[t.n_blocks];
}

__global__ void widen_slots_kernel(uint64_t n, const uint16_t *__restrict__ in, uint32_t *__restrict__ out) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint16_t s = in[i];
        out[i] = s == 0xFFFFu ? 0xFFFFFFFFu : (uint32_t) s;
    }
}

unsigned grid_of(uint64_t n) { return (unsigned) std::max<uint64_t>(1, std::min<uint64_t>((n + 255) / 256, 4096)); }

}   // namespace

// Where block (rank r, builder b) starts in the gathered set: builder-major, inside a builder in rank order.
// counts[(r * n_sets + b) * 2 + {0, 1}] = paths, nodes of that block; path_base / node_base get world * n_sets entries in
// the same (r, b) indexing; totals[2] = paths, nodes of the result.  Pure host arithmetic (tests check it on the CPU).
extern "C" int tlm_exchange_layout(int world, int n_sets, const uint64_t *counts, uint64_t *path_base, uint64_t *node_base,
                                   uint64_t *totals) {
    TLM_CHECK(world >= 1 && n_sets >= 1 && counts && path_base && node_base && totals, TLM_EINVAL,
              "tlm_exchange_layout: bad argument");
    uint64_t p = 0, n = 0;
    for (int b = 0; b < n_sets; b++)
        for (int r = 0; r < world; r++) {
            const size_t k = (size_t) r * n_sets + b;
            path_base[k] = p;
            node_base[k] = n;
            p += counts[2 * k];
            n += counts[2 * k + 1];
        }
    totals[0] = p;
    totals[1] = n;
    return TLM_OK;
}

extern "C" int tlm_dist_unique_id(uint8_t *id_out) {
    TLM_CHECK(id_out, TLM_EINVAL, "tlm_dist_unique_id: null argument");
    RcclApi *api = rccl_api();
    if (!api) return TLM_EHIP;
    static_assert(sizeof(ncclUniqueId) == TLM_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    TLM_NCCL(api, api->GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return TLM_OK;
}

extern "C" int tlm_dist_init(tlm_ctx *ctx, int rank, int world, const uint8_t *id_bytes) {
    TLM_CHECK(ctx && id_bytes, TLM_EINVAL, "tlm_dist_init: null argument");
    TLM_CHECK(world >= 1 && rank >= 0 && rank < world, TLM_EINVAL, "tlm_dist_init: rank %d of %d", rank, world);
    TLM_CHECK(world * 4 <= TLM_EXCHANGE_MAX_BLOCKS, TLM_ELIMIT, "tlm_dist_init: at most %d ranks", TLM_EXCHANGE_MAX_BLOCKS / 4);
    TLM_CHECK(!ctx->comm, TLM_EINVAL, "tlm_dist_init: the context already has a communicator");
    RcclApi *api = rccl_api();
    if (!api) return TLM_EHIP;
    TLM_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t comm = nullptr;
    TLM_NCCL(api, api->CommInitRank(&comm, world, id, rank));
    ctx->comm = comm;
    ctx->rank = rank;
    ctx->world = world;
    return TLM_OK;
}

extern "C" void tlm_dist_finalize(tlm_ctx *ctx) {
    if (!ctx || !ctx->comm) return;
    RcclApi *api = rccl_api();
    (void) hipSetDevice(ctx->device);
    (void) hipStreamSynchronize(ctx->stream);
    if (api) (void) api->CommDestroy((ncclComm_t) ctx->comm);
    ctx->comm = nullptr;
    ctx->rank = 0;
    ctx->world = 1;
}

extern "C" int tlm_dist_rank(const tlm_ctx *ctx) { return ctx ? ctx->rank : 0; }
extern "C" int tlm_dist_world(const tlm_ctx *ctx) { return ctx ? ctx->world : 1; }

extern "C" int tlm_dist_anchor_range(const tlm_graph *g, int rank, int world, uint32_t *begin, uint32_t *end) {
    TLM_CHECK(g && begin && end && world >= 1 && rank >= 0 && rank < world, TLM_EINVAL, "tlm_dist_anchor_range: bad argument");
    const uint32_t na = (uint32_t) g->h_anchors.size();
    const uint32_t per = (na + (uint32_t) world - 1) / (uint32_t) world;      // contiguous bundles of ceil(na / world)
    *begin = std::min(na, per * (uint32_t) rank);
    *end = std::min(na, per * (uint32_t) rank + per);
    return TLM_OK;
}

extern "C" int tlm_pathset_allgather(tlm_ctx *ctx, int n_sets, tlm_pathset *const *sets, tlm_pathset **out,
                                     tlm_exchange_info *info) {
    TLM_CHECK(ctx && sets && out && n_sets >= 1 && n_sets <= 8, TLM_EINVAL, "tlm_pathset_allgather: bad argument");
    for (int b = 0; b < n_sets; b++)
        TLM_CHECK(sets[b] && sets[b]->ctx == ctx, TLM_EINVAL, "tlm_pathset_allgather: set %d is null or of another context", b);
    const int world = ctx->world, rank = ctx->rank;
    TLM_CHECK(world * n_sets <= TLM_EXCHANGE_MAX_BLOCKS, TLM_ELIMIT, "tlm_pathset_allgather: %d blocks", world * n_sets);
    if (info) std::memset(info, 0, sizeof(*info));
    const auto t_wall = std::chrono::steady_clock::now();
    // (TLM_EXCHANGE_SELFTEST: a one-rank communicator goes through the whole exchange -- sizes collective, placement,
    // fix-up -- so that a 1-GPU box exercises everything but the peer transfers)
    if (world == 1 && !(ctx->comm && getenv("TLM_EXCHANGE_SELFTEST"))) {       // nothing to exchange: paths_ of one PathManager
        const int rc = tlm_pathset_concat(ctx, n_sets, sets, out);
        if (info) info->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_wall).count();
        return rc;
    }
    RcclApi *api = rccl_api();
    if (!api) return TLM_EHIP;
    TLM_CHECK(ctx->comm, TLM_EINVAL, "tlm_pathset_allgather: tlm_dist_init has not been called");
    ncclComm_t comm = (ncclComm_t) ctx->comm;
    TLM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    TLM_HIP(hipEventRecord(ctx->ev0, s));

    // ---- 0. my slots in 16 bits
    uint64_t my_nodes = 0;
    for (int b = 0; b < n_sets; b++) my_nodes += sets[b]->n_nodes_total;
    Tmp s16, wide_flag;
    if (int rc = tlm_pool_alloc(&s16.p, (my_nodes ? my_nodes : 1) * 2, s)) return rc;
    if (int rc = tlm_pool_alloc(&wide_flag.p, 4, s)) return rc;
    TLM_HIP(hipMemsetAsync(wide_flag.p, 0, 4, s));
    {
        uint64_t at = 0;
        for (int b = 0; b < n_sets; b++) {
            const uint64_t n = sets[b]->n_nodes_total;
            if (n) hipLaunchKernelGGL(narrow_slots_kernel, dim3(grid_of(n)), dim3(256), 0, s, n, sets[b]->slots,
                                      (uint16_t *) s16.p + at, (uint32_t *) wide_flag.p);
            at += n;
        }
    }
    uint32_t my_wide = 0;
    TLM_HIP(hipMemcpyAsync(&my_wide, wide_flag.p, 4, hipMemcpyDeviceToHost, s));
    TLM_HIP(hipStreamSynchronize(s));

    // ---- 1. sizes
    const size_t per_rank = (size_t) n_sets * 2 + 1;
    std::vector<uint64_t> mine(per_rank), all(per_rank * world);
    for (int b = 0; b < n_sets; b++) {
        mine[2 * b] = sets[b]->n_paths;
        mine[2 * b + 1] = sets[b]->n_nodes_total;
    }
    mine[2 * n_sets] = my_wide;
    Tmp d_mine, d_all;
    if (int rc = tlm_pool_alloc(&d_mine.p, per_rank * 8, s)) return rc;
    if (int rc = tlm_pool_alloc(&d_all.p, per_rank * 8 * world, s)) return rc;
    TLM_HIP(hipMemcpyAsync(d_mine.p, mine.data(), per_rank * 8, hipMemcpyHostToDevice, s));
    TLM_NCCL(api, api->AllGather(d_mine.p, d_all.p, per_rank, ncclUint64, comm, s));
    TLM_HIP(hipMemcpyAsync(all.data(), d_all.p, per_rank * 8 * world, hipMemcpyDeviceToHost, s));
    TLM_HIP(hipStreamSynchronize(s));
    bool wide = false;
    std::vector<uint64_t> counts((size_t) world * n_sets * 2), path_base((size_t) world * n_sets), node_base((size_t) world * n_sets);
    for (int r = 0; r < world; r++) {
        for (int k = 0; k < 2 * n_sets; k++) counts[(size_t) r * n_sets * 2 + k] = all[r * per_rank + k];
        wide |= all[r * per_rank + 2 * n_sets] != 0;
    }
    uint64_t totals[2];
    tlm_exchange_layout(world, n_sets, counts.data(), path_base.data(), node_base.data(), totals);

    // ---- 2. the result, and every block straight into its place
    tlm_pathset *ps = new tlm_pathset();
    ps->ctx = ctx;
    ps->device = ctx->device;
    ps->n_paths = totals[0];
    ps->n_nodes_total = totals[1];
    struct Guard {
        tlm_pathset *p;
        ~Guard() { if (p) tlm_pathset_destroy(p); }
    } guard{ps};
    auto palloc = [&](void **p, size_t bytes) { return tlm_pool_alloc(p, bytes ? bytes : 8, s); };
    if (int rc = palloc((void **) &ps->unit, totals[0] * 8)) return rc;
    if (int rc = palloc((void **) &ps->length, totals[0] * 8)) return rc;
    if (int rc = palloc((void **) &ps->off, (totals[0] + 1) * 8)) return rc;
    if (int rc = palloc((void **) &ps->nodes, totals[1] * 4)) return rc;
    if (int rc = palloc((void **) &ps->slots, totals[1] * 4)) return rc;
    Tmp r16;            // received 16-bit slots, result layout
    if (!wide)
        if (int rc = tlm_pool_alloc(&r16.p, (totals[1] ? totals[1] : 1) * 2, s)) return rc;
    const size_t slot_bytes = wide ? 4 : 2;
    uint64_t sent = 0, received = 0;
    TLM_NCCL(api, api->GroupStart());
    uint64_t my_node_at = 0;
    for (int b = 0; b < n_sets; b++) {
        const tlm_pathset *q = sets[b];
        const void *my_slots = wide ? (const void *) q->slots : (const void *) ((const uint16_t *) s16.p + my_node_at);
        for (int r = 0; r < world; r++) {
            const size_t k = (size_t) r * n_sets + b;
            const uint64_t np = counts[2 * k], nn = counts[2 * k + 1], pb = path_base[k], nb = node_base[k];
            char *dst_slots = wide ? (char *) (ps->slots + nb) : (char *) ((uint16_t *) r16.p + nb);
            if (r == rank) {
                if (np) {
                    TLM_HIP(hipMemcpyAsync(ps->unit + pb, q->unit, np * 8, hipMemcpyDeviceToDevice, s));
                    TLM_HIP(hipMemcpyAsync(ps->length + pb, q->length, np * 8, hipMemcpyDeviceToDevice, s));
                    TLM_HIP(hipMemcpyAsync(ps->off + pb, q->off, np * 8, hipMemcpyDeviceToDevice, s));
                }
                if (nn) {
                    TLM_HIP(hipMemcpyAsync(ps->nodes + nb, q->nodes, nn * 4, hipMemcpyDeviceToDevice, s));
                    TLM_HIP(hipMemcpyAsync(dst_slots, my_slots, nn * slot_bytes, hipMemcpyDeviceToDevice, s));
                }
                continue;
            }
            // to peer r: my block b; from peer r: its block b.  Both sides skip empty pieces (they know the sizes).
            const uint64_t mp = q->n_paths, mn = q->n_nodes_total;
            if (mp) {
                TLM_NCCL(api, api->Send(q->unit, mp * 8, ncclUint8, r, comm, s));
                TLM_NCCL(api, api->Send(q->length, mp * 8, ncclUint8, r, comm, s));
                TLM_NCCL(api, api->Send(q->off, mp * 8, ncclUint8, r, comm, s));
                sent += mp * 24;
            }
            if (mn) {
                TLM_NCCL(api, api->Send(q->nodes, mn * 4, ncclUint8, r, comm, s));
                TLM_NCCL(api, api->Send(my_slots, mn * slot_bytes, ncclUint8, r, comm, s));
                sent += mn * (4 + slot_bytes);
            }
            if (np) {
                TLM_NCCL(api, api->Recv(ps->unit + pb, np * 8, ncclUint8, r, comm, s));
                TLM_NCCL(api, api->Recv(ps->length + pb, np * 8, ncclUint8, r, comm, s));
                TLM_NCCL(api, api->Recv(ps->off + pb, np * 8, ncclUint8, r, comm, s));
                received += np * 24;
            }
            if (nn) {
                TLM_NCCL(api, api->Recv(ps->nodes + nb, nn * 4, ncclUint8, r, comm, s));
                TLM_NCCL(api, api->Recv(dst_slots, nn * slot_bytes, ncclUint8, r, comm, s));
                received += nn * (4 + slot_bytes);
            }
        }
        my_node_at += q->n_nodes_total;
    }
    TLM_NCCL(api, api->GroupEnd());

    // ---- 3. fix-up
    BlockTable t;
    std::memset(&t, 0, sizeof(t));
    t.n_blocks = (uint32_t) (world * n_sets);
    {
        uint32_t j = 0;
        for (int b = 0; b < n_sets; b++)
            for (int r = 0; r < world; r++, j++) {
                const size_t k = (size_t) r * n_sets + b;
                t.path_begin[j] = path_base[k];
                t.node_begin[j] = node_base[k];
                t.builder[j] = (uint8_t) b;
            }
        t.path_begin[j] = totals[0];
        t.node_begin[j] = totals[1];
    }
    hipLaunchKernelGGL(fixup_paths_kernel, dim3(grid_of(totals[0])), dim3(256), 0, s, t, ps->unit, ps->off);
    if (!wide && totals[1])
        hipLaunchKernelGGL(widen_slots_kernel, dim3(grid_of(totals[1])), dim3(256), 0, s, totals[1], (const uint16_t *) r16.p,
                           ps->slots);
    TLM_HIP(hipEventRecord(ctx->ev1, s));
    TLM_HIP(hipStreamSynchronize(s));
    TLM_HIP(hipGetLastError());
    if (info) {
        float ms = 0;
        (void) hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
        info->device_ms = ms;
        info->bytes_sent = sent;
        info->bytes_received = received;
        info->slot_bits = wide ? 32 : 16;
        info->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_wall).count();
    }
    guard.p = nullptr;
    *out = ps;
    return TLM_OK;
}

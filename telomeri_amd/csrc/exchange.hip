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
// itself does not depend on it, and single-GPU hosts never load its half gigabyte.  TLM_RCCL_LIB names another library
// with the same entry points (tests/fake_rccl: several ranks on ONE GPU, which RCCL refuses).
//
// FAILING TOGETHER.  Every exchange call is a sequence of collectives; a rank that leaves it alone leaves its peers
// waiting in RCCL for ever.  So (i) whatever can fail locally -- allocations, local kernels -- happens in stages whose
// outcome the ranks SWAP (agree(): a 4-byte-per-rank ncclAllGather through a buffer the context owns since tlm_dist_init)
// before the next data collective: one failure and everybody returns, the failing rank its own code, the others
// TLM_EPEER; (ii) the stretches in which peers wait for this rank are bracketed by a Collective guard: any early return
// inside closes an open group (ncclGroupEnd) and aborts the communicator (ncclCommAbort), after which the context is
// out of the job (comm_broken) and the launcher ends the other ranks.
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
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;      // optional
};

RcclApi *rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.handle ? &api : nullptr;
    tried = true;
    void *h = nullptr;
    if (const char *named = getenv("TLM_RCCL_LIB")) {                 // an explicit choice is never silently replaced
        h = dlopen(named, RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            tlm_set_error("TLM_RCCL_LIB=%s cannot be loaded: %s", named, dlerror());
            return nullptr;
        }
    }
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);       // the copy already in the process (PyTorch's)
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
    api.Broadcast = (decltype(api.Broadcast)) sym("ncclBroadcast");
    if (!ok) {
        tlm_set_error("the RCCL library lacks an expected symbol");
        return nullptr;
    }
    api.CommAbort = (decltype(api.CommAbort)) dlsym(h, "ncclCommAbort");
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

// the communicator is given up: peers that still wait for this rank are ended by the launcher
void break_comm(tlm_ctx *ctx, RcclApi *api) {
    if (!ctx->comm) return;
    if (api && api->CommAbort) (void) api->CommAbort((ncclComm_t) ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_broken = true;
}

// A stretch of an exchange call in which peers wait for this rank.  Leaving it early (the TLM_HIP / TLM_NCCL macros
// return) closes an open group and gives the communicator up; done() marks the regular end.
struct Collective {
    tlm_ctx *ctx;
    RcclApi *api;
    bool group_open = false, finished = false;
    int group_start() {
        TLM_NCCL(api, api->GroupStart());
        group_open = true;
        return TLM_OK;
    }
    int group_end() {
        group_open = false;
        TLM_NCCL(api, api->GroupEnd());
        return TLM_OK;
    }
    void done() { finished = true; }
    ~Collective() {
        if (finished) return;
        std::string why = tlm_last_error();         // keep the message of what went wrong, not of the tidying
        if (group_open) (void) api->GroupEnd();
        break_comm(ctx, api);
        tlm_set_error("%s [the communicator has been aborted]", why.c_str());
    }
};

// test hook, CHECKED BUILD ONLY (make EXTRA=-DTLM_CHECKED -> telomeri_amd/lib_checked; the product library has no such
// switch): TLM_EXCHANGE_FAULT="<rank>:<point>" makes that rank behave as if a local step had failed at <point> (prep, alloc,
// group, fetch); tests/test_exchange_world.py checks that every rank then fails, none hangs
bool fault_here(const tlm_ctx *ctx, const char *point) {
#ifndef TLM_CHECKED
    (void) ctx;
    (void) point;
    return false;
#else
    const char *f = getenv("TLM_EXCHANGE_FAULT");
    if (!f) return false;
    char *colon = nullptr;
    const long r = std::strtol(f, &colon, 10);
    if (!colon || *colon != ':' || r != ctx->rank || std::strcmp(colon + 1, point) != 0) return false;
    tlm_set_error("injected failure at '%s' on rank %d (TLM_EXCHANGE_FAULT)", point, ctx->rank);
    return true;
#endif
}

// Swap the ranks' local status; TLM_OK only if every rank is fine.  With a failure somewhere every rank returns: the
// failing one(s) their own code, the others TLM_EPEER.  Scratch words [0] = mine, [64 .. 64 + world) = everybody's.
int agree(tlm_ctx *ctx, RcclApi *api, int local_rc, const char *where) {
    std::string my_error = local_rc ? tlm_last_error() : "";
    hipStream_t s = ctx->stream;
    int32_t *d = (int32_t *) ctx->xchg_scratch;
    int32_t mine = local_rc, all[TLM_EXCHANGE_MAX_BLOCKS] = {0};
    {
        Collective col{ctx, api};
        TLM_HIP(hipMemcpyAsync(d, &mine, 4, hipMemcpyHostToDevice, s));
        TLM_NCCL(api, api->AllGather(d, d + 64, 1, ncclInt32, (ncclComm_t) ctx->comm, s));
        TLM_HIP(hipMemcpyAsync(all, d + 64, 4 * (size_t) ctx->world, hipMemcpyDeviceToHost, s));
        TLM_HIP(hipStreamSynchronize(s));
        col.done();
    }
    if (local_rc) {
        tlm_set_error("%s", my_error.c_str());
        return local_rc;
    }
    for (int r = 0; r < ctx->world; r++)
        if (all[r] != 0) {
            tlm_set_error("rank %d failed in %s (code %d); rank %d leaves the exchange with it", r, where, all[r], ctx->rank);
            return TLM_EPEER;
        }
    return TLM_OK;
}

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
    if (!ctx->xchg_scratch) {
        if (int rc = tlm_pool_alloc(&ctx->xchg_scratch, TLM_XCHG_SCRATCH_BYTES, ctx->stream)) return rc;
        TLM_HIP(hipMemsetAsync(ctx->xchg_scratch, 0, TLM_XCHG_SCRATCH_BYTES, ctx->stream));
        TLM_HIP(hipStreamSynchronize(ctx->stream));
    }
    ncclComm_t comm = nullptr;
    {
        const ncclResult_t r = api->CommInitRank(&comm, world, id, rank);
        if (r != ncclSuccess) {                 // (no communicator: tlm_dist_finalize has nothing to do, the scratch goes here)
            tlm_pool_free(ctx->xchg_scratch, true);
            ctx->xchg_scratch = nullptr;
            tlm_set_error("ncclCommInitRank failed: %s", api->GetErrorString ? api->GetErrorString(r) : "?");
            return TLM_EHIP;
        }
    }
    ctx->comm = comm;
    ctx->comm_broken = false;
    ctx->rank = rank;
    ctx->world = world;
    return TLM_OK;
}

extern "C" void tlm_dist_finalize(tlm_ctx *ctx) {
    if (!ctx) return;
    if (!ctx->comm && !ctx->comm_broken) {      // never initialised (or initialisation failed): only the scratch may be left
        if (ctx->xchg_scratch) {
            (void) hipSetDevice(ctx->device);
            tlm_pool_free(ctx->xchg_scratch);
            ctx->xchg_scratch = nullptr;
        }
        return;
    }
    (void) hipSetDevice(ctx->device);
    (void) hipStreamSynchronize(ctx->stream);
    if (ctx->comm) {
        RcclApi *api = rccl_api();
        if (api) (void) api->CommDestroy((ncclComm_t) ctx->comm);
    }
    tlm_pool_free(ctx->xchg_scratch, true);
    ctx->xchg_scratch = nullptr;
    ctx->comm = nullptr;
    ctx->comm_broken = false;
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

// ---- (ABI 5) the SHARDED LOAD: every rank brings its share of the overlap text to its device, the ranks swap the shares.
// Which 1 GiB pieces of a text of `bytes` bytes rank `rank` of `world` uploads: a contiguous run, plain arithmetic.
extern "C" int tlm_dist_text_run(uint64_t bytes, int rank, int world, uint64_t *byte_begin, uint64_t *byte_end) {
    TLM_CHECK(byte_begin && byte_end && world >= 1 && rank >= 0 && rank < world, TLM_EINVAL, "tlm_dist_text_run: bad argument");
    const uint64_t n = (bytes + TLM_TEXT_PIECE - 1) / TLM_TEXT_PIECE;
    *byte_begin = std::min<uint64_t>(bytes, (n * (uint64_t) rank / (uint64_t) world) * TLM_TEXT_PIECE);
    *byte_end = std::min<uint64_t>(bytes, (n * ((uint64_t) rank + 1) / (uint64_t) world) * TLM_TEXT_PIECE);
    return TLM_OK;
}

extern "C" int tlm_dist_text_pieces(const tlm_ctx *ctx, uint64_t bytes, uint64_t *piece_begin, uint64_t *piece_end) {
    TLM_CHECK(ctx && piece_begin && piece_end, TLM_EINVAL, "tlm_dist_text_pieces: null argument");
    const uint64_t n = (bytes + TLM_TEXT_PIECE - 1) / TLM_TEXT_PIECE;
    const uint64_t world = (uint64_t) (ctx->world > 0 ? ctx->world : 1), rank = (uint64_t) ctx->rank;
    *piece_begin = n * rank / world;
    *piece_end = n * (rank + 1) / world;
    return TLM_OK;
}

// COLLECTIVE.  `text` (device, `bytes` bytes, the same size on every rank) holds this rank's pieces; afterwards it holds all of
// them.  Every rank sends its run to every peer and receives every peer's run STRAIGHT INTO ITS PLACE, inside one
// ncclGroupStart/End -- the same full-mesh form as tlm_pathset_allgather: every pair of GPUs on its own xGMI link, no ring, no
// staging, runs of different lengths.  `piece`: TLM_TEXT_PIECE (smaller in the checked build's tests).  `local_rc`: what
// bringing the own pieces came to; swapped first (agree), so that a rank whose upload failed takes everybody out instead of
// leaving them inside the group.
int tlm_text_share(tlm_ctx *ctx, char *text, uint64_t bytes, uint64_t piece, int local_rc, tlm_exchange_info *info) {
    if (info) *info = tlm_exchange_info{};
    if (ctx->world <= 1 && !ctx->comm) return local_rc;
    TLM_CHECK(!ctx->comm_broken, TLM_EPEER, "tlm_text_share: the communicator was aborted after an earlier failure");
    RcclApi *api = rccl_api();
    if (!api) return TLM_EHIP;
    TLM_CHECK(ctx->comm, TLM_EINVAL, "tlm_text_share: the context has no communicator");
    TLM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const auto t0 = std::chrono::steady_clock::now();
    if (int rc = agree(ctx, api, local_rc ? local_rc : (fault_here(ctx, "text") ? TLM_EHIP : TLM_OK), "the sharded load (text upload)")) return rc;
    {   // every rank must be looking at the same file: the sizes are swapped too (a rank with another size would leave its
        // peers with sends and receives that never meet); scratch bytes [0, 8) = mine, [256, 256 + 8 world) = everybody's
        uint64_t mine = bytes, all[TLM_EXCHANGE_MAX_BLOCKS] = {0};
        uint64_t *d = (uint64_t *) ctx->xchg_scratch;
        TLM_CHECK((size_t) ctx->world <= TLM_EXCHANGE_MAX_BLOCKS && 8 * (32 + (size_t) ctx->world) <= TLM_XCHG_SCRATCH_BYTES, TLM_ELIMIT,
                  "tlm_text_share: too many ranks");
        {
            Collective col{ctx, api};
            TLM_HIP(hipMemcpyAsync(d, &mine, 8, hipMemcpyHostToDevice, s));
            TLM_NCCL(api, api->AllGather(d, d + 32, 8, ncclUint8, (ncclComm_t) ctx->comm, s));
            TLM_HIP(hipMemcpyAsync(all, d + 32, 8 * (size_t) ctx->world, hipMemcpyDeviceToHost, s));
            TLM_HIP(hipStreamSynchronize(s));
            col.done();
        }
        for (int r = 0; r < ctx->world; r++)
            TLM_CHECK(all[r] == bytes, TLM_EINVAL, "the sharded load: rank %d sees %llu bytes of text, rank %d sees %llu (not the same file?)",
                      ctx->rank, (unsigned long long) bytes, r, (unsigned long long) all[r]);
    }
    const uint64_t n = (bytes + piece - 1) / piece;
    const uint64_t world = (uint64_t) ctx->world;
    auto run_of = [&](uint64_t r, uint64_t &b0, uint64_t &b1) {        // (tlm_dist_text_pieces, in bytes)
        b0 = std::min<uint64_t>(bytes, (n * r / world) * piece);
        b1 = std::min<uint64_t>(bytes, (n * (r + 1) / world) * piece);
    };
    uint64_t m0, m1;
    run_of((uint64_t) ctx->rank, m0, m1);
    uint64_t sent = 0, received = 0;
    {
        Collective col{ctx, api};
        if (int rc = col.group_start()) return rc;
        for (uint64_t r = 0; r < world; r++) {
            if ((int) r == ctx->rank) continue;
            uint64_t b0, b1;
            run_of(r, b0, b1);
            // piece by piece (runs are whole pieces but for the text's end): no single transfer of more than 2^30 elements,
            // whatever a layer underneath does with its counts; the k-th send to a peer meets its k-th receive from here
            for (uint64_t at = m0; at < m1; at += piece) {
                const uint64_t len = std::min<uint64_t>(piece, m1 - at);
                TLM_NCCL(api, api->Send(text + at, len, ncclUint8, (int) r, (ncclComm_t) ctx->comm, s));
                sent += len;
            }
            for (uint64_t at = b0; at < b1; at += piece) {
                const uint64_t len = std::min<uint64_t>(piece, b1 - at);
                TLM_NCCL(api, api->Recv(text + at, len, ncclUint8, (int) r, (ncclComm_t) ctx->comm, s));
                received += len;
            }
        }
        if (int rc = col.group_end()) return rc;
        TLM_HIP(hipStreamSynchronize(s));
        col.done();
    }
    if (info) {
        info->bytes_sent = sent;
        info->bytes_received = received;
        info->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return TLM_OK;
}

/* The SLICED partition (ABI 4; the default of the hosts since round 4): rank r of `world` walks, of EVERY anchor, the slice
 * [rb * r / world, rb * (r + 1) / world) of its Monte-Carlo trials and the slice [d * r / world, d * (r + 1) / world) of its
 * d deterministic units.  Trials of one anchor are independent walks of the same neighbourhood (per-walk seeds), so every
 * rank gets the same share of every anchor's work whatever the anchors weigh: hostile config, 8 ranks: max / mean 1.007
 * against 1.64 for contiguous bundles (profiles/r04_predicted_scaling_athaliana_hard.json).  units: ascending, or null to
 * ask for the count alone. */
extern "C" int tlm_dist_unit_share(const tlm_graph *g, const tlm_walk_params *p, int rank, int world, int deterministic,
                                   uint64_t *units, uint64_t *n_units) {
    TLM_CHECK(g && n_units && world >= 1 && rank >= 0 && rank < world && (deterministic || p), TLM_EINVAL,
              "tlm_dist_unit_share: bad argument");
    const uint64_t na = g->h_anchors.size(), R = (uint64_t) rank, W = (uint64_t) world;
    uint64_t n = 0;
    if (!deterministic) {
        const uint64_t rb = (uint64_t) std::max(p->rebuild_attempts, 0);
        const uint64_t t0 = rb * R / W, t1 = rb * (R + 1) / W;
        for (uint64_t a = 0; a < na; a++)
            for (uint64_t t = t0; t < t1; t++, n++)
                if (units) units[n] = a * rb + t;
    } else {
        for (uint64_t a = 0; a < na; a++) {
            const uint64_t base = g->h_det_off[a], d = g->h_det_off[a + 1] - base;
            for (uint64_t k = d * R / W; k < d * (R + 1) / W; k++, n++)
                if (units) units[n] = base + k;
        }
    }
    *n_units = n;
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
    TLM_CHECK(!ctx->comm_broken, TLM_EPEER, "tlm_pathset_allgather: the communicator was aborted after an earlier failure");
    RcclApi *api = rccl_api();
    if (!api) return TLM_EHIP;
    TLM_CHECK(ctx->comm, TLM_EINVAL, "tlm_pathset_allgather: tlm_dist_init has not been called");
    ncclComm_t comm = (ncclComm_t) ctx->comm;
    TLM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;

    // ---- 0. local: my slots in 16 bits (anything that fails here is reported to the peers with the sizes)
    uint64_t my_nodes = 0;
    for (int b = 0; b < n_sets; b++) my_nodes += sets[b]->n_nodes_total;
    Tmp s16, wide_flag;
    uint32_t my_wide = 0;
    auto prep = [&]() -> int {
        if (fault_here(ctx, "prep")) return TLM_EHIP;
        TLM_HIP(hipEventRecord(ctx->ev0, s));
        if (int rc = tlm_pool_alloc(&s16.p, (my_nodes ? my_nodes : 1) * 2, s)) return rc;
        if (int rc = tlm_pool_alloc(&wide_flag.p, 4, s)) return rc;
        TLM_HIP(hipMemsetAsync(wide_flag.p, 0, 4, s));
        uint64_t at = 0;
        for (int b = 0; b < n_sets; b++) {
            const uint64_t n = sets[b]->n_nodes_total;
            if (n) hipLaunchKernelGGL(narrow_slots_kernel, dim3(grid_of(n)), dim3(256), 0, s, n, sets[b]->slots,
                                      (uint16_t *) s16.p + at, (uint32_t *) wide_flag.p);
            at += n;
        }
        TLM_HIP(hipMemcpyAsync(&my_wide, wide_flag.p, 4, hipMemcpyDeviceToHost, s));
        TLM_HIP(hipStreamSynchronize(s));
        TLM_HIP(hipGetLastError());
        return TLM_OK;
    };
    const int rc_prep = prep();
    const std::string prep_error = rc_prep ? tlm_last_error() : "";

    // ---- 1. sizes + status of every rank, through the context's scratch: [status, (paths, nodes) x n_sets, wide]
    const size_t per_rank = (size_t) n_sets * 2 + 2;
    std::vector<uint64_t> mine(per_rank), all(per_rank * world);
    mine[0] = (uint64_t) (int64_t) rc_prep;
    for (int b = 0; b < n_sets; b++) {
        mine[1 + 2 * b] = sets[b]->n_paths;
        mine[2 + 2 * b] = sets[b]->n_nodes_total;
    }
    mine[per_rank - 1] = my_wide;
    {
        Collective col{ctx, api};
        uint64_t *d_mine = (uint64_t *) ((char *) ctx->xchg_scratch + 1024), *d_all = (uint64_t *) ((char *) ctx->xchg_scratch + 2048);
        static_assert(2048 + (TLM_EXCHANGE_MAX_BLOCKS / 4) * (8 * 2 + 2) * 8 <= TLM_XCHG_SCRATCH_BYTES, "scratch holds the sizes of 16 ranks");
        TLM_HIP(hipMemcpyAsync(d_mine, mine.data(), per_rank * 8, hipMemcpyHostToDevice, s));
        TLM_NCCL(api, api->AllGather(d_mine, d_all, per_rank, ncclUint64, comm, s));
        TLM_HIP(hipMemcpyAsync(all.data(), d_all, per_rank * 8 * world, hipMemcpyDeviceToHost, s));
        TLM_HIP(hipStreamSynchronize(s));
        col.done();
    }
    if (rc_prep) {
        tlm_set_error("%s", prep_error.c_str());
        return rc_prep;
    }
    bool wide = false;
    std::vector<uint64_t> counts((size_t) world * n_sets * 2), path_base((size_t) world * n_sets), node_base((size_t) world * n_sets);
    for (int r = 0; r < world; r++) {
        TLM_CHECK(all[r * per_rank] == 0, TLM_EPEER, "rank %d failed before the exchange (code %lld); rank %d leaves it too", r,
                  (long long) (int64_t) all[r * per_rank], rank);
        for (int k = 0; k < 2 * n_sets; k++) counts[(size_t) r * n_sets * 2 + k] = all[r * per_rank + 1 + k];
        wide |= all[r * per_rank + per_rank - 1] != 0;
    }
    uint64_t totals[2];
    tlm_exchange_layout(world, n_sets, counts.data(), path_base.data(), node_base.data(), totals);

    // ---- 2. the result (local allocations: agreed on before any block moves), and every block straight into its place
    tlm_pathset *ps = new tlm_pathset();
    ps->ctx = ctx;
    ps->device = ctx->device;
    ps->n_paths = totals[0];
    ps->n_nodes_total = totals[1];
    struct Guard {
        tlm_pathset *p;
        ~Guard() { if (p) tlm_pathset_destroy(p); }
    } guard{ps};
    Tmp r16;            // received 16-bit slots, result layout
    auto alloc_result = [&]() -> int {
        if (fault_here(ctx, "alloc")) return TLM_EHIP;
        auto palloc = [&](void **p, size_t bytes) { return tlm_pool_alloc(p, bytes ? bytes : 8, s); };
        if (int rc = palloc((void **) &ps->unit, totals[0] * 8)) return rc;
        if (int rc = palloc((void **) &ps->length, totals[0] * 8)) return rc;
        if (int rc = palloc((void **) &ps->off, (totals[0] + 1) * 8)) return rc;
        if (int rc = palloc((void **) &ps->nodes, totals[1] * 4)) return rc;
        if (int rc = palloc((void **) &ps->slots, totals[1] * 4)) return rc;
        if (!wide)
            if (int rc = tlm_pool_alloc(&r16.p, (totals[1] ? totals[1] : 1) * 2, s)) return rc;
        return TLM_OK;
    };
    if (int rc = agree(ctx, api, alloc_result(), "tlm_pathset_allgather (result buffers)")) return rc;
    const size_t slot_bytes = wide ? 4 : 2;
    uint64_t sent = 0, received = 0;
    Collective col{ctx, api};
    if (int rc = col.group_start()) return rc;
    uint64_t my_node_at = 0;
    bool first_send = true;
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
                if (first_send && fault_here(ctx, "group")) return TLM_EHIP;
                first_send = false;
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
    if (int rc = col.group_end()) return rc;
    col.done();

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
    }
    // blocks were placed builder by builder in rank order, which IS paths_ order when every rank walked a contiguous range
    // of units; the blocks of a sliced partition interleave: ascending (builder | unit) is restored here
    guard.p = nullptr;
    tlm_pathset *sorted = nullptr;
    if (int rc = tlm_pathset_sort_by_unit(ctx, ps, &sorted)) {
        tlm_pathset_destroy(ps);
        return rc;
    }
    if (info) info->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_wall).count();
    *out = sorted;
    return TLM_OK;
}

// ================================================================================================================
// The LAZY form of the exchange.  The consumers of the gathered Path set (reference Main.cpp:247-352: filterUnique,
// getPathsBetweenAnchors, PathGroup / PathWindow grouping, the consensus tournament) read of EVERY path its length, its
// end nodes and whether it equals another path; node lists they need for the ~200 consensus paths the scaffold is made
// of.  So what travels for every path is a 45-byte summary (unit, length, front, back, node count, 64-bit hash of the node
// sequence, "equals its predecessor" flag -- exact, computed where the paths are), one ncclAllGather of a few MB; node
// lists stay on the rank that built them and are broadcast by their owner when somebody asks (tlm_gathered_fetch, a
// collective every rank joins with the same request).  A. thaliana config at N = 8: 2.1 MB per rank instead of 0.98 GB to
// each of 7 peers.
struct tlm_gathered {
    tlm_ctx *ctx = nullptr;
    int world = 1, rank = 0, n_sets = 0;
    tlm_pathset *local = nullptr;           // this rank's sets, concatenated (builder tags in `unit`)
    std::vector<uint64_t> counts;           // [world][n_sets][2]
    // every path of the job, in paths_ order (ascending builder tag | unit)
    std::vector<uint64_t> unit, hash, local_index;      // local_index: position inside the owner's concatenated set
    std::vector<int64_t> length;
    std::vector<uint32_t> front, back, n_nodes, owner;
    std::vector<uint8_t> dup, block_first;
};

namespace {

struct SummaryRow {         // one path on the wire (SoA on the device, this is only the byte budget): 45 bytes
    static constexpr size_t BYTES = 8 + 8 + 4 + 4 + 4 + 8 + 1;
};

}   // namespace

extern "C" void tlm_gathered_destroy(tlm_gathered *g) {
    if (!g) return;
    tlm_pathset_destroy(g->local);
    delete g;
}

extern "C" int tlm_pathset_gather_summaries(tlm_ctx *ctx, int n_sets, tlm_pathset *const *sets, tlm_gathered **out,
                                            tlm_exchange_info *info) {
    TLM_CHECK(ctx && sets && out && n_sets >= 1 && n_sets <= 8, TLM_EINVAL, "tlm_pathset_gather_summaries: bad argument");
    const int world = ctx->world, rank = ctx->rank;
    TLM_CHECK(world * n_sets <= TLM_EXCHANGE_MAX_BLOCKS, TLM_ELIMIT, "tlm_pathset_gather_summaries: %d blocks", world * n_sets);
    TLM_CHECK(!ctx->comm_broken, TLM_EPEER, "tlm_pathset_gather_summaries: the communicator was aborted after an earlier failure");
    const bool single = world == 1 && !(ctx->comm && getenv("TLM_EXCHANGE_SELFTEST"));
    RcclApi *api = single ? nullptr : rccl_api();
    if (!single && !api) return TLM_EHIP;
    TLM_CHECK(single || ctx->comm, TLM_EINVAL, "tlm_pathset_gather_summaries: tlm_dist_init has not been called");
    if (info) std::memset(info, 0, sizeof(*info));
    const auto t_wall = std::chrono::steady_clock::now();
    TLM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    tlm_gathered *g = new tlm_gathered();
    struct Guard {
        tlm_gathered *p;
        ~Guard() { tlm_gathered_destroy(p); }
    } guard{g};
    g->ctx = ctx;
    g->world = world;
    g->rank = rank;
    g->n_sets = n_sets;
    // local: this rank's sets as one (what fails here is reported to the peers with the sizes)
    auto prep = [&]() -> int {
        if (fault_here(ctx, "prep")) return TLM_EHIP;
        TLM_HIP(hipEventRecord(ctx->ev0, s));
        return tlm_pathset_concat(ctx, n_sets, sets, &g->local);
    };
    const int rc_prep = prep();
    const std::string prep_error = rc_prep ? tlm_last_error() : "";
    if (single && rc_prep) return rc_prep;
    const uint64_t mine = rc_prep ? 0 : g->local->n_paths;

    // ---- sizes: [status, (paths, nodes) of every block]
    const size_t per_rank = (size_t) n_sets * 2, wire = per_rank + 1;
    std::vector<uint64_t> my_counts(wire), all_counts(wire * world);
    my_counts[0] = (uint64_t) (int64_t) rc_prep;
    g->counts.assign(per_rank * world, 0);
    for (int b = 0; b < n_sets; b++) {
        my_counts[1 + 2 * b] = sets[b]->n_paths;
        my_counts[2 + 2 * b] = sets[b]->n_nodes_total;
    }
    if (single) {
        for (size_t k = 0; k < per_rank; k++) g->counts[k] = my_counts[1 + k];
    } else {
        Collective col{ctx, api};
        uint64_t *d_mine = (uint64_t *) ((char *) ctx->xchg_scratch + 1024), *d_all = (uint64_t *) ((char *) ctx->xchg_scratch + 2048);
        TLM_HIP(hipMemcpyAsync(d_mine, my_counts.data(), wire * 8, hipMemcpyHostToDevice, s));
        TLM_NCCL(api, api->AllGather(d_mine, d_all, wire, ncclUint64, (ncclComm_t) ctx->comm, s));
        TLM_HIP(hipMemcpyAsync(all_counts.data(), d_all, wire * 8 * world, hipMemcpyDeviceToHost, s));
        TLM_HIP(hipStreamSynchronize(s));
        col.done();
        if (rc_prep) {
            tlm_set_error("%s", prep_error.c_str());
            return rc_prep;
        }
        for (int r = 0; r < world; r++) {
            TLM_CHECK(all_counts[r * wire] == 0, TLM_EPEER, "rank %d failed before the exchange (code %lld); rank %d leaves it too", r,
                      (long long) (int64_t) all_counts[r * wire], rank);
            for (size_t k = 0; k < per_rank; k++) g->counts[(size_t) r * per_rank + k] = all_counts[r * wire + 1 + k];
        }
    }
    uint64_t max_paths = 0;
    std::vector<uint64_t> rank_paths(world, 0);
    for (int r = 0; r < world; r++) {
        for (int b = 0; b < n_sets; b++) rank_paths[r] += g->counts[(size_t) r * per_rank + 2 * b];
        max_paths = std::max(max_paths, rank_paths[r]);
    }

    // ---- summaries: this rank's columns, padded to the longest rank, one all-gather
    // slab of a rank: [unit u64 x P][length i64 x P][hash u64 x P][front u32 x P][back u32 x P][n_nodes u32 x P][dup u8 x P]
    const uint64_t P = std::max<uint64_t>(max_paths, 1);
    const size_t slab = (size_t) P * SummaryRow::BYTES;
    Tmp d_slab, d_allslabs;
    char *h_all = nullptr;
    auto fill = [&]() -> int {
        if (fault_here(ctx, "alloc")) return TLM_EHIP;
        // the slabs of all ranks land in pinned host memory kept by the context (a pageable destination is staged by the
        // runtime at a fraction of the link's rate: 16.7 MB of summaries per step for the A. thaliana-shaped config)
        if (ctx->xchg_pinned_bytes < slab * world) {
            if (ctx->xchg_pinned) (void) hipHostFree(ctx->xchg_pinned);
            ctx->xchg_pinned = nullptr;
            ctx->xchg_pinned_bytes = 0;
            const size_t want = slab * world + (slab * world) / 4;
            TLM_HIP(hipHostMalloc(&ctx->xchg_pinned, want, hipHostMallocDefault));
            ctx->xchg_pinned_bytes = want;
        }
        h_all = (char *) ctx->xchg_pinned;
        TLM_CHECK(rank_paths[rank] == mine, TLM_EINTERNAL, "tlm_pathset_gather_summaries: local count mismatch");
        if (int rc = tlm_pool_alloc(&d_slab.p, slab, s)) return rc;
        if (!single)
            if (int rc = tlm_pool_alloc(&d_allslabs.p, slab * world, s)) return rc;
        char *base = (char *) d_slab.p;
        uint64_t *d_unit = (uint64_t *) base;
        int64_t *d_len = (int64_t *) (base + 8 * P);
        uint64_t *d_hash = (uint64_t *) (base + 16 * P);
        uint32_t *d_front = (uint32_t *) (base + 24 * P), *d_back = (uint32_t *) (base + 28 * P), *d_cnt = (uint32_t *) (base + 32 * P);
        uint8_t *d_dup = (uint8_t *) (base + 36 * P);
        TLM_HIP(hipMemsetAsync(base, 0, slab, s));
        if (mine) {
            TLM_HIP(hipMemcpyAsync(d_unit, g->local->unit, mine * 8, hipMemcpyDeviceToDevice, s));
            TLM_HIP(hipMemcpyAsync(d_len, g->local->length, mine * 8, hipMemcpyDeviceToDevice, s));
            if (int rc = tlm_pathset_summary_dev(g->local, s, d_front, d_back, d_cnt, d_hash, d_dup)) return rc;
        }
        return TLM_OK;
    };
    if (single) {
        if (int rc = fill()) return rc;
        TLM_HIP(hipMemcpyAsync(h_all, d_slab.p, slab, hipMemcpyDeviceToHost, s));
        TLM_HIP(hipEventRecord(ctx->ev1, s));
        TLM_HIP(hipStreamSynchronize(s));
    } else {
        if (int rc = agree(ctx, api, fill(), "tlm_pathset_gather_summaries (summary buffers)")) return rc;
        Collective col{ctx, api};
        TLM_NCCL(api, api->AllGather(d_slab.p, d_allslabs.p, slab, ncclUint8, (ncclComm_t) ctx->comm, s));
        TLM_HIP(hipMemcpyAsync(h_all, d_allslabs.p, slab * world, hipMemcpyDeviceToHost, s));
        TLM_HIP(hipEventRecord(ctx->ev1, s));
        TLM_HIP(hipStreamSynchronize(s));
        col.done();
    }

    // ---- paths_ order on the host: ascending (builder tag | unit).  Every rank's slab is ascending (tlm_pathset_concat:
    // builder by builder, units ascending inside), so this is a merge of `world` sorted runs; with contiguous bundles it
    // degenerates to "builder by builder, rank order inside", with a sliced partition the ranks alternate anchor by anchor
    uint64_t total = 0;
    for (int r = 0; r < world; r++) total += rank_paths[r];
    g->unit.resize(total); g->hash.resize(total); g->local_index.resize(total); g->length.resize(total);
    g->front.resize(total); g->back.resize(total); g->n_nodes.resize(total); g->owner.resize(total);
    g->dup.resize(total); g->block_first.assign(total, 0);
    {
        std::vector<uint64_t> head(world, 0);
        auto unit_of = [&](int r, uint64_t k) {
            uint64_t u;
            std::memcpy(&u, h_all + slab * r + 8 * k, 8);
            return u;
        };
        int prev_r = -1;
        uint64_t prev_k = 0, prev_unit = 0;
        for (uint64_t at = 0; at < total; at++) {
            int r = -1;
            uint64_t best = 0;
            for (int q = 0; q < world; q++) {
                if (head[q] >= rank_paths[q]) continue;
                const uint64_t u = unit_of(q, head[q]);
                if (r < 0 || u < best) {        // (equal units on two ranks cannot happen: a unit has one owner; lowest rank first)
                    r = q;
                    best = u;
                }
            }
            const uint64_t k = head[r]++;
            const char *rb = h_all + slab * r;
            g->unit[at] = best;
            std::memcpy(&g->length[at], rb + 8 * P + 8 * k, 8);
            std::memcpy(&g->hash[at], rb + 16 * P + 8 * k, 8);
            std::memcpy(&g->front[at], rb + 24 * P + 4 * k, 4);
            std::memcpy(&g->back[at], rb + 28 * P + 4 * k, 4);
            std::memcpy(&g->n_nodes[at], rb + 32 * P + 4 * k, 4);
            g->dup[at] = (uint8_t) rb[36 * P + k];
            g->owner[at] = (uint32_t) r;
            g->local_index[at] = k;
            // the owner compared path k with path k - 1 of ITS set: that flag stands iff k - 1 is also the predecessor in
            // paths_ order; otherwise the predecessor lives elsewhere and the caller settles it (tlm_gathered_fetch) when the
            // two summaries coincide
            // (a builder's first path is marked too, as before: Path sets of different builder calls meet there)
            if (!(r == prev_r && k == prev_k + 1) || (best >> 56) != (prev_unit >> 56)) {
                g->block_first[at] = 1;
                g->dup[at] = 0;
            }
            prev_r = r;
            prev_k = k;
            prev_unit = best;
        }
    }
    if (info) {
        float ms = 0;
        (void) hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
        info->device_ms = ms;
        info->bytes_sent = single ? 0 : (uint64_t) (slab + per_rank * 8) * (uint64_t) (world - 1);
        info->bytes_received = info->bytes_sent;
        info->slot_bits = 0;
        info->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_wall).count();
    }
    guard.p = nullptr;
    *out = g;
    return TLM_OK;
}

extern "C" int tlm_gathered_sizes(const tlm_gathered *g, uint64_t *n_paths, uint64_t *n_local_paths) {
    TLM_CHECK(g, TLM_EINVAL, "tlm_gathered_sizes: null argument");
    if (n_paths) *n_paths = g->unit.size();
    if (n_local_paths) *n_local_paths = g->local->n_paths;
    return TLM_OK;
}

extern "C" int tlm_gathered_export(const tlm_gathered *g, uint64_t *unit, int64_t *length, uint32_t *front, uint32_t *back,
                                   uint32_t *n_nodes, uint64_t *hash, uint32_t *owner, uint8_t *dup_of_prev,
                                   uint8_t *block_first) {
    TLM_CHECK(g, TLM_EINVAL, "tlm_gathered_export: null argument");
    const size_t n = g->unit.size();
    auto put = [n](void *dst, const void *src, size_t elem) { if (dst && n) std::memcpy(dst, src, n * elem); };
    put(unit, g->unit.data(), 8); put(length, g->length.data(), 8); put(front, g->front.data(), 4); put(back, g->back.data(), 4);
    put(n_nodes, g->n_nodes.data(), 4); put(hash, g->hash.data(), 8); put(owner, g->owner.data(), 4);
    put(dup_of_prev, g->dup.data(), 1); put(block_first, g->block_first.data(), 1);
    return TLM_OK;
}

// COLLECTIVE: every rank calls it with the same `index` list (positions in paths_ order).  For every owner that appears
// in the list: the owner gathers the requested node / slot lists out of its set into a staging buffer and broadcasts it
// (sizes are known to everybody from the summaries); everybody ends up with all requested lists, in request order.
extern "C" int tlm_gathered_fetch(tlm_ctx *ctx, const tlm_gathered *g, uint64_t n, const uint64_t *index, uint32_t *nodes_host,
                                  uint32_t *slots_host) {
    TLM_CHECK(ctx && g && g->ctx == ctx && (n == 0 || index), TLM_EINVAL, "tlm_gathered_fetch: bad argument");
    if (!n) return TLM_OK;
    const uint64_t total_paths = g->unit.size();
    for (uint64_t k = 0; k < n; k++)
        TLM_CHECK(index[k] < total_paths, TLM_EINVAL, "tlm_gathered_fetch: path %llu of %llu", (unsigned long long) index[k],
                  (unsigned long long) total_paths);
    const int world = g->world, rank = g->rank;
    TLM_CHECK(!ctx->comm_broken, TLM_EPEER, "tlm_gathered_fetch: the communicator was aborted after an earlier failure");
    const bool single = world == 1 && !(ctx->comm && getenv("TLM_EXCHANGE_SELFTEST"));
    RcclApi *api = single ? nullptr : rccl_api();
    if (!single && !api) return TLM_EHIP;
    TLM_CHECK(single || ctx->comm, TLM_EINVAL, "tlm_gathered_fetch: the context has no communicator");
    TLM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    // staging layout: owner by owner, inside an owner in request order
    std::vector<uint64_t> dst(n + 1, 0);            // where request k lies in the staging buffer
    std::vector<uint64_t> owner_begin(world + 1, 0);
    std::vector<std::vector<uint64_t>> req_of(world);
    for (uint64_t k = 0; k < n; k++) req_of[g->owner[index[k]]].push_back(k);
    uint64_t at = 0;
    for (int o = 0; o < world; o++) {
        owner_begin[o] = at;
        for (uint64_t k : req_of[o]) {
            dst[k] = at;
            at += g->n_nodes[index[k]];
        }
    }
    owner_begin[world] = at;
    const uint64_t total = at;
    Tmp st_nodes, st_slots, d_idx, d_dst;
    // local: my share of the request gathered into the staging buffer (its outcome is swapped before anybody broadcasts)
    auto stage = [&]() -> int {
        if (fault_here(ctx, "fetch")) return TLM_EHIP;
        if (int rc = tlm_pool_alloc(&st_nodes.p, (total ? total : 1) * 4, s)) return rc;
        if (int rc = tlm_pool_alloc(&st_slots.p, (total ? total : 1) * 4, s)) return rc;
        const std::vector<uint64_t> &mine = req_of[rank];
        if (mine.empty()) return TLM_OK;
        std::vector<uint64_t> li(mine.size()), ld(mine.size() + 1);
        for (size_t j = 0; j < mine.size(); j++) {
            li[j] = g->local_index[index[mine[j]]];
            ld[j] = dst[mine[j]];
        }
        ld[mine.size()] = owner_begin[rank + 1];
        if (int rc = tlm_pool_alloc(&d_idx.p, li.size() * 8, s)) return rc;
        if (int rc = tlm_pool_alloc(&d_dst.p, ld.size() * 8, s)) return rc;
        TLM_HIP(hipMemcpyAsync(d_idx.p, li.data(), li.size() * 8, hipMemcpyHostToDevice, s));
        TLM_HIP(hipMemcpyAsync(d_dst.p, ld.data(), ld.size() * 8, hipMemcpyHostToDevice, s));
        if (int rc = tlm_pathset_gather_dev(g->local, s, li.size(), (const uint64_t *) d_idx.p, (const uint64_t *) d_dst.p,
                                            (uint32_t *) st_nodes.p, (uint32_t *) st_slots.p))
            return rc;
        TLM_HIP(hipStreamSynchronize(s));       // li / ld go out of scope
        return TLM_OK;
    };
    if (single) {
        if (int rc = stage()) return rc;
    } else {
        if (int rc = agree(ctx, api, stage(), "tlm_gathered_fetch (staging)")) return rc;
        Collective col{ctx, api};
        if (int rc = col.group_start()) return rc;
        for (int o = 0; o < world; o++) {
            const uint64_t len = owner_begin[o + 1] - owner_begin[o];
            if (!len) continue;
            uint32_t *pn = (uint32_t *) st_nodes.p + owner_begin[o], *psl = (uint32_t *) st_slots.p + owner_begin[o];
            TLM_NCCL(api, api->Broadcast(pn, pn, len * 4, ncclUint8, o, (ncclComm_t) ctx->comm, s));
            TLM_NCCL(api, api->Broadcast(psl, psl, len * 4, ncclUint8, o, (ncclComm_t) ctx->comm, s));
        }
        if (int rc = col.group_end()) return rc;
        col.done();
    }
    std::vector<uint32_t> hn(total), hs(total);
    if (total) {
        TLM_HIP(hipMemcpyAsync(hn.data(), st_nodes.p, total * 4, hipMemcpyDeviceToHost, s));
        TLM_HIP(hipMemcpyAsync(hs.data(), st_slots.p, total * 4, hipMemcpyDeviceToHost, s));
    }
    TLM_HIP(hipStreamSynchronize(s));
    TLM_HIP(hipGetLastError());
    uint64_t out_at = 0;
    for (uint64_t k = 0; k < n; k++) {
        const uint64_t len = g->n_nodes[index[k]];
        if (nodes_host && len) std::memcpy(nodes_host + out_at, hn.data() + dst[k], len * 4);
        if (slots_host && len) std::memcpy(slots_host + out_at, hs.data() + dst[k], len * 4);
        out_at += len;
    }
    return TLM_OK;
}

// telomeri_amd/csrc/csr_build.hip -- OverlapGraph::load's graph construction on the device.
//
// Replaces the body of OverlapGraph::buildFrom / nodeIndex (reference src/OverlapGraph.cpp:204-258)
// for all files of one graph:
//   1. filter_score per file (filter_score.hip)                          -> keep, SI/OS/ES per record
//   2. exclusive scan of keep                                            -> rank k of each kept record
//   3. first-seen node numbering among kept records, query before target (:208-218): atomicMin of
//      the position 2k+role per name id, then a radix sort of the (<= n_names) first positions
//   4. two adjacency entries per kept record (:241-248): e=2k owned by the target (unswapped),
//      e=2k+1 owned by the query (q/t index, start, end swapped; scores, strand as they are);
//      a STABLE radix sort of (owner, e) gives CSR order = per-node record order
//   5. gather of the record columns into the CSR arrays (hot: {nbr|anchor, ES}, {nbr|anchor, OS},
//      q_start, t_start; cold: q_end, t_end, SI, strand)
//   6. per-node order-safety flags for the roulette sums, anchor list, DET unit table
// The sort and the scans are rocPRIM device primitives (native ROCm, no CUDA layer); everything
// touching the reference's semantics is a kernel in this file.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "tlm_common.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace {

// Every buffer of a build -- temporaries and the graph's own arrays -- comes from the stream-ordered pool
// (tlm_pool_alloc): a second build reuses the first one's memory instead of paying for fresh GB-sized mappings.
static thread_local hipStream_t t_build_stream = nullptr;     // the stream of the build running on this thread

struct DevBuf {   // RAII for temporaries
    void *p = nullptr;
    ~DevBuf() { if (p) tlm_pool_free(p); }
    template <class T> T *as() { return (T *) p; }
};

int dev_alloc(DevBuf &b, size_t bytes) {
    return tlm_pool_alloc(&b.p, bytes ? bytes : 16, t_build_stream);
}

template <class T> int dev_alloc_keep(T **out, size_t n) {
    return tlm_pool_alloc((void **) out, (n ? n : 1) * sizeof(T), t_build_stream);
}

// Phase marks of a build: HIP events on the build stream; interval k (between marks k and k+1) is booked on a phase of
// tlm_graph_info.phase_ms once the build has ended.  Host waits for the device are counted (host_syncs).
enum BuildPhase { PH_FILTER = 0, PH_RANK, PH_FIRST_SEEN, PH_NODES, PH_OWNER_SORT, PH_GATHER, PH_FLAGS, PH_BFS, PH_ORDER_SORT,
                  PH_LAYOUT, PH_DET_TABLE, PH_OTHER };
static_assert(PH_OTHER + 1 == TLM_BUILD_PHASES, "phase table");

struct Phases {
    tlm_ctx *ctx;
    hipStream_t s;
    int n = 0;
    int phase_of[64];
    uint32_t syncs = 0;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double malloc_ms0 = tlm_pool_thread_malloc_ms();
    // the HOST's side of the same timeline: when each mark was made, and how long the host waited in sync() since the
    // previous mark -- a phase whose host time (waits excluded) is long was held up on the host, whatever its kernels took
    std::chrono::steady_clock::time_point t_host[64];
    double wait_ms[64];
    double wait_since_mark = 0, wait_total = 0;
    Phases(tlm_ctx *c, hipStream_t st) : ctx(c), s(st) { mark(PH_OTHER); }
    void mark(int phase) {          // `phase` ENDS here (the first mark only opens the timeline)
        if (n >= 64) return;
        t_host[n] = std::chrono::steady_clock::now();
        wait_ms[n] = wait_since_mark;
        wait_since_mark = 0;
        while ((int) ctx->phase_ev.size() <= n) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return;
            ctx->phase_ev.push_back(e);
        }
        if (hipEventRecord(ctx->phase_ev[n], s) != hipSuccess) return;
        phase_of[n++] = phase;
    }
    int sync() {
        syncs++;
        const auto w0 = std::chrono::steady_clock::now();
        TLM_HIP(hipStreamSynchronize(s));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
        wait_since_mark += ms;
        wait_total += ms;
        return TLM_OK;
    }
    void collect(tlm_graph_info &info) {     // after the final sync
        for (int k = 1; k < n; k++) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, ctx->phase_ev[k - 1], ctx->phase_ev[k]) == hipSuccess) info.phase_ms[phase_of[k]] += ms;
            info.phase_host_ms[phase_of[k]] += std::chrono::duration<double, std::milli>(t_host[k] - t_host[k - 1]).count() - wait_ms[k];
        }
        info.sync_wait_ms += wait_total;
        info.host_syncs += syncs;
        info.build_wall_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        info.build_alloc_ms += tlm_pool_thread_malloc_ms() - malloc_ms0;
    }
};

struct FileCols {
    const int32_t *q_id, *t_id, *q_len, *q_start, *q_end, *t_len, *t_start, *t_end;
    const uint8_t *strand;
    uint64_t base;      // global record index of the file's record 0
    uint64_t n;
    int anchors;
};
constexpr int MAX_FILES = 4;
struct FileTable {
    FileCols f[MAX_FILES];
    int n_files;
};

__device__ inline int file_of(const FileTable &ft, uint64_t g) {
    int f = 0;
#pragma unroll
    for (int i = 1; i < MAX_FILES; i++)
        if (i < ft.n_files && g >= ft.f[i].base) f = i;
    return f;
}

struct KeepToU32 {
    __host__ __device__ uint32_t operator()(uint8_t k) const { return k; }
};

// step 3a: first position (2k + role) at which a kept record mentions each name; also kept_rec[k] = g
__global__ __launch_bounds__(256) void first_seen_kernel(FileTable ft, int fi, const uint8_t *__restrict__ keep,
                                                         const uint32_t *__restrict__ krank,
                                                         uint32_t *__restrict__ first_pos,
                                                         uint32_t *__restrict__ kept_rec) {
    const FileCols &f = ft.f[fi];
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < f.n; i += stride) {
        const uint64_t g = f.base + i;
        if (!keep[g]) continue;
        const uint32_t k = krank[g];
        kept_rec[k] = (uint32_t) g;
        // (a plain read first: after the first few records of a name nearly every atomic would lose anyway, and a
        // stale value only means one atomic too many)
        const uint32_t q = (uint32_t) f.q_id[i], t = (uint32_t) f.t_id[i];
        if (2u * k < __builtin_nontemporal_load(&first_pos[q])) atomicMin(&first_pos[q], 2u * k);
        if (2u * k + 1u < __builtin_nontemporal_load(&first_pos[t])) atomicMin(&first_pos[t], 2u * k + 1u);
    }
}

__global__ void iota_kernel(uint32_t *p, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

// step 3b: names sorted by first position -> node index; node attributes come from the creating
// record (OverlapGraph.cpp:211,217): length = that sequence's length, anchor = (contig position
// of the record == role of the name in it), ContigPosition per :130-138.
__global__ __launch_bounds__(256) void make_nodes_kernel(FileTable ft, uint32_t n_nodes,
                                                         const uint32_t *__restrict__ sorted_pos,
                                                         const uint32_t *__restrict__ sorted_name,
                                                         const uint32_t *__restrict__ kept_rec,
                                                         const uint8_t *__restrict__ name_is_ctg,
                                                         uint32_t *__restrict__ node_of_name,
                                                         uint32_t *__restrict__ node_len,
                                                         uint8_t *__restrict__ node_anchor,
                                                         int32_t *__restrict__ node_name_id) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const uint32_t pos = sorted_pos[i], name = sorted_name[i];
    const uint32_t role = pos & 1u;              // 0 = query, 1 = target
    const uint64_t g = kept_rec[pos >> 1];
    const int fi = file_of(ft, g);
    const FileCols &f = ft.f[fi];
    const uint64_t r = g - f.base;
    bool anchor = false;
    if (f.anchors) {
        const bool pos_query = name_is_ctg[f.q_id[r]] != 0;   // ContigPosition::QUERY else TARGET
        anchor = pos_query ? (role == 0) : (role == 1);
    }
    node_of_name[name] = i;
    node_len[i] = (uint32_t) (role == 0 ? f.q_len[r] : f.t_len[r]);
    node_anchor[i] = anchor ? 1 : 0;
    node_name_id[i] = (int32_t) name;
}

// step 4: (owner, e) pairs
__global__ __launch_bounds__(256) void make_entries_kernel(FileTable ft, uint64_t n_kept,
                                                           const uint32_t *__restrict__ kept_rec,
                                                           const uint32_t *__restrict__ node_of_name,
                                                           uint32_t *__restrict__ owner, uint32_t *__restrict__ eidx) {
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; k < n_kept; k += stride) {
        const uint64_t g = kept_rec[k];
        const FileCols &f = ft.f[file_of(ft, g)];
        const uint64_t r = g - f.base;
        const uint32_t qn = node_of_name[f.q_id[r]], tn = node_of_name[f.t_id[r]];
        owner[2 * k] = tn;        // nodes_[tn_index].edges.push_back(e)          :243
        owner[2 * k + 1] = qn;    // swap; nodes_[qn_index].edges.push_back(e)    :245-248
        eidx[2 * k] = (uint32_t) (2 * k);
        eidx[2 * k + 1] = (uint32_t) (2 * k + 1);
    }
}

__global__ __launch_bounds__(256) void row_ptr_kernel(uint64_t n_adj, uint32_t n_nodes,
                                                      const uint32_t *__restrict__ sorted_owner,
                                                      uint32_t *__restrict__ row_ptr) {
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    for (uint64_t d = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; d < n_adj; d += stride) {
        // every node owns at least one entry (nodes exist only through kept records)
        if (d == 0 || sorted_owner[d] != sorted_owner[d - 1]) row_ptr[sorted_owner[d]] = (uint32_t) d;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) row_ptr[n_nodes] = (uint32_t) n_adj;
}

// step 5a: one 48-byte record per KEPT overlap, in kept order (kept_rec is increasing, so the column reads are almost
// coalesced): everything both adjacency entries of the overlap need.  The gather below then touches one such record
// per entry instead of a dozen scattered 4-byte words.
struct KeptFields {
    uint4 a;    // qn, tn, q_start, q_end          (node indices of query and target)
    uint4 b;    // t_start, t_end, ES bits, OS bits
    uint4 c;    // SI bits, strand | anchor(qn) << 1 | anchor(tn) << 2 | proper overlap << 3 (see proper_sides), -, -
};

// Is an overlap a PROPER one -- a dovetail or a containment: on each side at least one of the two sequences ends (within
// 1000 bp, and the overhangs are at most a fifth of the aligned span; the classification assemblers use) -- or an INTERNAL
// match, where both sequences run on past the overlap on the same side (two reads sharing a repeat copy; a chimeric read
// against the reads of one of its loci)?  Used by the walk-order numbering only (graph_finish).
__host__ __device__ inline bool proper_sides(int64_t q5, int64_t q3, int64_t t5, int64_t t3, int64_t span) {
    const int64_t e5 = q5 < t5 ? q5 : t5, e3 = q3 < t3 ? q3 : t3;      // what BOTH sequences have beyond the overlap
    return e5 <= 1000 && e3 <= 1000 && 5 * span >= 4 * (span + e5 + e3);
}
__host__ __device__ inline bool proper_overlap(int64_t qs, int64_t qe, int64_t ql, int64_t ts, int64_t te, int64_t tl, bool minus) {
    const int64_t span = (qe - qs) < (te - ts) ? (qe - qs) : (te - ts);
    if (proper_sides(qs, ql - qe, ts, tl - te, span)) return true;
    // a '-' record names the query's coordinates on its own strand: its left end meets the target's right side
    return minus && proper_sides(ql - qe, qs, ts, tl - te, span);
}

__global__ __launch_bounds__(256) void kept_fields_kernel(FileTable ft, uint64_t n_kept,
                                                          const uint32_t *__restrict__ kept_rec,
                                                          const uint32_t *__restrict__ node_of_name,
                                                          const uint8_t *__restrict__ node_anchor,
                                                          const float *__restrict__ r_si, const float *__restrict__ r_os,
                                                          const float *__restrict__ r_es, KeptFields *__restrict__ out) {
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; k < n_kept; k += stride) {
        const uint64_t g = kept_rec[k];
        const FileCols &f = ft.f[file_of(ft, g)];
        const uint64_t r = g - f.base;
        const uint32_t qn = node_of_name[f.q_id[r]], tn = node_of_name[f.t_id[r]];
        KeptFields o;
        o.a = make_uint4(qn, tn, (uint32_t) f.q_start[r], (uint32_t) f.q_end[r]);
        o.b = make_uint4((uint32_t) f.t_start[r], (uint32_t) f.t_end[r], __float_as_uint(r_es[g]), __float_as_uint(r_os[g]));
        const uint32_t st = f.strand ? (f.strand[r] != 0 ? 1u : 0u) : 0u;
        const uint32_t proper = proper_overlap(f.q_start[r], f.q_end[r], f.q_len[r], f.t_start[r], f.t_end[r], f.t_len[r], st != 0) ? 8u : 0u;
        o.c = make_uint4(__float_as_uint(r_si[g]), st | ((uint32_t) node_anchor[qn] << 1) | ((uint32_t) node_anchor[tn] << 2) | proper, 0u, 0u);
        out[k] = o;
    }
}

// step 5b
__global__ __launch_bounds__(256) void gather_kernel(uint64_t n_adj, const uint32_t *__restrict__ sorted_e,
                                                     const KeptFields *__restrict__ kf, uint2 *__restrict__ ent_es,
                                                     uint2 *__restrict__ ent_os, uint32_t *__restrict__ q_start,
                                                     uint32_t *__restrict__ q_end, uint32_t *__restrict__ t_start,
                                                     uint32_t *__restrict__ t_end, float *__restrict__ si,
                                                     uint8_t *__restrict__ strand, uint8_t *__restrict__ proper,
                                                     uint32_t *__restrict__ n_internal) {
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    uint32_t internal = 0;
    for (uint64_t d = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; d < n_adj; d += stride) {
        const uint32_t e = sorted_e[d];
        const bool swapped = (e & 1u) != 0;
        const KeptFields f = kf[e >> 1];
        const uint32_t qn = f.a.x, tn = f.a.y, qs = f.a.z, qe = f.a.w, ts = f.b.x, te = f.b.y;
        const uint32_t nbr = swapped ? tn : qn;       // Edge::q_index of the entry
        const uint32_t anchor = swapped ? (f.c.y >> 2) & 1u : (f.c.y >> 1) & 1u;
        const uint32_t word = nbr | (anchor << 31);
        ent_es[d] = make_uint2(word, f.b.z);
        ent_os[d] = make_uint2(word, f.b.w);
        q_start[d] = swapped ? ts : qs;
        q_end[d] = swapped ? te : qe;
        t_start[d] = swapped ? qs : ts;
        t_end[d] = swapped ? qe : te;
        si[d] = __uint_as_float(f.c.x);
        strand[d] = (uint8_t) (f.c.y & 1u);
        proper[d] = (uint8_t) ((f.c.y >> 3) & 1u);      // (both entries of an overlap: the classification is symmetric)
        internal += ((f.c.y >> 3) & 1u) ^ 1u;
    }
    // how many entries are internal matches (saturating): a graph without any is numbered along all its entries, as before
    for (int d = 32; d > 0; d >>= 1) internal += __shfl_xor(internal, d);
    if ((threadIdx.x & 63) == 0 && internal) atomicOr(n_internal, 1u);
}

// step 6a: is every roulette sum over a subset of this node's list independent of the order of
// the additions, and non-decreasing along the list?  All m are >= 0 and multiples of 2^kmin; if
// sum m < 2^(kmin+52) every partial sum of every subset in every order is exactly representable
// in a double.  One wave
// per node; the criterion is evaluated for the four metrics of Utils::getMetric.
__device__ inline void safety_of(float m, int &kmin, double &tot, bool &bad) {
    const uint32_t u = __float_as_uint(m);
    const uint32_t ex = (u >> 23) & 0xFFu, man = u & 0x7FFFFFu;
    if (ex == 0xFFu) { bad = true; return; }          // inf / NaN
    if (ex == 0 && man == 0) return;                  // +-0 contributes nothing
    if (u >> 31) { bad = true; return; }              // negative: prefix sums would not be monotone
    const uint32_t full = ex ? (man | 0x800000u) : man;
    const int e = (ex ? (int) ex : 1) - 127 - 23;      // value = full * 2^e
    const int k = e + (__ffs((int) full) - 1);
    kmin = min(kmin, k);
    tot += fabs((double) m);
}

__global__ __launch_bounds__(256) void node_flags_kernel(uint32_t n_nodes, const uint32_t *__restrict__ row_ptr,
                                                         const uint2 *__restrict__ ent_es,
                                                         const uint2 *__restrict__ ent_os,
                                                         uint8_t *__restrict__ node_flags,
                                                         uint32_t *__restrict__ unsafe_cnt,
                                                         uint32_t *__restrict__ max_deg) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    uint32_t local_max = 0;
    for (uint32_t node = wave; node < n_nodes; node += n_waves) {
        const uint32_t r0 = row_ptr[node], r1 = row_ptr[node + 1];
        local_max = max(local_max, r1 - r0);
        int kmin[4] = {1 << 20, 1 << 20, 1 << 20, 1 << 20};
        double tot[4] = {0, 0, 0, 0};
        bool bad[4] = {false, false, false, false};
        for (uint32_t i = r0 + lane; i < r1; i += 64) {
            const float es = __uint_as_float(ent_es[i].y), os = __uint_as_float(ent_os[i].y);
            safety_of(es, kmin[TLM_METRIC_ES], tot[TLM_METRIC_ES], bad[TLM_METRIC_ES]);
            safety_of(os, kmin[TLM_METRIC_OS], tot[TLM_METRIC_OS], bad[TLM_METRIC_OS]);
            safety_of(sqrtf(es), kmin[TLM_METRIC_ES_SQRT], tot[TLM_METRIC_ES_SQRT], bad[TLM_METRIC_ES_SQRT]);
            safety_of(sqrtf(os), kmin[TLM_METRIC_OS_SQRT], tot[TLM_METRIC_OS_SQRT], bad[TLM_METRIC_OS_SQRT]);
        }
        uint8_t flags = 0;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            int k = kmin[m];
            double t = tot[m];
            int b = bad[m];
            for (int d = 32; d > 0; d >>= 1) {
                k = min(k, __shfl_xor(k, d));
                t += __shfl_xor(t, d);
                b |= __shfl_xor(b, d);
            }
            // t is itself rounded; 2^(k+52) leaves a factor 2 of slack for that
            const bool safe = !b && (k == (1 << 20) || t < ldexp(1.0, k + 52));
            if (safe) flags |= (uint8_t) (1u << m);
            else if (lane == 0) atomicAdd(&unsafe_cnt[m], 1u);
        }
        if (lane == 0) node_flags[node] = flags;
    }
    for (int d = 32; d > 0; d >>= 1) local_max = max(local_max, (uint32_t) __shfl_xor((int) local_max, d));
    if (lane == 0) atomicMax(max_deg, local_max);
}

// step 6b: DET units of an anchor = first edges up to and including the first one that leads to an
// anchor (PathManager.cpp:204,224-229)
__global__ void det_count_kernel(uint32_t n_anchors, const uint32_t *__restrict__ anchors,
                                 const uint32_t *__restrict__ row_ptr, const uint2 *__restrict__ ent,
                                 uint64_t *__restrict__ cnt) {
    uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_anchors) return;
    const uint32_t node = anchors[a];
    uint64_t c = 0;
    for (uint32_t i = row_ptr[node]; i < row_ptr[node + 1]; i++) {
        c++;
        if (ent[i].x >> 31) break;
    }
    cnt[a] = c;
}

// ---- locality renumbering for the walk layout ---------------------------------------------
// Walks start at anchors and move through the anchors' neighbourhoods, testing the visited flag of every
// neighbour of the current node.  With the reference's first-seen numbering those flags are scattered over
// the whole N-bit set; numbering the nodes by (nearest anchor, BFS level from it, reference index) puts the
// neighbours of a node into a window of a few hundred ids, so a wave's 64 probes fall into one or two cache
// lines.  The renumbering is internal to the walk layout: every result is reported in reference indices.
// Which entries the numbering follows (round 4).  Real overlap sets (minimap2 on real reads, reference run.sh:38-45) hold
// what the analytic friendly sets do not: INTERNAL matches (proper_sides above), which join loci that are megabases apart;
// a BFS that follows them reaches the whole graph in half a dozen levels and the numbering loses every locality (hostile
// config: 99 % of the Monte-Carlo steps probed outside the LDS band, a band move every 3.7 steps, 87 % of the steps on
// 8-byte rows).  So the BFS follows PROPER overlaps only (proper[e], one byte per entry, written by gather_kernel from the
// record it copies anyway), and a node most of whose list is internal matches (a read inside a repeat family, a chimeric
// read) is not expanded while anything else is left to expand.  A numbering heuristic only: every entry is walked as
// before, results do not depend on it.

// proper[] of a graph adopted as CSR (tlm_graph_from_csr with the cold columns): from the entry arrays
__global__ __launch_bounds__(256) void proper_flags_kernel(uint32_t n_nodes, const uint32_t *__restrict__ row_ptr,
                                                           const uint2 *__restrict__ ents, const uint32_t *__restrict__ q_start,
                                                           const uint32_t *__restrict__ q_end, const uint32_t *__restrict__ t_start,
                                                           const uint32_t *__restrict__ t_end, const uint8_t *__restrict__ strand,
                                                           const uint32_t *__restrict__ node_len, uint8_t *__restrict__ proper,
                                                           uint32_t *__restrict__ n_internal) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t n = wave; n < n_nodes; n += n_waves) {        // in the owner's list t_index == owner (OverlapGraph.cpp:243-248)
        const uint32_t r0 = row_ptr[n], r1 = row_ptr[n + 1];
        bool internal = false;
        for (uint32_t e = r0 + lane; e < r1; e += 64) {
            const bool ok = proper_overlap(q_start[e], q_end[e], node_len[ents[e].x & 0x7FFFFFFFu], t_start[e], t_end[e], node_len[n],
                                           strand && strand[e]);
            proper[e] = ok ? 1 : 0;
            internal |= !ok;
        }
        if (__ballot(internal) && lane == 0) atomicOr(n_internal, 1u);
    }
}

// suspect[n] = 1: more than a quarter of n's list are internal matches (never an anchor: the BFS starts there)
__global__ __launch_bounds__(256) void suspect_kernel(uint32_t n_nodes, const uint32_t *__restrict__ row_ptr,
                                                      const uint8_t *__restrict__ proper, const uint32_t *__restrict__ n_internal,
                                                      const uint8_t *__restrict__ node_anchor, uint8_t *__restrict__ suspect) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    if (*n_internal == 0u) return;           // (nothing to classify: bfs_expand_kernel does not look at `suspect` then)
    for (uint32_t n = wave; n < n_nodes; n += n_waves) {
        const uint32_t r0 = row_ptr[n], deg = row_ptr[n + 1] - r0;
        uint32_t internal = 0;
        for (uint32_t j = lane; j < deg; j += 64) internal += proper[r0 + j] ? 0u : 1u;
        for (int d = 32; d > 0; d >>= 1) internal += __shfl_xor(internal, d);
        if (lane == 0) suspect[n] = (!node_anchor[n] && 4u * internal > deg) ? 1 : 0;
    }
}

__global__ void bfs_init_kernel(uint32_t n_anchors, const uint32_t *__restrict__ anchors, unsigned long long *__restrict__ key,
                                uint32_t *__restrict__ frontier) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_anchors) return;
    key[anchors[k]] = (unsigned long long) k;          // level 0, source k
    frontier[k] = anchors[k];
}

// One level.  The size of the frontier is read from device memory (counts[0]; the next level's goes to counts[1]): the
// host queues a batch of levels without waiting for any of them (a level past the last one finds an empty frontier).
// `proper` null (a graph adopted as CSR without the cold columns): every entry is followed, nothing is deferred.
__global__ __launch_bounds__(256) void bfs_expand_kernel(uint32_t *__restrict__ counts, const uint32_t *__restrict__ frontier,
                                                         uint32_t level, const uint32_t *__restrict__ row_ptr,
                                                         const uint2 *__restrict__ ents, const uint8_t *__restrict__ proper,
                                                         const uint32_t *__restrict__ n_internal,
                                                         const uint8_t *__restrict__ suspect, unsigned long long *__restrict__ key,
                                                         uint32_t *__restrict__ next, uint32_t *__restrict__ deferred,
                                                         uint32_t *__restrict__ n_deferred) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n_f = counts[0];
    uint32_t *n_next = counts + 1;
    const bool classify = proper != nullptr && *n_internal != 0u;      // (a graph of proper overlaps only: every entry, as before)
    for (uint32_t f = wave; f < n_f; f += n_waves) {
        const uint32_t u = frontier[f];
        const unsigned long long nk = ((unsigned long long) (level + 1) << 32) | (key[u] & 0xFFFFFFFFull);
        for (uint32_t e = row_ptr[u] + lane; e < row_ptr[u + 1]; e += 64) {
            if (classify && !proper[e]) continue;
            const uint32_t v = ents[e].x & 0x7FFFFFFFu;
            // the minimum over all discoverers of v in this level is deterministic; exactly one of them sees "unseen"
            const unsigned long long old = atomicMin(&key[v], nk);
            if (old == ~0ull) {
                if (classify && suspect[v]) deferred[atomicAdd(n_deferred, 1u)] = v;      // expanded only when nothing else is left
                else next[atomicAdd(n_next, 1u)] = v;
            }
        }
    }
}

__global__ void bfs_sortkey_kernel(uint32_t n_nodes, const unsigned long long *__restrict__ key,
                                   unsigned long long *__restrict__ sortkey, uint32_t *__restrict__ ids) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const unsigned long long k = key[i];
    // (source anchor, level); nodes no anchor reaches keep ~0 and sort last (no walk ever visits them)
    sortkey[i] = k == ~0ull ? ~0ull : ((k & 0xFFFFFFFFull) << 32) | (k >> 32);
    ids[i] = i;
}

// number of names that occur in a kept record = first index of the sorted first positions holding 0xFFFFFFFF
__global__ void count_seen_kernel(uint32_t n_names, const uint32_t *__restrict__ sorted_pos, uint32_t *__restrict__ out) {
    uint32_t lo = 0, hi = n_names;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (sorted_pos[mid] == 0xFFFFFFFFu) hi = mid;
        else lo = mid + 1;
    }
    *out = lo;
}

__global__ void anchor_walk_ids_kernel(uint32_t n_anchors, const uint32_t *__restrict__ anchors,
                                       const uint32_t *__restrict__ new_of_old, uint32_t *__restrict__ w_anchors) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_anchors) w_anchors[k] = new_of_old[anchors[k]];
}

__global__ void iota_perm_kernel(uint32_t n_nodes, uint32_t *__restrict__ old_of_new) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_nodes) old_of_new[i] = i;
}

__global__ void invert_perm_kernel(uint32_t n_nodes, const uint32_t *__restrict__ old_of_new, uint32_t *__restrict__ new_of_old) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_nodes) new_of_old[old_of_new[i]] = i;
}

// Padded length of a row of the walk layout: a multiple of 8 entries (rows start on 64-byte boundaries, a lane of the
// walk kernel owns 4 or 8 consecutive entries), and of 24 for the degrees the kernel reads with 6 entries per lane
// (257..384): whatever the lane width, the last lane's entries end inside the row's own padding.
constexpr uint32_t WALK_ROW0 = 256;      // the walk layout's arrays start with this many zero entries; the first row follows
__host__ __device__ inline uint32_t tlm_padded_degree(uint32_t deg) {
    // lists read 6 or 12 entries per lane (257..384, 513..768): a lane's entries must not reach the next row
    return ((deg > 256u && deg <= 384u) || (deg > 512u && deg <= 768u)) ? ((deg + 23u) / 24u) * 24u : (deg + 7u) & ~7u;
}

// walk layout: padded degrees in walk order, then (after an exclusive scan) the per-node record and the row copies
__global__ void padded_degree_kernel(uint32_t n_nodes, const uint32_t *__restrict__ row_ptr,
                                     const uint32_t *__restrict__ old_of_new, uint32_t *__restrict__ pd) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_nodes) {
        const uint32_t o = old_of_new[i];
        pd[i] = tlm_padded_degree(row_ptr[o + 1] - row_ptr[o]);
    }
}

__global__ __launch_bounds__(256) void walk_layout_kernel(uint32_t n_nodes, const uint32_t *__restrict__ row_ptr,
                                                          const uint32_t *__restrict__ old_of_new,
                                                          const uint32_t *__restrict__ new_of_old,
                                                          const uint32_t *__restrict__ w_start,
                                                          const uint8_t *__restrict__ node_flags,
                                                          const uint2 *__restrict__ ent_es,
                                                          const uint2 *__restrict__ ent_os,
                                                          const uint32_t *__restrict__ q_start,
                                                          const uint32_t *__restrict__ t_start, uint2 *__restrict__ w_row,
                                                          uint2 *__restrict__ w_ent_es, uint2 *__restrict__ w_ent_os,
                                                          uint2 *__restrict__ w_qt, uint32_t *__restrict__ w_m_es,
                                                          uint32_t *__restrict__ w_m_os, uint16_t *__restrict__ w_d16) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t node = wave; node < n_nodes; node += n_waves) {     // node = index in walk order
        const uint32_t o = old_of_new[node];
        const uint32_t r0 = row_ptr[o], deg = row_ptr[o + 1] - r0, w0 = w_start[node];
        const uint32_t pdeg = tlm_padded_degree(deg);
        bool anchor_nbr = false;        // does the list hold an anchor?  (bit 28: the forced-anchor test can be skipped)
        bool far_nbr = false;           // a neighbour further than 32767 ids away (walk order): no 16-bit relative id
        for (uint32_t j = lane; j < deg; j += 64) {
            const uint32_t x = ent_es[r0 + j].x;
            anchor_nbr |= (x >> 31) != 0;
            const uint32_t rel = new_of_old[x & 0x7FFFFFFFu] - node + 32768u;
            far_nbr |= rel > 65535u;
        }
        // bit 29: the row may be read in its compact form (metric + 16-bit relative id, no anchor among the neighbours)
        // (a node without a list has no row of its own: a lane would read its successor's relative ids against the wrong node)
#ifdef TLM_ABLATE_NOCOMPACT     /* timing-only build (results stay right): every step reads 8-byte rows */
        const bool compact = false;
        (void) far_nbr;
#else
        const bool compact = deg != 0 && !__ballot(anchor_nbr) && !__ballot(far_nbr);
#endif
        const uint32_t fl = (uint32_t) node_flags[o] | (__ballot(anchor_nbr) ? 16u : 0u) | (compact ? 32u : 0u);
        if (lane == 0) w_row[node] = make_uint2(w0, deg | (fl << 24));
        for (uint32_t j = lane; j < pdeg; j += 64) {
            const bool real = j < deg;
            // padding names the row's own node: visited whenever a Monte-Carlo walk stands on it, so the walk kernel
            // needs no validity test per entry (no anchor bit, metric 0)
            uint2 es = make_uint2(node, 0u), os = make_uint2(node, 0u);
            if (real) {
                es = ent_es[r0 + j];
                os = ent_os[r0 + j];
                const uint32_t w = new_of_old[es.x & 0x7FFFFFFFu] | (es.x & 0x80000000u);   // list order is kept
                es.x = w;
                os.x = w;
            }
            w_ent_es[w0 + j] = es;
            w_ent_os[w0 + j] = os;
            w_m_es[w0 + j] = es.y;
            w_m_os[w0 + j] = os.y;
            w_d16[w0 + j] = (uint16_t) (compact ? (es.x & 0x7FFFFFFFu) - node + 32768u : 32768u);
            w_qt[w0 + j] = real ? make_uint2(q_start[r0 + j], t_start[r0 + j]) : make_uint2(0u, 0u);
        }
    }
}

// OverlapGraph::Edge records of given (node, slot) pairs (tlm_graph_edges)
__global__ void edges_kernel(uint64_t n, uint32_t n_nodes, const uint32_t *__restrict__ nodes, const uint32_t *__restrict__ slots,
                             const uint32_t *__restrict__ row_ptr, const uint2 *__restrict__ ent_es,
                             const uint2 *__restrict__ ent_os, const uint32_t *__restrict__ q_start,
                             const uint32_t *__restrict__ q_end, const uint32_t *__restrict__ t_start,
                             const uint32_t *__restrict__ t_end, const float *__restrict__ si,
                             const uint8_t *__restrict__ strand, tlm_edge *__restrict__ out, uint32_t *__restrict__ bad) {
    const uint64_t k = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t node = nodes[k];
    if (node >= n_nodes || slots[k] >= row_ptr[node + 1] - row_ptr[node]) {
        atomicAdd(bad, 1u);
        return;
    }
    const uint64_t e = (uint64_t) row_ptr[node] + slots[k];
    tlm_edge o;
    o.q_index = ent_es[e].x & 0x7FFFFFFFu;
    o.t_index = node;
    o.q_start = q_start[e]; o.q_end = q_end[e]; o.t_start = t_start[e]; o.t_end = t_end[e];
    o.overlap_score = __uint_as_float(ent_os[e].y);
    o.sequence_identity = si[e];
    o.extension_score = __uint_as_float(ent_es[e].y);
    o.relative_strand = strand[e];
    out[k] = o;
}

// OverlapGraph::stats (OverlapGraph.cpp:24-95): extrema over nodes and adjacency entries.  Floats are compared through
// an order-preserving u32 image; NaNs never replace an extremum in the reference's `if (min > x)` / `if (max < x)`
// form and are skipped here.  slots: 0 min_len 1 max_len 2 min_con 3 max_con 4 min_ol 5 max_ol 6..11 min/max OS, ES, SI
// (u32 images), 12 anchors
__device__ __forceinline__ uint32_t f32_ord(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ __launch_bounds__(256) void stats_nodes_kernel(uint32_t n_nodes, const uint32_t *__restrict__ row_ptr,
                                                          const uint32_t *__restrict__ node_len,
                                                          const uint8_t *__restrict__ node_anchor,
                                                          unsigned long long *__restrict__ acc) {
    unsigned long long mn_l = ~0ull, mx_l = 0, mn_c = ~0ull, mx_c = 0, an = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes; i += (uint64_t) gridDim.x * blockDim.x) {
        const unsigned long long l = node_len[i], c = row_ptr[i + 1] - row_ptr[i];
        mn_l = min(mn_l, l); mx_l = max(mx_l, l); mn_c = min(mn_c, c); mx_c = max(mx_c, c);
        an += node_anchor[i] != 0;
    }
    atomicMin(&acc[0], mn_l); atomicMax(&acc[1], mx_l); atomicMin(&acc[2], mn_c); atomicMax(&acc[3], mx_c);
    atomicAdd(&acc[12], an);
}

__global__ __launch_bounds__(256) void stats_edges_kernel(uint64_t n_adj, const uint2 *__restrict__ ent_es,
                                                          const uint2 *__restrict__ ent_os, const uint32_t *__restrict__ q_start,
                                                          const uint32_t *__restrict__ q_end, const uint32_t *__restrict__ t_start,
                                                          const uint32_t *__restrict__ t_end, const float *__restrict__ si,
                                                          unsigned long long *__restrict__ acc) {
    unsigned long long mn_ol = ~0ull, mx_ol = 0;
    uint32_t mn[3] = {~0u, ~0u, ~0u}, mx[3] = {0u, 0u, 0u};
    for (uint64_t e = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; e < n_adj; e += (uint64_t) gridDim.x * blockDim.x) {
        // int ol = std::max(unsigned, unsigned); compared with and stored into ulong extrema (:65-71)
        const int ol = (int) max(q_end[e] - q_start[e], t_end[e] - t_start[e]);
        const unsigned long long olu = (unsigned long long) (long long) ol;
        mn_ol = min(mn_ol, olu); mx_ol = max(mx_ol, olu);
        const float v[3] = {__uint_as_float(ent_os[e].y), __uint_as_float(ent_es[e].y), si[e]};
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (v[k] == v[k]) {
                const uint32_t o = f32_ord(v[k] == 0.0f ? 0.0f : v[k]);
                mn[k] = min(mn[k], o);
                mx[k] = max(mx[k], o);
            }
    }
    atomicMin(&acc[4], mn_ol); atomicMax(&acc[5], mx_ol);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        atomicMin(&acc[6 + 2 * k], (unsigned long long) mn[k]);
        atomicMax(&acc[7 + 2 * k], (unsigned long long) mx[k]);
    }
}

unsigned grid_for(uint64_t n) {
    uint64_t b = (n + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    return (unsigned) (b ? b : 1);
}

}  // namespace

const char *const BUILD_PHASE_NAMES[TLM_BUILD_PHASES] = {
    "filter_score", "rank_scan", "first_seen", "name_sort+nodes", "entries+owner_sort", "gather", "node_flags", "bfs",
    "order_sort", "walk_layout", "det_table", "other"};

extern "C" const char *tlm_build_phase_name(int phase) {
    return phase >= 0 && phase < TLM_BUILD_PHASES ? BUILD_PHASE_NAMES[phase] : "";
}

static int graph_finish(tlm_ctx *ctx, tlm_graph *g, Phases &ph) {
    hipStream_t s = ctx->stream;
    t_build_stream = s;
    const uint32_t N = g->n_nodes;
    if (int rc = tlm_pool_alloc((void **) &g->node_flags, N ? N : 1, s)) return rc;
    DevBuf cnts;
    if (int rc = dev_alloc(cnts, 8 * sizeof(uint32_t))) return rc;
    TLM_HIP(hipMemsetAsync(cnts.p, 0, 8 * sizeof(uint32_t), s));
    if (N) {
        hipLaunchKernelGGL(node_flags_kernel, dim3(grid_for((uint64_t) N * 64)), dim3(256), 0, s, N, g->row_ptr,
                           g->ent_es, g->ent_os, g->node_flags, cnts.as<uint32_t>(), cnts.as<uint32_t>() + 4);
    }
    uint32_t *const h_cnts = (uint32_t *) ctx->pinned_small;
    TLM_HIP(hipMemcpyAsync(h_cnts, cnts.p, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    // the anchors, ascending: a stable select over 0..N-1 by the anchor flags, on the device (the host used to fetch all N
    // flags into pageable memory -- a blocking copy in the middle of the build -- and pick them out itself)
    DevBuf sel, n_sel;
    if (int rc = dev_alloc(sel, ((size_t) N + 1) * 4)) return rc;
    if (int rc = dev_alloc(n_sel, 8)) return rc;
    TLM_HIP(hipMemsetAsync(n_sel.p, 0, 8, s));
    if (N) {
        size_t tb = 0;
        TLM_HIP(rocprim::select(nullptr, tb, rocprim::counting_iterator<uint32_t>(0), g->node_anchor, sel.as<uint32_t>(),
                                n_sel.as<uint32_t>(), (size_t) N, s));
        DevBuf tmp;
        if (int rc = dev_alloc(tmp, tb)) return rc;
        TLM_HIP(rocprim::select(tmp.p, tb, rocprim::counting_iterator<uint32_t>(0), g->node_anchor, sel.as<uint32_t>(),
                                n_sel.as<uint32_t>(), (size_t) N, s));
    }
    TLM_HIP(hipMemcpyAsync(h_cnts + 8, n_sel.p, 4, hipMemcpyDeviceToHost, s));
    ph.mark(PH_FLAGS);
    if (int rc = ph.sync()) return rc;
    for (int m = 0; m < 4; m++) g->info.unsafe_nodes[m] = h_cnts[m];
    g->info.max_degree = h_cnts[4];
    TLM_CHECK(h_cnts[4] < (1u << 24), TLM_ELIMIT, "a node of degree %u exceeds the 2^24 layout limit", h_cnts[4]);
    const uint32_t na = h_cnts[8];
    g->h_anchors.assign(na, 0);
    if (na) {       // (a few hundred words; the walk calls need them on the host for their unit arithmetic)
        TLM_HIP(hipMemcpyAsync(g->h_anchors.data(), sel.p, (size_t) na * 4, hipMemcpyDeviceToHost, s));
        if (int rc = ph.sync()) return rc;
    }
    g->info.n_anchors = na;
    g->info.n_nodes = N;
    g->info.n_adj = g->n_adj;
    if (int rc = tlm_pool_alloc((void **) &g->anchors, (na ? na : 1) * sizeof(uint32_t), s)) return rc;
    if (int rc = tlm_pool_alloc((void **) &g->w_anchors, (na ? na : 1) * sizeof(uint32_t), s)) return rc;
    if (int rc = tlm_pool_alloc((void **) &g->det_off, ((size_t) na + 1) * sizeof(uint64_t), s)) return rc;
    g->h_det_off.assign((size_t) na + 1, 0);
    if (na) TLM_HIP(hipMemcpyAsync(g->anchors, sel.p, na * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));

    // ---- walk order: (nearest anchor, BFS level, reference index); identity if TLM_NO_LOCALIZE is set
    if (int rc = tlm_pool_alloc((void **) &g->w_old_of_new, ((size_t) N + 1) * 4, s)) return rc;
    DevBuf new_of_old;
    if (int rc = dev_alloc(new_of_old, ((size_t) N + 1) * 4)) return rc;
    const bool localize = N && na && !std::getenv("TLM_NO_LOCALIZE");
    g->info.bfs_levels = 0;
    if (N) hipLaunchKernelGGL(iota_perm_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, g->w_old_of_new);
    ph.mark(PH_OTHER);
    if (localize) {
        constexpr uint32_t BATCH = 16;      // levels queued per host round trip
        DevBuf key, f0, f1, cnt, sortkey, sortkey_out, ids, suspect, deferred, stall;
        // the numbering follows proper overlaps and defers nodes whose lists are mostly internal matches (see proper_entry)
        const bool classify = g->q_end && g->t_end && !std::getenv("TLM_ORDER_ALL_EDGES");
        DevBuf proper_made, internal_made;
        const uint8_t *proper = nullptr;
        const uint32_t *n_internal = nullptr;
        if (classify && g->proper_tmp) {
            proper = g->proper_tmp;         // written by gather_kernel, with the "any internal match at all?" word behind it
            n_internal = g->n_internal_tmp;
        } else if (classify) {
            if (int rc = dev_alloc(proper_made, (size_t) g->n_adj + 1)) return rc;
            if (int rc = dev_alloc(internal_made, 4)) return rc;
            TLM_HIP(hipMemsetAsync(internal_made.p, 0, 4, s));
            hipLaunchKernelGGL(proper_flags_kernel, dim3(grid_for(std::min<uint64_t>((uint64_t) N * 64, 1u << 20))), dim3(256), 0, s, N,
                               g->row_ptr, g->ent_es, g->q_start, g->q_end, g->t_start, g->t_end, g->strand, g->node_len,
                               proper_made.as<uint8_t>(), internal_made.as<uint32_t>());
            proper = proper_made.as<uint8_t>();
            n_internal = internal_made.as<uint32_t>();
        }
        if (int rc = dev_alloc(suspect, (size_t) N)) return rc;
        if (int rc = dev_alloc(deferred, (size_t) N * 4)) return rc;
        if (int rc = dev_alloc(stall, 16)) return rc;
        TLM_HIP(hipMemsetAsync(stall.p, 0, 16, s));
        if (classify)
            hipLaunchKernelGGL(suspect_kernel, dim3(grid_for(std::min<uint64_t>((uint64_t) N * 64, 1u << 20))), dim3(256), 0, s, N,
                               g->row_ptr, proper, n_internal, g->node_anchor, suspect.as<uint8_t>());
        if (int rc = dev_alloc(key, (size_t) N * 8)) return rc;
        if (int rc = dev_alloc(f0, (size_t) N * 4)) return rc;
        if (int rc = dev_alloc(f1, (size_t) N * 4)) return rc;
        if (int rc = dev_alloc(cnt, (BATCH + 1) * 4)) return rc;
        if (int rc = dev_alloc(sortkey, (size_t) N * 8)) return rc;
        if (int rc = dev_alloc(sortkey_out, (size_t) N * 8)) return rc;
        if (int rc = dev_alloc(ids, (size_t) N * 4)) return rc;
        TLM_HIP(hipMemsetAsync(key.p, 0xFF, (size_t) N * 8, s));
        hipLaunchKernelGGL(bfs_init_kernel, dim3((na + 255) / 256), dim3(256), 0, s, na, g->anchors,
                           key.as<unsigned long long>(), f0.as<uint32_t>());
        uint32_t level = 0;
        static_assert(128 + (2 * (BATCH + 1) + 1) * 4 <= TLM_PINNED_SMALL_BYTES, "the BFS counters fit the pinned words");
        uint32_t *const h_cnt = (uint32_t *) ((char *) ctx->pinned_small + 128);        // up: this batch's initial counters
        uint32_t *const h_got = h_cnt + (BATCH + 1);                                    // down: what the batch counted
        uint32_t *const h_deferred = h_got + (BATCH + 1);                               // down: nodes waiting in `deferred`
        *h_deferred = 0u;
        constexpr size_t cnt_bytes = (BATCH + 1) * sizeof(uint32_t);
        std::memset(h_cnt, 0, cnt_bytes);
        h_cnt[0] = na;
        uint32_t *fr[2] = {f0.as<uint32_t>(), f1.as<uint32_t>()};
        // every level of a batch runs the same grid: one wave per frontier node, strided
        const unsigned bfs_grid = grid_for(std::min<uint64_t>((uint64_t) N * 64, 1u << 20));
        for (bool done = false; !done;) {
            TLM_HIP(hipMemcpyAsync(cnt.p, h_cnt, cnt_bytes, hipMemcpyHostToDevice, s));     // {frontier size, 0, ..., 0}
            for (uint32_t j = 0; j < BATCH; j++) {
                hipLaunchKernelGGL(bfs_expand_kernel, dim3(bfs_grid), dim3(256), 0, s, cnt.as<uint32_t>() + j,
                                   fr[(level + j) & 1u], level + j, g->row_ptr, g->ent_es, proper, n_internal, suspect.as<uint8_t>(),
                                   key.as<unsigned long long>(), fr[(level + j + 1) & 1u], deferred.as<uint32_t>(),
                                   stall.as<uint32_t>());
            }
            TLM_HIP(hipMemcpyAsync(h_got, cnt.p, cnt_bytes, hipMemcpyDeviceToHost, s));
            if (classify) TLM_HIP(hipMemcpyAsync(h_deferred, stall.p, 4, hipMemcpyDeviceToHost, s));
            if (int rc = ph.sync()) return rc;
            uint32_t j = 1;
            while (j <= BATCH && h_got[j] != 0) j++;        // level + j - 1 = the last level with a frontier
            done = j <= BATCH;
            level += done ? j : BATCH;
            TLM_CHECK(level <= N + BATCH, TLM_EINTERNAL, "BFS did not terminate");
            uint32_t carry = h_got[BATCH];
            if (done && classify && *h_deferred != 0u) {
                // the frontier has died with deferred nodes waiting (a stall: once or twice per hostile graph, never on a
                // graph of proper overlaps): they are the next frontier.  Decided here, where the host reads the counters
                // anyway -- two more kernels per level for this cost 8 ms on the 47 levels of the headline graph
                carry = *h_deferred;
                TLM_HIP(hipMemcpyAsync(fr[level & 1u], deferred.p, (size_t) carry * 4, hipMemcpyDeviceToDevice, s));
                TLM_HIP(hipMemsetAsync(stall.p, 0, 4, s));
                done = false;
            }
            std::memset(h_cnt, 0, cnt_bytes);
            h_cnt[0] = carry;
        }
        g->info.bfs_levels = level;
        ph.mark(PH_BFS);
        hipLaunchKernelGGL(bfs_sortkey_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, key.as<unsigned long long>(),
                           sortkey.as<unsigned long long>(), ids.as<uint32_t>());
        size_t tb = 0;
        TLM_HIP(rocprim::radix_sort_pairs(nullptr, tb, sortkey.as<unsigned long long>(), sortkey_out.as<unsigned long long>(),
                                          ids.as<uint32_t>(), g->w_old_of_new, (size_t) N, 0, 64, s));
        DevBuf tmp;
        if (int rc = dev_alloc(tmp, tb)) return rc;
        TLM_HIP(rocprim::radix_sort_pairs(tmp.p, tb, sortkey.as<unsigned long long>(), sortkey_out.as<unsigned long long>(),
                                          ids.as<uint32_t>(), g->w_old_of_new, (size_t) N, 0, 64, s));
        ph.mark(PH_ORDER_SORT);
    }
    if (N) hipLaunchKernelGGL(invert_perm_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, g->w_old_of_new,
                              new_of_old.as<uint32_t>());
    if (na) hipLaunchKernelGGL(anchor_walk_ids_kernel, dim3((na + 255) / 256), dim3(256), 0, s, na, g->anchors,
                               new_of_old.as<uint32_t>(), g->w_anchors);

    // ---- walk layout
    DevBuf det_cnt;
    std::vector<uint64_t> h_det_cnt(na);
    {
        DevBuf pd, ws;
        if (int rc = dev_alloc(pd, ((size_t) N + 1) * 4)) return rc;
        if (int rc = dev_alloc(ws, ((size_t) N + 1) * 4)) return rc;
        uint64_t apad = 0;
        if (N) {
            TLM_CHECK(g->n_adj + 23ull * N < (1ull << 32) - 512, TLM_ELIMIT, "graph too large for the padded u32 walk layout");
            hipLaunchKernelGGL(padded_degree_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, g->row_ptr, g->w_old_of_new,
                               pd.as<uint32_t>());
            size_t tb = 0;
            TLM_HIP(rocprim::exclusive_scan(nullptr, tb, pd.as<uint32_t>(), ws.as<uint32_t>(), WALK_ROW0, (size_t) N,
                                            rocprim::plus<uint32_t>(), s));
            DevBuf tmp;
            if (int rc = dev_alloc(tmp, tb)) return rc;
            TLM_HIP(rocprim::exclusive_scan(tmp.p, tb, pd.as<uint32_t>(), ws.as<uint32_t>(), WALK_ROW0, (size_t) N,
                                            rocprim::plus<uint32_t>(), s));
            uint32_t *const h_last = (uint32_t *) ((char *) ctx->pinned_small + 64);
            TLM_HIP(hipMemcpyAsync(h_last, ws.as<uint32_t>() + (N - 1), 4, hipMemcpyDeviceToHost, s));
            TLM_HIP(hipMemcpyAsync(h_last + 1, pd.as<uint32_t>() + (N - 1), 4, hipMemcpyDeviceToHost, s));
            // the DET unit table (reference layout) rides on the same round trip
            if (na) {
                if (int rc = dev_alloc(det_cnt, na * sizeof(uint64_t))) return rc;
                hipLaunchKernelGGL(det_count_kernel, dim3((na + 63) / 64), dim3(64), 0, s, na, g->anchors, g->row_ptr,
                                   g->ent_es, det_cnt.as<uint64_t>());
                TLM_HIP(hipMemcpyAsync(h_det_cnt.data(), det_cnt.p, na * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
            }
            if (int rc = ph.sync()) return rc;
            apad = (uint64_t) h_last[0] + h_last[1];
        }
        g->n_adj_pad = apad;
        for (uint32_t a = 0; a < na; a++) g->h_det_off[a + 1] = g->h_det_off[a] + h_det_cnt[a];
        TLM_HIP(hipMemcpyAsync(g->det_off, g->h_det_off.data(), ((size_t) na + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        ph.mark(PH_DET_TABLE);
        if (int rc = tlm_pool_alloc((void **) &g->w_row, ((size_t) N + 1) * sizeof(uint2), s)) return rc;
        if (int rc = tlm_pool_alloc((void **) &g->w_ent_es, (apad + 256) * sizeof(uint2), s)) return rc;
        if (int rc = tlm_pool_alloc((void **) &g->w_ent_os, (apad + 256) * sizeof(uint2), s)) return rc;
        if (int rc = tlm_pool_alloc((void **) &g->w_qt, (apad + 256) * sizeof(uint2), s)) return rc;
        if (int rc = tlm_pool_alloc((void **) &g->w_m_es, (apad + 256) * 4, s)) return rc;
        if (int rc = tlm_pool_alloc((void **) &g->w_m_os, (apad + 256) * 4, s)) return rc;
        if (int rc = tlm_pool_alloc((void **) &g->w_d16, (apad + 256) * 2, s)) return rc;
        TLM_HIP(hipMemsetAsync(g->w_m_es + apad, 0, 256 * 4, s));
        TLM_HIP(hipMemsetAsync(g->w_m_os + apad, 0, 256 * 4, s));
        // relative id 32768 = "the row's own node": whatever a lane reads past the last row stays a valid node id
        TLM_HIP(hipMemsetD16Async((hipDeviceptr_t) (g->w_d16 + apad), 0x8000, 256, s));
        // the block of zero entries in front of the first row (lanes past the end of a compact list read it)
        TLM_HIP(hipMemsetAsync(g->w_m_es, 0, WALK_ROW0 * 4, s));
        TLM_HIP(hipMemsetAsync(g->w_m_os, 0, WALK_ROW0 * 4, s));
        TLM_HIP(hipMemsetD16Async((hipDeviceptr_t) g->w_d16, 0x8000, WALK_ROW0, s));
        TLM_HIP(hipMemsetAsync(g->w_ent_es, 0, WALK_ROW0 * sizeof(uint2), s));
        TLM_HIP(hipMemsetAsync(g->w_ent_os, 0, WALK_ROW0 * sizeof(uint2), s));
        TLM_HIP(hipMemsetAsync(g->w_qt, 0, WALK_ROW0 * sizeof(uint2), s));
        // 256 zero entries of slack after the last row: a lane's 8-entry read never leaves the buffer
        TLM_HIP(hipMemsetAsync(g->w_ent_es + apad, 0, 256 * sizeof(uint2), s));
        TLM_HIP(hipMemsetAsync(g->w_ent_os + apad, 0, 256 * sizeof(uint2), s));
        TLM_HIP(hipMemsetAsync(g->w_qt + apad, 0, 256 * sizeof(uint2), s));
        if (N) {
            hipLaunchKernelGGL(walk_layout_kernel, dim3(grid_for((uint64_t) N * 64)), dim3(256), 0, s, N, g->row_ptr,
                               g->w_old_of_new, new_of_old.as<uint32_t>(), ws.as<uint32_t>(), g->node_flags, g->ent_es,
                               g->ent_os, g->q_start, g->t_start, g->w_row, g->w_ent_es, g->w_ent_os, g->w_qt, g->w_m_es,
                               g->w_m_os, g->w_d16);
        }
        ph.mark(PH_LAYOUT);
    }
    if (int rc = ph.sync()) return rc;       // the graph is complete when the call returns
    TLM_HIP(hipGetLastError());
    return TLM_OK;
}

int tlm_graph_finish(tlm_ctx *ctx, tlm_graph *g) {     // graphs adopted from a finished CSR (tlm_graph_from_csr)
    Phases ph(ctx, ctx->stream);
    const int rc = graph_finish(ctx, g, ph);
    if (rc == TLM_OK) ph.collect(g->info);
    return rc;
}

extern "C" int tlm_graph_kept_records(const tlm_graph *g, uint32_t *rec_host) {
    TLM_CHECK(g && (rec_host || g->info.n_kept == 0), TLM_EINVAL, "tlm_graph_kept_records: null argument");
    TLM_CHECK(g->kept_rec || g->info.n_kept == 0, TLM_EINVAL, "tlm_graph_kept_records: graph was not made by tlm_graph_build");
    TLM_HIP(hipSetDevice(g->device));
    if (g->info.n_kept) TLM_HIP(hipMemcpy(rec_host, g->kept_rec, g->info.n_kept * 4, hipMemcpyDeviceToHost));
    return TLM_OK;
}

extern "C" int tlm_graph_build_host(tlm_ctx *ctx, const tlm_filter_params *p, const tlm_records *files_host, int n_files,
                                    uint32_t n_names, const uint8_t *name_is_ctg_host, tlm_graph **out) {
    TLM_CHECK(ctx && p && files_host && out, TLM_EINVAL, "tlm_graph_build_host: null argument");
    TLM_CHECK(n_files >= 1 && n_files <= MAX_FILES, TLM_EINVAL, "tlm_graph_build_host: n_files must be 1..%d", MAX_FILES);
    TLM_HIP(hipSetDevice(ctx->device));
    std::vector<DevBuf> bufs((size_t) n_files * 10 + 1);
    tlm_records dev[MAX_FILES];
    size_t b = 0;
    auto up = [&](const void *src, size_t bytes, const void **dst) -> int {
        *dst = nullptr;
        if (!src) return TLM_OK;
        if (int rc = dev_alloc(bufs[b], bytes)) return rc;
        if (bytes) TLM_HIP(hipMemcpyAsync(bufs[b].p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        *dst = bufs[b].p;
        b++;
        return TLM_OK;
    };
    for (int i = 0; i < n_files; i++) {
        const tlm_records &h = files_host[i];
        dev[i] = h;
        const size_t n4 = (size_t) h.n * 4;
        if (int rc = up(h.q_id, n4, (const void **) &dev[i].q_id)) return rc;
        if (int rc = up(h.t_id, n4, (const void **) &dev[i].t_id)) return rc;
        if (int rc = up(h.q_len, n4, (const void **) &dev[i].q_len)) return rc;
        if (int rc = up(h.q_start, n4, (const void **) &dev[i].q_start)) return rc;
        if (int rc = up(h.q_end, n4, (const void **) &dev[i].q_end)) return rc;
        if (int rc = up(h.t_len, n4, (const void **) &dev[i].t_len)) return rc;
        if (int rc = up(h.t_start, n4, (const void **) &dev[i].t_start)) return rc;
        if (int rc = up(h.t_end, n4, (const void **) &dev[i].t_end)) return rc;
        if (int rc = up(h.matches, n4, (const void **) &dev[i].matches)) return rc;
        if (int rc = up(h.strand, (size_t) h.n, (const void **) &dev[i].strand)) return rc;
    }
    const uint8_t *ctg_dev = nullptr;
    if (int rc = up(name_is_ctg_host, n_names, (const void **) &ctg_dev)) return rc;
    TLM_HIP(hipStreamSynchronize(ctx->stream));
    return tlm_graph_build(ctx, p, dev, n_files, n_names, ctg_dev, out);
}

extern "C" int tlm_graph_build(tlm_ctx *ctx, const tlm_filter_params *p, const tlm_records *files, int n_files,
                               uint32_t n_names, const uint8_t *name_is_ctg_dev, tlm_graph **out) {
    TLM_CHECK(ctx && p && files && out, TLM_EINVAL, "tlm_graph_build: null argument");
    TLM_CHECK(n_files >= 1 && n_files <= MAX_FILES, TLM_EINVAL, "tlm_graph_build: n_files must be 1..%d", MAX_FILES);
    TLM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    t_build_stream = s;
    FileTable ft;
    std::memset(&ft, 0, sizeof(ft));
    ft.n_files = n_files;
    uint64_t R = 0;
    for (int i = 0; i < n_files; i++) {
        const tlm_records &r = files[i];
        TLM_CHECK(r.n == 0 || (r.q_id && r.t_id && r.q_len && r.q_start && r.q_end && r.t_len && r.t_start && r.t_end
                               && r.matches), TLM_EINVAL, "tlm_graph_build: file %d has a null column", i);
        ft.f[i] = FileCols{r.q_id, r.t_id, r.q_len, r.q_start, r.q_end, r.t_len, r.t_start, r.t_end, r.strand, R, r.n,
                           r.anchors};
        R += r.n;
    }
    TLM_CHECK(R < (1ull << 31), TLM_ELIMIT, "tlm_graph_build: %llu records exceed the 2^31 layout limit",
              (unsigned long long) R);
    TLM_CHECK(n_names < (1u << 31), TLM_ELIMIT, "tlm_graph_build: too many names");

    tlm_graph *g = new tlm_graph();
    g->ctx = ctx;
    g->device = ctx->device;
    (void) hipGetLastError();
    struct Guard {   // destroy the half-built graph on any early return
        tlm_graph *g;
        ~Guard() { if (g) tlm_graph_destroy(g); }
    } guard{g};

    TLM_HIP(hipEventRecord(ctx->ev0, s));
    Phases ph(ctx, s);
    // 1. filter + score
    DevBuf keep, r_si, r_os, r_es, krank, nk;
    if (int rc = dev_alloc(keep, R)) return rc;
    if (int rc = dev_alloc(r_si, R * 4)) return rc;
    if (int rc = dev_alloc(r_os, R * 4)) return rc;
    if (int rc = dev_alloc(r_es, R * 4)) return rc;
    if (int rc = dev_alloc(krank, (R + 1) * 4)) return rc;
    if (int rc = dev_alloc(nk, (MAX_FILES + 1) * sizeof(unsigned long long))) return rc;
    TLM_HIP(hipMemsetAsync(nk.p, 0, (MAX_FILES + 1) * sizeof(unsigned long long), s));
    ph.mark(PH_OTHER);
    for (int i = 0; i < n_files; i++) {
        const uint64_t b = ft.f[i].base;
        tlm_launch_filter_score(s, *p, files[i], keep.as<uint8_t>() + b, r_si.as<float>() + b, r_os.as<float>() + b,
                                r_es.as<float>() + b, nk.as<unsigned long long>() + i);
    }
    ph.mark(PH_FILTER);
    // (read-backs of the build land in the context's pinned words: [0, 64) bytes counts, [64, 128) sizes, [128, ..) BFS)
    unsigned long long *const h_nk = (unsigned long long *) ctx->pinned_small;
    TLM_HIP(hipMemcpyAsync(h_nk, nk.p, MAX_FILES * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    // 2. rank of each kept record (queued before the host waits for the counts)
    {
        auto in = rocprim::make_transform_iterator(keep.as<uint8_t>(), KeepToU32());
        size_t tb = 0;
        TLM_HIP(rocprim::exclusive_scan(nullptr, tb, in, krank.as<uint32_t>(), 0u, (size_t) R, rocprim::plus<uint32_t>(), s));
        DevBuf tmp;
        if (int rc = dev_alloc(tmp, tb)) return rc;
        TLM_HIP(rocprim::exclusive_scan(tmp.p, tb, in, krank.as<uint32_t>(), 0u, (size_t) R, rocprim::plus<uint32_t>(), s));
    }
    ph.mark(PH_RANK);
    if (int rc = ph.sync()) return rc;
    uint64_t K = 0;
    for (int i = 0; i < n_files; i++) { g->info.n_kept_per_file[i] = h_nk[i]; K += h_nk[i]; }
    g->info.n_records = R;
    g->info.n_kept = K;
    const uint64_t A = 2 * K;
    TLM_CHECK(A < (1ull << 32) - 64, TLM_ELIMIT, "tlm_graph_build: %llu adjacency entries exceed the u32 CSR layout",
              (unsigned long long) A);
    g->n_adj = A;

    // 3. first-seen numbering
    DevBuf first_pos, name_iota, sorted_pos, sorted_name, node_of_name;
    if (int rc = dev_alloc(first_pos, (size_t) n_names * 4)) return rc;
    if (int rc = dev_alloc_keep(&g->kept_rec, (size_t) K)) return rc;
    if (int rc = dev_alloc(name_iota, (size_t) n_names * 4)) return rc;
    if (int rc = dev_alloc(sorted_pos, (size_t) n_names * 4)) return rc;
    if (int rc = dev_alloc(sorted_name, (size_t) n_names * 4)) return rc;
    if (int rc = dev_alloc(node_of_name, (size_t) n_names * 4)) return rc;
    TLM_HIP(hipMemsetAsync(first_pos.p, 0xFF, (size_t) n_names * 4, s));
    for (int i = 0; i < n_files; i++) {
        if (!ft.f[i].n) continue;
        hipLaunchKernelGGL(first_seen_kernel, dim3(grid_for(ft.f[i].n)), dim3(256), 0, s, ft, i, keep.as<uint8_t>(),
                           krank.as<uint32_t>(), first_pos.as<uint32_t>(), g->kept_rec);
    }
    ph.mark(PH_FIRST_SEEN);
    uint32_t N = 0;
    if (n_names) {
        hipLaunchKernelGGL(iota_kernel, dim3((n_names + 255) / 256), dim3(256), 0, s, name_iota.as<uint32_t>(), n_names);
        size_t tb = 0;
        TLM_HIP(rocprim::radix_sort_pairs(nullptr, tb, first_pos.as<uint32_t>(), sorted_pos.as<uint32_t>(),
                                          name_iota.as<uint32_t>(), sorted_name.as<uint32_t>(), (size_t) n_names, 0, 32, s));
        DevBuf tmp;
        if (int rc = dev_alloc(tmp, tb)) return rc;
        TLM_HIP(rocprim::radix_sort_pairs(tmp.p, tb, first_pos.as<uint32_t>(), sorted_pos.as<uint32_t>(),
                                          name_iota.as<uint32_t>(), sorted_name.as<uint32_t>(), (size_t) n_names, 0, 32, s));
        // N = number of names that occur in a kept record
        uint32_t *n_dev = (uint32_t *) (nk.as<unsigned long long>() + MAX_FILES);
        hipLaunchKernelGGL(count_seen_kernel, dim3(1), dim3(1), 0, s, n_names, sorted_pos.as<uint32_t>(), n_dev);
        uint32_t *const h_n = (uint32_t *) ((char *) ctx->pinned_small + 64);
        TLM_HIP(hipMemcpyAsync(h_n, n_dev, 4, hipMemcpyDeviceToHost, s));
        if (int rc = ph.sync()) return rc;
        N = *h_n;
    }
    g->n_nodes = N;
    if (int rc = dev_alloc_keep(&g->node_len, N)) return rc;
    if (int rc = dev_alloc_keep(&g->node_anchor, N)) return rc;
    if (int rc = dev_alloc_keep(&g->node_name_id, N)) return rc;
    if (N) {
        hipLaunchKernelGGL(make_nodes_kernel, dim3((N + 255) / 256), dim3(256), 0, s, ft, N, sorted_pos.as<uint32_t>(),
                           sorted_name.as<uint32_t>(), g->kept_rec, name_is_ctg_dev,
                           node_of_name.as<uint32_t>(), g->node_len, g->node_anchor, g->node_name_id);
    }
    ph.mark(PH_NODES);

    // 4. adjacency entries, stable sort by owner
    if (int rc = dev_alloc_keep(&g->row_ptr, (size_t) N + 1)) return rc;
    if (int rc = dev_alloc_keep(&g->ent_es, A)) return rc;
    if (int rc = dev_alloc_keep(&g->ent_os, A)) return rc;
    if (int rc = dev_alloc_keep(&g->q_start, A)) return rc;
    if (int rc = dev_alloc_keep(&g->q_end, A)) return rc;
    if (int rc = dev_alloc_keep(&g->t_start, A)) return rc;
    if (int rc = dev_alloc_keep(&g->t_end, A)) return rc;
    if (int rc = dev_alloc_keep(&g->si, A)) return rc;
    if (int rc = dev_alloc_keep(&g->strand, A)) return rc;
    TLM_HIP(hipMemsetAsync(g->row_ptr, 0, ((size_t) N + 1) * 4, s));
    DevBuf proper, n_internal;      // one byte per adjacency entry: proper overlap? (graph_finish numbers the nodes along those)
    if (A) {
        DevBuf owner, eidx, sorted_owner, sorted_e;
        if (int rc = dev_alloc(owner, A * 4)) return rc;
        if (int rc = dev_alloc(eidx, A * 4)) return rc;
        if (int rc = dev_alloc(sorted_owner, A * 4)) return rc;
        if (int rc = dev_alloc(sorted_e, A * 4)) return rc;
        hipLaunchKernelGGL(make_entries_kernel, dim3(grid_for(K)), dim3(256), 0, s, ft, K, g->kept_rec,
                           node_of_name.as<uint32_t>(), owner.as<uint32_t>(), eidx.as<uint32_t>());
        unsigned end_bit = 1;
        while ((1ull << end_bit) < (uint64_t) N && end_bit < 32) end_bit++;
        size_t tb = 0;
        TLM_HIP(rocprim::radix_sort_pairs(nullptr, tb, owner.as<uint32_t>(), sorted_owner.as<uint32_t>(), eidx.as<uint32_t>(),
                                          sorted_e.as<uint32_t>(), (size_t) A, 0, end_bit, s));
        DevBuf tmp;
        if (int rc = dev_alloc(tmp, tb)) return rc;
        TLM_HIP(rocprim::radix_sort_pairs(tmp.p, tb, owner.as<uint32_t>(), sorted_owner.as<uint32_t>(), eidx.as<uint32_t>(),
                                          sorted_e.as<uint32_t>(), (size_t) A, 0, end_bit, s));
        hipLaunchKernelGGL(row_ptr_kernel, dim3(grid_for(A)), dim3(256), 0, s, A, N, sorted_owner.as<uint32_t>(), g->row_ptr);
        ph.mark(PH_OWNER_SORT);
        // 5. gather, through one compact record per kept overlap
        DevBuf kf;
        if (int rc = dev_alloc(kf, (size_t) K * sizeof(KeptFields))) return rc;
        hipLaunchKernelGGL(kept_fields_kernel, dim3(grid_for(K)), dim3(256), 0, s, ft, K, g->kept_rec,
                           node_of_name.as<uint32_t>(), g->node_anchor, r_si.as<float>(), r_os.as<float>(),
                           r_es.as<float>(), kf.as<KeptFields>());
        if (int rc = dev_alloc(proper, (size_t) A + 1)) return rc;
        if (int rc = dev_alloc(n_internal, 4)) return rc;
        TLM_HIP(hipMemsetAsync(n_internal.p, 0, 4, s));
        hipLaunchKernelGGL(gather_kernel, dim3(grid_for(A)), dim3(256), 0, s, A, sorted_e.as<uint32_t>(),
                           kf.as<KeptFields>(), g->ent_es, g->ent_os, g->q_start, g->q_end, g->t_start, g->t_end, g->si,
                           g->strand, proper.as<uint8_t>(), n_internal.as<uint32_t>());
        g->proper_tmp = proper.as<uint8_t>();       // for the walk-order numbering of graph_finish only
        g->n_internal_tmp = n_internal.as<uint32_t>();
        ph.mark(PH_GATHER);
    }
    // 6.
    const int rc_finish = graph_finish(ctx, g, ph);
    g->proper_tmp = nullptr;
    g->n_internal_tmp = nullptr;
    if (rc_finish) return rc_finish;
    TLM_HIP(hipEventRecord(ctx->ev1, s));
    TLM_HIP(hipEventSynchronize(ctx->ev1));
    float ms = 0;
    TLM_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    g->info.build_ms = ms;
    ph.collect(g->info);
    g->info.filter_ms = g->info.phase_ms[PH_FILTER];
    TLM_HIP(hipGetLastError());
    guard.g = nullptr;
    *out = g;
    return TLM_OK;
}

extern "C" int tlm_graph_edges(const tlm_graph *g, uint64_t n, const uint32_t *nodes_host, const uint32_t *slots_host,
                               tlm_edge *out_host) {
    TLM_CHECK(g && (n == 0 || (nodes_host && slots_host && out_host)), TLM_EINVAL, "tlm_graph_edges: null argument");
    if (!n) return TLM_OK;
    TLM_HIP(hipSetDevice(g->device));
    hipStream_t s = g->ctx->stream;
    t_build_stream = s;
    DevBuf dn, dsl, out, bad;
    if (int rc = dev_alloc(dn, n * 4)) return rc;
    if (int rc = dev_alloc(dsl, n * 4)) return rc;
    if (int rc = dev_alloc(out, n * sizeof(tlm_edge))) return rc;
    if (int rc = dev_alloc(bad, 4)) return rc;
    TLM_HIP(hipMemcpyAsync(dn.p, nodes_host, n * 4, hipMemcpyHostToDevice, s));
    TLM_HIP(hipMemcpyAsync(dsl.p, slots_host, n * 4, hipMemcpyHostToDevice, s));
    TLM_HIP(hipMemsetAsync(bad.p, 0, 4, s));
    hipLaunchKernelGGL(edges_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, n, g->n_nodes, dn.as<uint32_t>(),
                       dsl.as<uint32_t>(), g->row_ptr, g->ent_es, g->ent_os, g->q_start, g->q_end, g->t_start, g->t_end, g->si,
                       g->strand, out.as<tlm_edge>(), bad.as<uint32_t>());
    uint32_t h_bad = 0;
    TLM_HIP(hipMemcpyAsync(out_host, out.p, n * sizeof(tlm_edge), hipMemcpyDeviceToHost, s));
    TLM_HIP(hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, s));
    TLM_HIP(hipStreamSynchronize(s));
    TLM_HIP(hipGetLastError());
    TLM_CHECK(h_bad == 0, TLM_EINVAL, "tlm_graph_edges: %u (node, slot) pairs lie outside the graph", h_bad);
    return TLM_OK;
}

extern "C" int tlm_graph_get_stats(const tlm_graph *g, tlm_graph_stats *out) {
    TLM_CHECK(g && out, TLM_EINVAL, "tlm_graph_get_stats: null argument");
    TLM_HIP(hipSetDevice(g->device));
    hipStream_t s = g->ctx->stream;
    t_build_stream = s;
    DevBuf acc;
    if (int rc = dev_alloc(acc, 16 * 8)) return rc;
    unsigned long long h[16];
    for (int i = 0; i < 16; i++) h[i] = 0;
    h[0] = h[2] = h[4] = ~0ull;
    h[6] = h[8] = h[10] = 0xFFFFFFFFull;
    TLM_HIP(hipMemcpyAsync(acc.p, h, sizeof(h), hipMemcpyHostToDevice, s));
    if (g->n_nodes)
        hipLaunchKernelGGL(stats_nodes_kernel, dim3(grid_for(g->n_nodes)), dim3(256), 0, s, g->n_nodes, g->row_ptr, g->node_len,
                           g->node_anchor, acc.as<unsigned long long>());
    if (g->n_adj)
        hipLaunchKernelGGL(stats_edges_kernel, dim3(grid_for(g->n_adj)), dim3(256), 0, s, g->n_adj, g->ent_es, g->ent_os,
                           g->q_start, g->q_end, g->t_start, g->t_end, g->si, acc.as<unsigned long long>());
    TLM_HIP(hipMemcpyAsync(h, acc.p, sizeof(h), hipMemcpyDeviceToHost, s));
    TLM_HIP(hipStreamSynchronize(s));
    TLM_HIP(hipGetLastError());
    auto unord = [](unsigned long long o) {
        uint32_t b = (uint32_t) o;
        b = (b & 0x80000000u) ? (b & 0x7FFFFFFFu) : ~b;
        float f;
        std::memcpy(&f, &b, 4);
        return f;
    };
    out->anchors = h[12];
    out->reads = g->n_nodes - h[12];
    out->min_len = g->n_nodes ? h[0] : ~0ull;
    out->max_len = h[1];
    out->min_con = g->n_nodes ? h[2] : ~0ull;
    out->max_con = h[3];
    out->edges = g->info.n_kept;
    out->min_ol = h[4];
    out->max_ol = h[5];
    // the reference starts its minima at FLT_MAX and its maxima at 0 (:40-43): a maximum never drops below 0
    const bool any[3] = {h[6] != 0xFFFFFFFFull, h[8] != 0xFFFFFFFFull, h[10] != 0xFFFFFFFFull};
    float mn[3], mx[3];
    for (int k = 0; k < 3; k++) {
        mn[k] = any[k] ? std::min(unord(h[6 + 2 * k]), 3.402823466e+38f) : 3.402823466e+38f;
        mx[k] = any[k] ? std::max(unord(h[7 + 2 * k]), 0.0f) : 0.0f;
    }
    out->min_os = mn[0]; out->max_os = mx[0];
    out->min_es = mn[1]; out->max_es = mx[1];
    out->min_si = mn[2]; out->max_si = mx[2];
    return TLM_OK;
}

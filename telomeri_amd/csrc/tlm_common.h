// telomeri_amd/csrc/tlm_common.h -- internal declarations shared by the HIP translation units.
#ifndef TLM_COMMON_H
#define TLM_COMMON_H

#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/telomeri_hip.h"

void tlm_set_error(const char *fmt, ...);

#define TLM_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            tlm_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return TLM_EHIP;                                                                   \
        }                                                                                      \
    } while (0)

#define TLM_CHECK(cond, code, ...)                                                             \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            tlm_set_error(__VA_ARGS__);                                                        \
            return (code);                                                                     \
        }                                                                                      \
    } while (0)

struct tlm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int n_cus = 0;
    int waves_per_cu = 0;          // 0 = default
    uint32_t path_capacity = 0;    // 0 = default
    uint64_t arena_limit = 0;      // bytes; 0 = default (tlm_ctx_set_arena_limit)
    bool walk_accounting = false;  // tlm_ctx_set_walk_accounting: launch the builds of the walk kernels that count layout bytes
    std::vector<hipEvent_t> phase_ev;   // graph-build phase marks, created on first use
    void *comm = nullptr;          // ncclComm_t (exchange.hip); rank / world of the one-process-per-GPU job
    int rank = 0, world = 1;
    void *xchg_scratch = nullptr;  // device, TLM_XCHG_SCRATCH_BYTES: status words and block sizes of the exchange calls; owned
                                   // since tlm_dist_init so that telling the peers "I failed" never needs an allocation
    bool comm_broken = false;      // the communicator was aborted after a failure inside a collective sequence
    // pinned host memory for the few words the graph build reads back between its phases (counts, sizes): a device-to-host
    // copy into pageable memory goes through the runtime's staging path and may block the queueing thread
    void *pinned_small = nullptr;
    // pinned host memory of the exchange (the summaries of every rank arrive here; grows, goes with the context)
    void *xchg_pinned = nullptr;
    size_t xchg_pinned_bytes = 0;
    void *upload_window = nullptr; // device memory shared with an upload helper process (tlm_upload_window_create)
    void *peer_window = nullptr;   // ... and, in the helper, the other process's window (tlm_upload_window_open)
    // host-side tidying that may wait (unpinning and unmapping 15 GB of uploaded text takes 0.65 s and blocks every
    // allocation of the process meanwhile): started when the device has just been given a long kernel, or at the latest
    // when the context goes
    std::mutex deferred_mu;
    std::vector<std::function<void()>> deferred;
    std::vector<std::thread> deferred_running;
};
#define TLM_XCHG_SCRATCH_BYTES 16384
#define TLM_PINNED_SMALL_BYTES 4096
void tlm_ctx_defer(tlm_ctx *ctx, std::function<void()> fn);
void tlm_ctx_start_deferred(tlm_ctx *ctx);      // run what has been deferred so far, in background threads
void tlm_ctx_join_deferred(tlm_ctx *ctx);       // start the rest and wait for all of it

// Device-resident graph.  HBM layout:
//   row_ptr[N+1]            u32   CSR offsets (A < 2^32)
//   ent_es[A], ent_os[A]    u32x2 {neighbour | anchor(neighbour) << 31, fp32 bits of ES resp. OS}
//                                 -- the 8 B a walk reads per edge-extension eval
//   q_start[A], t_start[A]  u32   length increments (Path::updateLength), read once per step
//   q_end, t_end, si, strand      cold columns for the Scaffolder / downloads
//   node_len[N] u32, node_anchor[N] u8, node_flags[N] u8 (bit m: metric m's sums are order-safe)
struct tlm_graph {
    tlm_ctx *ctx = nullptr;
    int device = 0;                    // copy of ctx->device: destroy must not touch a dead context
    uint32_t n_nodes = 0;
    uint64_t n_adj = 0;
    uint32_t *row_ptr = nullptr;
    uint2 *ent_es = nullptr, *ent_os = nullptr;
    uint32_t *q_start = nullptr, *t_start = nullptr, *q_end = nullptr, *t_end = nullptr;
    float *si = nullptr;
    uint8_t *strand = nullptr;
    const uint8_t *proper_tmp = nullptr;   // during tlm_graph_build only: 1 = proper overlap (csr_build.hip, walk-order numbering)
    const uint32_t *n_internal_tmp = nullptr;      // ... and a word that is non-zero when any entry is an internal match
    uint32_t *node_len = nullptr;
    uint8_t *node_anchor = nullptr;
    uint8_t *node_flags = nullptr;
    // walk layout: every row padded to a multiple of 8 entries (24 for degrees 257..384), so that a lane
    // reads its 4 / 6 / 8 consecutive entries with aligned 16-byte loads; padding names the row's own node; one
    // 8-byte record per node
    uint64_t n_adj_pad = 0;
    uint2 *w_row = nullptr;            // [N] {padded row start, degree | order-safety flags << 24}
    uint2 *w_ent_es = nullptr, *w_ent_os = nullptr;   // [n_adj_pad]
    uint2 *w_qt = nullptr;             // [n_adj_pad] {q_start, t_start}
    // compact copy of the same rows, same index space: 4-byte metric + 2-byte neighbour id relative to the
    // row's own node = 6 bytes per eval instead of 8; read by the walk kernel for every row whose flag bit 5 is set
    uint32_t *w_m_es = nullptr, *w_m_os = nullptr;   // [n_adj_pad] fp32 bits
    uint16_t *w_d16 = nullptr;         // [n_adj_pad] neighbour - owner + 32768 (walk-order ids)
    uint32_t *w_old_of_new = nullptr;  // [N] reference node index of each walk-order node
    uint32_t *w_anchors = nullptr;     // [n_anchors] walk-order index of each anchor
    int32_t *node_name_id = nullptr;
    uint32_t *kept_rec = nullptr;      // [n_kept] global record index of each kept record (tlm_graph_build only)
    uint32_t *anchors = nullptr;       // [n_anchors] node index, ascending
    uint64_t *det_off = nullptr;       // [n_anchors+1] prefix of DET units per anchor
    std::vector<uint32_t> h_anchors;
    std::vector<uint64_t> h_det_off;
    tlm_graph_info info{};
};

// internal per-unit status codes beyond the public TLM_UNIT_* ones
enum : uint8_t { ST_ABANDONED = 0, ST_ACCEPTED = 1, ST_NEGATIVE = 2, ST_SCRATCH_OVF = 3, ST_ARENA_OVF = 4, ST_NOT_RUN = 255 };

struct tlm_pathset {
    tlm_ctx *ctx = nullptr;
    int device = 0;
    uint64_t n_units = 0, n_paths = 0, n_nodes_total = 0;
    // packed result (device)
    uint64_t *unit = nullptr;
    int64_t *length = nullptr;
    uint64_t *off = nullptr;
    uint32_t *nodes = nullptr, *slots = nullptr;
    std::vector<tlm_unit_stat> stats;   // host copy
};

int tlm_pathset_sort_by_unit(tlm_ctx *ctx, tlm_pathset *in, tlm_pathset **out);     // pathset_ops.hip (consumes `in`)

// Every device buffer of the library comes from its own per-device cache (device_pool.hip): blocks are hipMalloc'd once
// and re-used by later requests of (nearly) the same size; hipMalloc / hipFree of hundreds of megabytes per builder call
// cost milliseconds and synchronise the device.  Nothing process-wide (the device's default memory pool) is touched.
// tlm_pool_free may run after the context is gone (path sets outlive it in the Python tests).
int tlm_pool_alloc(void **p, size_t bytes, hipStream_t s);
// device_is_synced: the caller has synchronised the device since the block's last use (no ordering event is needed)
void tlm_pool_free(void *p, bool device_is_synced = false);
void tlm_pool_forget_stream(int device, hipStream_t s);     // before a stream is destroyed
void tlm_pool_trim(int device);                              // return every cached block to the driver
void tlm_pool_info(int device, tlm_memory_info *out);
double tlm_pool_thread_malloc_ms();                          // host time this thread has spent in hipMalloc so far

// path-set helpers on device buffers (pathset_ops.hip), for the exchange
int tlm_pathset_summary_dev(const tlm_pathset *ps, hipStream_t s, uint32_t *front, uint32_t *back, uint32_t *n_nodes,
                            uint64_t *hash, uint8_t *dup);
int tlm_pathset_gather_dev(const tlm_pathset *ps, hipStream_t s, uint64_t n, const uint64_t *index_dev, const uint64_t *dst_off_dev,
                           uint32_t *nodes_dev, uint32_t *slots_dev);

// the sharded load's swap of text pieces (exchange.hip; see tlm_paf_text_share in the header)
int tlm_text_share(tlm_ctx *ctx, char *text, uint64_t bytes, uint64_t piece, int local_rc, tlm_exchange_info *info);

// graph build (csr_build.hip)
int tlm_graph_finish(tlm_ctx *ctx, tlm_graph *g);   // node_flags, anchors, det_off, info

// kernels launched from the API layer
void tlm_launch_filter_score(hipStream_t s, const tlm_filter_params &p, const tlm_records &r, uint8_t *keep, float *si,
                             float *os, float *es, unsigned long long *n_kept_dev);

#endif

/* include/telomeri_hip.h -- C ABI of libtelomeri_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary for the overlap-graph path-construction hot path of lirfu/Telomeri.
 * The reference has no FFI layer; its boundary is the C++ class surface main() calls
 * (SURVEY.md §8b).  Each entry point below replaces the body of one of those members; the host
 * classes in telomeri_amd/host/ keep the reference's signatures and call these.
 *
 *   reference interface (file:line)                                   entry point(s)
 *   ------------------------------------------------------------------------------------------
 *   OverlapGraph::filter            src/OverlapGraph.cpp:156-202      tlm_filter_score
 *   OverlapGraph::buildFrom scores  src/OverlapGraph.cpp:221-231      tlm_filter_score
 *   OverlapGraph::load / buildFrom  src/OverlapGraph.cpp:97-154,204-258
 *       (node numbering, anchor flags, per-node edge lists)           tlm_graph_build, tlm_graph_*
 *   PathManager::buildMonteCarlo    src/PathManager.cpp:12-177        tlm_walk_mc
 *   PathManager::buildDeterministic src/PathManager.cpp:179-386       tlm_walk_det
 *   Path / Path::updateLength       include/Path.hpp:7-21, src/Path.cpp:6-16   tlm_pathset_*
 *   Utils::Metrics / getMetric      include/Utils.hpp:43-60           `metric` argument
 *   PathManager::Parameters         include/PathManager.hpp:27-40     tlm_walk_params
 *   OverlapGraph::FilterParameters  include/OverlapGraph.hpp:67-78    tlm_filter_params
 *
 * Conventions: plain C types, no exceptions cross the ABI.  Every function returning int returns
 * 0 (TLM_OK) on success or a negative TLM_E* code; tlm_last_error() gives the message of the last
 * failure on the calling thread.  One host thread per context.  "dev" pointers are HIP device
 * pointers on the context's device, "host" pointers ordinary memory.  There is NO CPU fallback:
 * every entry point that computes runs on the GPU or fails.
 */
#ifndef TELOMERI_HIP_H
#define TELOMERI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 1: filter_score, graph, walks, path sets.  2: path summaries / fetch, Edge records, stats, the exchange (tlm_dist_*,
 * tlm_pathset_allgather, tlm_pathset_gather_summaries), memory accounting, PAF text on the device.  3 (additions only):
 * TLM_EPEER and the fail-together contract of the exchange calls, TLM_RCCL_LIB, tlm_ctx_set_walk_accounting with
 * tlm_counters.layout_bytes / step_class, tlm_ctx_release_workspaces, tlm_graph_info.phase_host_ms / sync_wait_ms.
 * 4 (additions only): walks over a LIST of units (tlm_walk_list, tlm_walk_builders_list) and the sliced partition
 * (tlm_dist_unit_share); the exchange calls return paths_ order (ascending builder | unit) whatever the partition.
 * 5 (additions only): the sharded load of a multi-rank job (tlm_dist_text_pieces / _run, tlm_paf_text_share,
 * tlm_upload_file_window_range, tlm_paf_share_info; tlm_paf_add_file shares by itself in a context with a communicator). */
#define TLM_ABI_VERSION 5

enum {
    TLM_OK = 0,
    TLM_EINVAL = -1,     /* bad argument */
    TLM_EHIP = -2,       /* a HIP runtime call failed (no device, out of memory, launch failure) */
    TLM_ELIMIT = -3,     /* a size limit of the device layout was exceeded */
    TLM_EINTERNAL = -4,
    TLM_EPEER = -5       /* multi-GPU: ANOTHER rank of the job failed (or the communicator has been given up after a failure
                            of this rank); every rank of an exchange call fails together instead of waiting for ever */
};

/* OverlapGraph::LengthCalculationMode (include/OverlapGraph.hpp:56-65), same numbering */
enum { TLM_MODE_MIN = 0, TLM_MODE_MAX = 1, TLM_MODE_AVG = 2, TLM_MODE_SUM = 3 };
/* Utils::Metrics (include/Utils.hpp:43-45): EXTENSION_SCORE, OVERLAP_SCORE, *_SQRT */
enum { TLM_METRIC_ES = 0, TLM_METRIC_OS = 1, TLM_METRIC_ES_SQRT = 2, TLM_METRIC_OS_SQRT = 3 };
/* per-unit walk outcome (tlm_unit_stat.status) */
enum { TLM_UNIT_ABANDONED = 0, TLM_UNIT_ACCEPTED = 1, TLM_UNIT_NEGATIVE = 2 };

typedef struct {
    int32_t mode;
    uint64_t min_overlap_length;
    float min_overlap_percentage;
    uint64_t max_overhang_length;
    float max_overhang_percentage;
} tlm_filter_params;

typedef struct {
    int32_t rebuild_attempts;      /* Monte-Carlo trials per anchor */
    int32_t backtrack_attempts;
    int64_t len_threshold;
} tlm_walk_params;

typedef struct {
    uint64_t units;                /* walk units executed */
    uint64_t steps;                /* selection scans (MC) / ranked steps (DET) */
    uint64_t evals;                /* adjacency entries examined by those scans: the "edge-extension evals" */
    uint64_t draws;                /* uniform doubles drawn */
    uint64_t accepted;             /* paths in the result */
    double kernel_ms;              /* duration of the walk kernel(s), HIP events on the context's stream */
    uint32_t kernel_launches;      /* walk-kernel launches behind kernel_ms (re-runs after an overflow count) */
    uint32_t waves;                /* persistent wavefronts of the first launch */
    double kernel_busy_ms;         /* first workgroup in -> last workgroup out, device clock: equals kernel_ms for a launch
                                      that starts at once; tlm_walk_builders queues its deterministic launch behind the
                                      Monte-Carlo launch, where kernel_ms includes the wait for free compute units */
    uint64_t layout_bytes;         /* (ABI 3) 0 unless tlm_ctx_set_walk_accounting is on: the bytes the steps' loads asked
                                      the walk layout for -- list entries as read (6-byte compact or 8-byte rows, padded
                                      to the lane width), row record, {q_start, t_start}; booked like kernel_ms */
    /* (ABI 3, accounting launches) steps by the form of their selection: 0-3 compact row read with 4 / 6 / 8 / 12 entries
     * per lane (lists up to 256 / 384 / 512 / 768), 4-7 the same on an 8-byte row (7: deterministic kernel only), 8 chunked
     * long list, 9 serial (order-unsafe row) */
    uint64_t step_class[10];
} tlm_counters;
#define TLM_STEP_CLASSES 10

typedef struct {
    uint8_t status;
    uint32_t n_nodes;              /* nodes on the path when the walk ended */
    uint32_t steps;
    uint64_t evals;
    uint32_t draws;
    int64_t length;                /* last Path::length() */
} tlm_unit_stat;

/* One PAF file as device-resident integer columns (SoA), `n` records each.
 * Names are ids into one name table shared by all files of a graph; the host tokeniser assigns
 * them (any injective numbering works: node indices are re-derived on the device in the
 * reference's first-seen order; "same name" (OverlapGraph.cpp:157) is q_id == t_id).  `anchors` is OverlapGraph::load's second argument. */
typedef struct {
    uint64_t n;
    const int32_t *q_id, *t_id;                         /* dev */
    const int32_t *q_len, *q_start, *q_end;             /* dev */
    const int32_t *t_len, *t_start, *t_end;             /* dev */
    const int32_t *matches;                             /* dev */
    const uint8_t *strand;                              /* dev: 1 = relative strand '-' */
    int32_t anchors;
} tlm_records;

typedef struct tlm_ctx tlm_ctx;
typedef struct tlm_graph tlm_graph;
typedef struct tlm_pathset tlm_pathset;

int tlm_abi_version(void);
const char *tlm_last_error(void);

/* ---- context ------------------------------------------------------------------------------- */
int tlm_ctx_create(int device, tlm_ctx **out);
void tlm_ctx_destroy(tlm_ctx *ctx);
/* The HIP stream all work of the context is queued on (a hipStream_t). */
void *tlm_ctx_stream(tlm_ctx *ctx);
int tlm_ctx_synchronize(tlm_ctx *ctx);
/* Tuning: persistent wavefronts per CU for the walk kernels (0 = default) and the capacity, in
 * nodes, of a wave's path buffer before a walk is re-run with a full-size buffer (0 = default). */
int tlm_ctx_set_walk_tuning(tlm_ctx *ctx, int waves_per_cu, uint32_t path_capacity);
/* Tuning: upper bound, in bytes, of the arena the FIRST launch of a walk call collects accepted paths in (0 = default:
 * sized from the number of units).  A launch that outgrows its arena is re-run for the units that did not fit with a
 * larger one (results unchanged); tests use a small limit to exercise that. */
int tlm_ctx_set_arena_limit(tlm_ctx *ctx, uint64_t bytes);
/* Measurement: with `on` != 0 the walk calls launch a second build of the same kernels that also adds up the bytes its
 * steps ask the walk layout for (tlm_counters.layout_bytes).  Same walks, same results; a little slower, so never on for a
 * timed launch: bench.py runs one accounting launch beside its timed steps to price the roofline against the bytes the
 * layout really has to move. */
int tlm_ctx_set_walk_accounting(tlm_ctx *ctx, int on);

/* Device memory held by the library on the context's device.  Every buffer (graph, temporaries, walk workspaces, path
 * sets) comes from a cache owned by the library: blocks are hipMalloc'd once and re-used by later calls; the device's
 * default memory pool and its attributes are never touched. */
typedef struct {
    uint64_t live_bytes;           /* in use by graphs, path sets, workspaces */
    uint64_t cached_bytes;         /* released by the library, kept for re-use */
    uint64_t peak_live_bytes;
    uint64_t driver_allocs;        /* hipMalloc calls so far */
    uint64_t reuses;               /* requests served from the cache */
    double driver_alloc_ms;        /* host time spent in those hipMalloc calls */
} tlm_memory_info;
int tlm_ctx_memory_info(tlm_ctx *ctx, tlm_memory_info *out);
/* Give the walk workspaces of the context and every cached block of its device back to the driver (graphs and path
 * sets stay).  Waits for the device. */
int tlm_ctx_trim(tlm_ctx *ctx);
/* The walk workspaces of the context (the arena accepted paths are collected in, the waves' path buffers and visited maps:
 * 16 GB after the builders of the A. thaliana-shaped job) go back to the library's CACHE -- not to the driver: what the
 * job allocates next (the concatenated Path set, its filtered copy) is carved out of them.  For a host that has made its
 * last walk call; a later walk call simply obtains workspaces again.  Waits for the context's streams. */
int tlm_ctx_release_workspaces(tlm_ctx *ctx);

/* ---- a2 + a3: filter decision and SI/OS/ES per record -------------------------------------- */
/* keep[n] (1 = passes OverlapGraph::filter), si/os/es[n] written for EVERY record; any output
 * pointer may be NULL.  n_kept (host, may be NULL) receives the number of kept records. */
int tlm_filter_score(tlm_ctx *ctx, const tlm_filter_params *p, const tlm_records *rec, uint8_t *keep_dev,
                     float *si_dev, float *os_dev, float *es_dev, uint64_t *n_kept);

/* ---- a4: the graph -------------------------------------------------------------------------- */
/* OverlapGraph::load for `n_files` files in order (Main.cpp:223 loads rr then cr into one graph):
 * filter, score, first-seen node numbering among kept records (query before target), anchor flag
 * and length fixed at node creation, two adjacency entries per kept record, per-node order =
 * record order.  `name_is_ctg[n_names]` (dev, uint8): name starts with "ctg", case-insensitive. */
int tlm_graph_build(tlm_ctx *ctx, const tlm_filter_params *p, const tlm_records *files, int n_files,
                    uint32_t n_names, const uint8_t *name_is_ctg_dev, tlm_graph **out);
/* The same with every column of `files` (and name_is_ctg) in HOST memory: the library uploads them.  This is
 * what a host program that has just tokenised its PAF files calls (telomeri_amd/host/). */
int tlm_graph_build_host(tlm_ctx *ctx, const tlm_filter_params *p, const tlm_records *files_host, int n_files,
                         uint32_t n_names, const uint8_t *name_is_ctg_host, tlm_graph **out);
/* Records kept by the filter, in order: rec[k] = index of the k-th kept record, counted over all files of the
 * build in load order (HOST buffer of tlm_graph_info.n_kept entries).  This is edges_ order (OverlapGraph.cpp:241).
 * Only for graphs made by tlm_graph_build*. */
int tlm_graph_kept_records(const tlm_graph *g, uint32_t *rec_host);
/* Adopt a finished CSR given in HOST memory (tests; hosts that build adjacency themselves).
 * Entry k of node i: nbr[row_ptr[i]+k]; q_end/t_end/si/strand may be NULL. */
int tlm_graph_from_csr(tlm_ctx *ctx, uint32_t n_nodes, uint64_t n_adj, const uint64_t *row_ptr,
                       const uint32_t *nbr, const uint32_t *q_start, const uint32_t *q_end,
                       const uint32_t *t_start, const uint32_t *t_end, const float *os, const float *si,
                       const float *es, const uint8_t *strand, const uint32_t *node_len,
                       const uint8_t *node_anchor, tlm_graph **out);
void tlm_graph_destroy(tlm_graph *g);

#define TLM_BUILD_PHASES 12
typedef struct {
    uint32_t n_nodes;
    uint32_t n_anchors;
    uint64_t n_adj;                /* adjacency entries = 2 * kept records */
    uint64_t n_records;            /* records seen by tlm_graph_build, all files */
    uint64_t n_kept;               /* edges_.size() */
    uint64_t n_kept_per_file[4];
    uint32_t max_degree;
    uint32_t unsafe_nodes[4];      /* per metric: nodes whose roulette sums are not order-independent */
    uint32_t bfs_levels;           /* levels of the multi-source BFS behind the walk layout's node order (0 = none) */
    double build_ms;               /* device time of tlm_graph_build */
    double filter_ms;              /* of which: the filter_score kernel */
    /* the build by phase (HIP events on the context's stream, so a host stall between two kernels is booked on the
     * phase it happens in); names from tlm_build_phase_name().  Phases a graph did not go through stay 0. */
    double phase_ms[TLM_BUILD_PHASES];
    double build_wall_ms;          /* host wall-clock of the build call */
    double build_alloc_ms;         /* of which: host time in hipMalloc (0 once the library's cache is warm) */
    uint32_t host_syncs;           /* host waits for the device during the build */
    /* (ABI 3) the HOST's time between the same marks, its waits for the device excluded: what the host spent queueing a
     * phase.  A phase with a few ms of kernels and tens of ms here was held up on the host (an allocation, a blocking
     * copy, the process being descheduled), not on the device.  sync_wait_ms = the excluded waits, all phases. */
    double phase_host_ms[TLM_BUILD_PHASES];
    double sync_wait_ms;
} tlm_graph_info;
int tlm_graph_get_info(const tlm_graph *g, tlm_graph_info *out);
const char *tlm_build_phase_name(int phase);

/* Copy the graph to HOST arrays (each pointer may be NULL to skip it).  Sizes from tlm_graph_info.
 * node_name_id[i] is the name id that created node i (-1 for tlm_graph_from_csr graphs). */
int tlm_graph_download(const tlm_graph *g, uint64_t *row_ptr, uint32_t *nbr, uint32_t *q_start, uint32_t *q_end,
                       uint32_t *t_start, uint32_t *t_end, float *os, float *si, float *es, uint8_t *strand,
                       uint32_t *node_len, uint8_t *node_anchor, int32_t *node_name_id);

/* OverlapGraph::Edge (include/OverlapGraph.hpp:27-43) of given adjacency entries: entry slots[k] of nodes[k]'s list.
 * For hosts that do not materialise nodes_[..].edges (233 M entries at the A. thaliana size) and need the Edge records of
 * a few paths only (Scaffolder, Scaffolder.cpp:63-99).  HOST buffers of n entries; the CSR layout of tlm_graph_download. */
typedef struct {
    uint32_t q_index, t_index;     /* in node X's list: t_index == X, q_index = neighbour */
    uint32_t q_start, q_end, t_start, t_end;
    float overlap_score, sequence_identity, extension_score;
    uint8_t relative_strand;
} tlm_edge;
int tlm_graph_edges(const tlm_graph *g, uint64_t n, const uint32_t *nodes_host, const uint32_t *slots_host,
                    tlm_edge *out_host);
/* What OverlapGraph::stats (OverlapGraph.cpp:24-95) reports, reduced on the device: the per-record extrema equal the
 * per-entry ones (both entries of a record carry its scores; the overlap length is symmetric under the swap). */
typedef struct {
    uint64_t anchors, reads;
    uint64_t min_len, max_len;     /* node lengths */
    uint64_t min_con, max_con;     /* list lengths */
    uint64_t edges;                /* edges_.size() = kept records */
    uint64_t min_ol, max_ol;       /* max(q_end - q_start, t_end - t_start) as the reference computes it (:65) */
    float min_os, max_os, min_es, max_es, min_si, max_si;
} tlm_graph_stats;
int tlm_graph_get_stats(const tlm_graph *g, tlm_graph_stats *out);

/* ---- a7 / a8: walks ------------------------------------------------------------------------- */
/* Walk units.  MC: unit = anchor_rank * rebuild_attempts + r, anchors in node
 * order.  DET: units enumerate, anchor by anchor, the first-edge positions the reference's loop
 * visits (it stops after the first edge that leads straight to an anchor, PathManager.cpp:224-229). */
uint64_t tlm_mc_units(const tlm_graph *g, const tlm_walk_params *p);
uint64_t tlm_det_units(const tlm_graph *g);
/* The unit tables, HOST buffers: anchor_nodes[n_anchors] = node index of each anchor in node order,
 * det_off[n_anchors+1] = prefix of DET units per anchor (anchor k owns DET units
 * [det_off[k], det_off[k+1]) and MC units [k*rebuild_attempts, (k+1)*rebuild_attempts)).  Either may be NULL. */
int tlm_graph_anchor_table(const tlm_graph *g, uint32_t *anchor_nodes, uint64_t *det_off);
/* Seed of the std::mt19937 owned by MC walk (anchor node, trial r) under `metric`. */
uint32_t tlm_walk_seed(int metric, uint32_t anchor_node, uint32_t r);

/* Run units [unit_begin, unit_end) and return the accepted paths, ordered by unit. */
int tlm_walk_mc(tlm_ctx *ctx, const tlm_graph *g, int metric, const tlm_walk_params *p, uint64_t unit_begin,
                uint64_t unit_end, tlm_pathset **out, tlm_counters *ctr);
int tlm_walk_det(tlm_ctx *ctx, const tlm_graph *g, int metric, uint64_t unit_begin, uint64_t unit_end,
                 tlm_pathset **out, tlm_counters *ctr);

/* The same for TWO metrics of the same units in ONE persistent launch (reference Main.cpp:237-243 runs every builder
 * under EXTENSION_SCORE and under OVERLAP_SCORE): the two walk sets share the device's work queue, which shortens the
 * tail of the launch.  out2[0] / ctr2[0] belong to metric_a, out2[1] / ctr2[1] to metric_b; results are identical to two
 * single-metric calls.  The launch's kernel_ms / kernel_launches are booked on ctr2[0].  Both metrics must be plain or
 * both *_SQRT. */
int tlm_walk_mc2(tlm_ctx *ctx, const tlm_graph *g, int metric_a, int metric_b, const tlm_walk_params *p,
                 uint64_t unit_begin, uint64_t unit_end, tlm_pathset **out2, tlm_counters *ctr2);
int tlm_walk_det2(tlm_ctx *ctx, const tlm_graph *g, int metric_a, int metric_b, uint64_t unit_begin, uint64_t unit_end,
                  tlm_pathset **out2, tlm_counters *ctr2);

/** The reference's four builder calls of main() (src/Main.cpp:237-243: buildMonteCarlo under two metrics, then
 *  buildDeterministic under the same two) as ONE call: the deterministic launch is queued on a second, low-priority
 *  stream behind the Monte-Carlo launch and fills the tail in which ever fewer Monte-Carlo waves are still walking.
 *  out4 / counters4 (may be null): MC metric_a, MC metric_b, DET metric_a, DET metric_b -- the same Path sets,
 *  element for element, as tlm_walk_mc2 + tlm_walk_det2 return. */
int tlm_walk_builders(tlm_ctx *ctx, const tlm_graph *g, int metric_a, int metric_b, const tlm_walk_params *p,
                      uint64_t mc_unit_begin, uint64_t mc_unit_end, uint64_t det_unit_begin, uint64_t det_unit_end,
                      tlm_pathset **out4, tlm_counters *counters4);

/* (ABI 4) The same walks over a LIST of units -- host array, strictly ascending unit ids -- instead of a range: what a rank
 * of a SLICED partition walks (tlm_dist_unit_share: a slice of every anchor's trials / first edges).  The Path sets hold
 * exactly those units, ordered by unit; unit_stats are in list order.  metric_b < 0: one metric (out2[0] / ctr2[0] only).
 * Same results, unit for unit, as the range calls (reference PathManager.cpp:20-29,194-204: the loops these units come from). */
int tlm_walk_list(tlm_ctx *ctx, const tlm_graph *g, int deterministic, int metric_a, int metric_b, const tlm_walk_params *p,
                  uint64_t n_units, const uint64_t *units, tlm_pathset **out2, tlm_counters *ctr2);
int tlm_walk_builders_list(tlm_ctx *ctx, const tlm_graph *g, int metric_a, int metric_b, const tlm_walk_params *p,
                           uint64_t n_mc_units, const uint64_t *mc_units, uint64_t n_det_units, const uint64_t *det_units,
                           tlm_pathset **out4, tlm_counters *counters4);

/* ---- a9: packed Path set -------------------------------------------------------------------- */
/* path i = nodes[off[i] .. off[i+1]); slots[k] = position, inside nodes[k]'s list, of the edge to
 * nodes[k+1] (0xFFFFFFFF at a path's last node); unit[i] ascending; length[i] = Path::length(). */
int tlm_pathset_sizes(const tlm_pathset *ps, uint64_t *n_paths, uint64_t *n_nodes_total, uint64_t *n_units);
/* Copy into caller buffers, device (dst_is_device != 0) or host; NULL pointers are skipped.
 * unit/length/off: uint64/int64/uint64 [n_paths], off has n_paths+1; nodes/slots: uint32 [n_nodes_total]. */
int tlm_pathset_export(const tlm_pathset *ps, int dst_is_device, uint64_t *unit, int64_t *length, uint64_t *off,
                       uint32_t *nodes, uint32_t *slots);
/* What grouping and bucketing need of every path, without its node list (HOST buffers of n_paths entries, any may be
 * NULL): first and last node (PathManager.cpp:633), node count, and a 64-bit hash of the node sequence -- two paths
 * with different (n_nodes, hash) differ; equal ones are compared node by node (Path::operator==, Path.cpp:46-49). */
int tlm_pathset_summary(tlm_ctx *ctx, const tlm_pathset *ps, uint32_t *front, uint32_t *back, uint32_t *n_nodes,
                        uint64_t *hash);
/* The node and slot lists of the paths index[0..n) (HOST), concatenated in that order into HOST buffers of
 * sum(n_nodes[index[k]]) entries. */
int tlm_pathset_fetch(tlm_ctx *ctx, const tlm_pathset *ps, uint64_t n, const uint64_t *index, uint32_t *nodes,
                      uint32_t *slots);
/* Per-unit outcomes, HOST buffer of n_units entries. */
int tlm_pathset_unit_stats(const tlm_pathset *ps, tlm_unit_stat *stats_host);
void tlm_pathset_destroy(tlm_pathset *ps);

/* ---- post-passes on Path sets, on the device -------------------------------- */
/* paths_ after several builder calls on one PathManager (Main.cpp:237-243): the sets in call order; unit[i] gets the
 * position k of its set in the top byte (k << 56).  n_units of the result is 0. */
int tlm_pathset_concat(tlm_ctx *ctx, int n, tlm_pathset *const *sets, tlm_pathset **out);
/* PathManager::filterUnique (PathManager.cpp:432-438): std::unique with Path::operator== (Path.cpp:46-49: same
 * node sequence) -- only ADJACENT duplicates go, the first of a run stays. */
int tlm_pathset_filter_unique(tlm_ctx *ctx, const tlm_pathset *in, tlm_pathset **out);
/* The same, CONSUMING `in`: when there is no adjacent duplicate (the usual case) *out is `in` itself -- no second copy of
 * a 10 GB Path set -- otherwise the filtered set, and `in` has been destroyed. */
int tlm_pathset_filter_unique_take(tlm_ctx *ctx, tlm_pathset *in, tlm_pathset **out);
/* PathManager::getPathsBetweenAnchors (PathManager.cpp:626-645): HOST buffers of n_paths entries each;
 * order[j] = path index, key[j] = (front node << 32) | back node, sorted by key (the std::map's order over
 * (front Node*, back Node*)), inside one key in path order.  A bucket is a run of equal keys. */
int tlm_pathset_anchor_pairs(tlm_ctx *ctx, const tlm_pathset *ps, uint32_t *order_host, uint64_t *key_host);

/* ---- PAF text -> records on the device ------------------------------------------------- */
/* OverlapGraph::load's parse loop (OverlapGraph.cpp:106-128) for CLEAN text: every line 16 whitespace-separated
 * tokens, the nine numeric ones plain [+-]digits inside int range, the strand one character, no blank lines -- there
 * line-wise parsing is the reference's token stream.  The text goes to the device (pinned slabs, parallel reads) and is
 * tokenised there; names are numbered in first-seen order over the files (query before target), the numbering a
 * sequential reader assigns.  *clean = 0: the text is something else; the host falls back to its exact token-stream
 * reader for the whole graph.  A handle that has refused a file (or on which an add failed) is good for tlm_paf_destroy
 * only: every later add, tlm_paf_text_begin and tlm_paf_finish on it returns TLM_EINVAL (the refused file's names are in
 * the handle's dictionary; numbering the other files' names past them would be wrong). */
typedef struct tlm_paf tlm_paf;
int tlm_paf_create(tlm_ctx *ctx, tlm_paf **out);
void tlm_paf_destroy(tlm_paf *p);
/* One file of the graph, in load order (at most 4).  `anchors` is OverlapGraph::load's second argument. */
int tlm_paf_add_file(tlm_paf *p, const char *path, int anchors, uint64_t *n_records, int *clean);
int tlm_paf_add_text(tlm_paf *p, const char *text_host, uint64_t bytes, int anchors, uint64_t *n_records, int *clean);

/* The same with the text brought to the device by ANOTHER process (an upload helper: telomeri_hip starts one): the file's
 * mapping, its registration with the device and the taking down of both belong to that process.  Unmapping 15.6 GB costs
 * 0.55 s of one core under the process's address-space lock and disturbs the device queues of the process that does it
 * -- the walking process pays neither.  The processes share a window of TLM_UPLOAD_SLOTS slots of 1 GiB of device memory:
 *   this process : tlm_upload_window_create (once; the HIP IPC handle goes to the helper), then per file
 *                  tlm_paf_text_begin, tlm_paf_text_piece(k) whenever the helper reports piece k (1 GiB) in its slot
 *                  k mod TLM_UPLOAD_SLOTS -- copies it to its place and frees the slot --, tlm_paf_text_commit (tokenises
 *                  like tlm_paf_add_file) or tlm_paf_text_abort;
 *   the helper   : tlm_upload_window_open (once), then per file tlm_upload_file_window, which calls `arrived(k)` when piece
 *                  k is in its slot and `slot_free(k)` before it overwrites a slot (to wait until the other side has
 *                  taken piece k - TLM_UPLOAD_SLOTS over); either returning non-zero gives the upload up. */
#define TLM_IPC_HANDLE_BYTES 64
#define TLM_UPLOAD_SLOTS 3
typedef int (*tlm_upload_piece_fn)(uint64_t piece, void *user);
int tlm_upload_window_create(tlm_ctx *ctx, uint8_t *handle_out, uint64_t *bytes_out);
int tlm_paf_text_begin(tlm_paf *p, uint64_t bytes);
int tlm_paf_text_piece(tlm_paf *p, uint64_t piece);
int tlm_paf_text_commit(tlm_paf *p, int anchors, uint64_t *n_records, int *clean);
void tlm_paf_text_abort(tlm_paf *p);
int tlm_upload_window_open(tlm_ctx *ctx, const uint8_t *handle);
int tlm_upload_file_window(tlm_ctx *ctx, const char *path, uint64_t bytes, tlm_upload_piece_fn arrived,
                           tlm_upload_piece_fn slot_free, void *user);
/* After the last file: final name ids.  n_names names, blob_bytes bytes of name text in all. */
int tlm_paf_finish(tlm_paf *p, uint32_t *n_names, uint64_t *blob_bytes);
/* HOST buffers (any may be NULL): name i = blob[name_off[i] .. name_off[i + 1]); is_ctg[i] = Utils::startsWithInsensitive(name, "ctg") */
int tlm_paf_names(const tlm_paf *p, uint32_t *name_off, char *blob, uint8_t *is_ctg);
/* The columns of file `file` (DEVICE pointers owned by `p`; q_id / t_id are the final name ids). */
int tlm_paf_records(const tlm_paf *p, int file, tlm_records *out);
/* tlm_graph_build over the files of `p` (the handle may be destroyed afterwards). */
int tlm_paf_build_graph(tlm_paf *p, const tlm_filter_params *fp, tlm_graph **out);
/* host wall-clock of the stages so far: text upload, tokenising, finish */
int tlm_paf_timing(const tlm_paf *p, double *upload_ms, double *parse_ms, double *finish_ms);

/* ---- multi-GPU: one process per GPU, ONE exchange ---------------------------------------- */
/* The graph is replicated (every rank builds it from the same files); rank r of `world` walks the units of the anchors
 * tlm_dist_anchor_range gives it (contiguous bundles of ceil(n_anchors / world) anchors: MC units
 * [a0 * rebuild_attempts, a1 * rebuild_attempts), DET units [det_off[a0], det_off[a1]) of tlm_graph_anchor_table) with no
 * collective in between; then every rank calls tlm_pathset_allgather once and holds paths_ as one PathManager would
 * after the reference's builder calls (Main.cpp:237-243), element for element.  The transport is RCCL (grouped
 * ncclSend / ncclRecv between every pair of ranks, each block received in its final place), loaded on first use. */
#define TLM_UNIQUE_ID_BYTES 128
#define TLM_EXCHANGE_MAX_BLOCKS 64
/* Rank 0 makes the id (ncclGetUniqueId); the host hands the 128 bytes to the other ranks (a file, a socket, the
 * launcher); then every rank calls tlm_dist_init with the same bytes.  One communicator per context.
 * The library binds RCCL at run time: librccl.so.1, or the library the environment variable TLM_RCCL_LIB names (tests use
 * it to put a stand-in underneath that lets several ranks share one GPU, which RCCL refuses: tests/fake_rccl/).
 * FAILING TOGETHER: the exchange calls below are collectives.  A rank that cannot go on (out of memory, a HIP error)
 * tells its peers before any data moves -- the ranks swap their status in a 4-byte-per-rank all-gather ahead of every data
 * collective -- and then ALL ranks return: the failing one its own code, the others TLM_EPEER.  A failure in the middle of
 * a collective sequence closes the open group, aborts the communicator (ncclCommAbort) and leaves the context out of the
 * job: its later exchange calls return TLM_EPEER at once; the launcher ends the remaining ranks (run.sh does). */
int tlm_dist_unique_id(uint8_t id_out[TLM_UNIQUE_ID_BYTES]);
int tlm_dist_init(tlm_ctx *ctx, int rank, int world, const uint8_t id[TLM_UNIQUE_ID_BYTES]);
void tlm_dist_finalize(tlm_ctx *ctx);
int tlm_dist_rank(const tlm_ctx *ctx);          /* 0 without tlm_dist_init */
int tlm_dist_world(const tlm_ctx *ctx);         /* 1 without tlm_dist_init */
int tlm_dist_anchor_range(const tlm_graph *g, int rank, int world, uint32_t *anchor_begin, uint32_t *anchor_end);
/* (ABI 4) The SLICED partition: rank r walks, of EVERY anchor, trials [rb r / world, rb (r + 1) / world) (deterministic = 0)
 * or the same slice of the anchor's first-edge units (deterministic = 1).  Every rank gets the same share of every anchor's
 * work, whatever the anchors weigh (contigs of 20 kb and of 2 Mb, hubs): no pilot run, no collective.  `units` (host,
 * ascending; NULL = count only) feeds tlm_walk_list / tlm_walk_builders_list.  The exchange calls return paths_ order for
 * any partition whose ranks own disjoint units. */
int tlm_dist_unit_share(const tlm_graph *g, const tlm_walk_params *p, int rank, int world, int deterministic, uint64_t *units,
                        uint64_t *n_units);

typedef struct {
    uint64_t bytes_sent;           /* by this rank, to all peers together */
    uint64_t bytes_received;
    uint32_t slot_bits;            /* 16, or 32 if some neighbour list has 65 535 entries or more */
    double device_ms;              /* HIP events around the exchange on the context's stream */
    double wall_ms;
} tlm_exchange_info;
/* (ABI 5) THE SHARDED LOAD.  OverlapGraph::load (OverlapGraph.cpp:94-131) reads every overlap file whole; with N ranks on
 * one node that is N uploads of the same text through one host's memory bus, and the load does not shrink with N.  Here
 * rank r brings only ITS pieces of each file to its device -- pieces of TLM_TEXT_PIECE bytes, the contiguous run
 * tlm_dist_text_pieces names -- and the ranks swap their runs device to device (every rank sends its run to every peer and
 * receives every peer's run straight into its place in the text buffer, one grouped exchange over the xGMI mesh).  From
 * there on every rank holds the SAME BYTES a single process holds: tokeniser, name numbering (first-seen order), filter,
 * graph build run exactly as at world 1, so the graph is the one-process graph by construction.
 *   tlm_paf_add_file            does all of it by itself when the context has a communicator of more than one rank
 *                               (TLM_LOAD_REPLICATED=1 in the environment: every rank uploads the whole file, as before);
 *   with an upload helper       tlm_paf_text_begin, tlm_upload_file_window_range on the helper's side for the pieces
 *                               [piece_begin, piece_end) and tlm_paf_text_piece for each of them, then tlm_paf_text_share
 *                               (COLLECTIVE: every rank of the communicator calls it for the same file; `failed` != 0 =
 *                               this rank could not bring its pieces: every rank then returns an error, none hangs), then
 *                               tlm_paf_text_commit. */
#define TLM_TEXT_PIECE (1ull << 30)
int tlm_dist_text_pieces(const tlm_ctx *ctx, uint64_t bytes, uint64_t *piece_begin, uint64_t *piece_end);
/* the same rule as plain arithmetic (no context, no device: tests check it on the CPU): pieces [n r / world, n (r + 1) / world)
 * of the n = ceil(bytes / TLM_TEXT_PIECE) pieces, as bytes of the text */
int tlm_dist_text_run(uint64_t bytes, int rank, int world, uint64_t *byte_begin, uint64_t *byte_end);
int tlm_paf_text_share(tlm_paf *p, int failed, tlm_exchange_info *info);
int tlm_upload_file_window_range(tlm_ctx *ctx, const char *path, uint64_t bytes, uint64_t piece_begin, uint64_t piece_end,
                                 tlm_upload_piece_fn arrived, tlm_upload_piece_fn slot_free, void *user);
/* what the last tlm_paf_add_file / tlm_paf_text_share of this handle moved between the ranks (zeros at world 1) */
int tlm_paf_share_info(const tlm_paf *p, tlm_exchange_info *info);

/* sets[0..n_sets): this rank's Path sets, one per builder call, in call order (the same n_sets on every rank).
 * *out: the sets of all ranks, builder by builder, inside a builder in rank order (= unit order); unit[i] carries the
 * builder in its top byte as tlm_pathset_concat does.  With world == 1 this IS tlm_pathset_concat. */
int tlm_pathset_allgather(tlm_ctx *ctx, int n_sets, tlm_pathset *const *sets, tlm_pathset **out, tlm_exchange_info *info);
/* The LAZY form of the exchange.  What the host stages read of EVERY path (Main.cpp:247-352: length, end nodes, equality)
 * travels as a 45-byte summary in one ncclAllGather; node lists stay on the rank that built them and are broadcast by
 * their owner when asked for.  `sets` as for tlm_pathset_allgather (they may be destroyed afterwards: the handle keeps
 * this rank's paths).  world == 1: no communication. */
typedef struct tlm_gathered tlm_gathered;
int tlm_pathset_gather_summaries(tlm_ctx *ctx, int n_sets, tlm_pathset *const *sets, tlm_gathered **out, tlm_exchange_info *info);
void tlm_gathered_destroy(tlm_gathered *g);
int tlm_gathered_sizes(const tlm_gathered *g, uint64_t *n_paths, uint64_t *n_local_paths);
/* HOST arrays of n_paths entries (any may be NULL), in paths_ order (builder by builder, rank order inside): unit (builder
 * in the top byte), Path::length(), first and last node, node count, hash of the node sequence (tlm_pathset_summary),
 * owning rank, dup_of_prev = 1 iff the path has the node sequence of its predecessor IN THE SAME BLOCK (exact: compared
 * where the paths are), block_first = 1 for the first path of a (rank, builder) block -- whose predecessor in paths_
 * order lives elsewhere: PathManager::filterUnique (PathManager.cpp:432-438) settles those few through tlm_gathered_fetch
 * when the two summaries coincide. */
int tlm_gathered_export(const tlm_gathered *g, uint64_t *unit, int64_t *length, uint32_t *front, uint32_t *back,
                        uint32_t *n_nodes, uint64_t *hash, uint32_t *owner, uint8_t *dup_of_prev, uint8_t *block_first);
/* COLLECTIVE: every rank calls it with the SAME index list (positions in paths_ order); every rank receives the node and
 * slot lists of those paths, concatenated in request order, in HOST buffers of sum(n_nodes[index[k]]) entries. */
int tlm_gathered_fetch(tlm_ctx *ctx, const tlm_gathered *g, uint64_t n, const uint64_t *index, uint32_t *nodes, uint32_t *slots);
/* The placement rule of that exchange, as host arithmetic: counts[(r * n_sets + b) * 2 + {0, 1}] = paths, nodes of rank
 * r's set b -> where that block starts in the result (same (r, b) indexing), and the result's totals[2]. */
int tlm_exchange_layout(int world, int n_sets, const uint64_t *counts, uint64_t *path_base, uint64_t *node_base,
                        uint64_t *totals);

#ifdef __cplusplus
}
#endif
#endif

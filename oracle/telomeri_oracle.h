/* oracle/telomeri_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's overlap-graph path-construction hot path:
 *   OverlapGraph::load / filter / buildFrom        reference src/OverlapGraph.cpp:97-258,297-324
 *   PathManager::buildMonteCarlo                   reference src/PathManager.cpp:12-177
 *   PathManager::buildDeterministic                reference src/PathManager.cpp:179-386
 *   Path::updateLength                             reference src/Path.cpp:6-16
 *   Utils::getMetric                               reference include/Utils.hpp:47-60
 * plus libstdc++'s std::mt19937 / generate_canonical<double,53> (GCC 11.4,
 * bits/random.tcc:3348-3380), the only third-party arithmetic on the path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * PINNED: tests/test_oracle_vs_reference.py checks it against the unmodified reference
 * (oracle/_ref/ref_dump) when that binary is present, and tests/test_oracle_golden.py against
 * the fixtures under tests/golden/ that the same binary produced (tests/golden/make_golden.py).
 *
 * Stated deviations from the reference (results identical):
 *   - names are interned through a hash table instead of the O(N) reverse scan (nodeIndex);
 *   - adjacency is CSR (per-node order = record order) instead of vector<Edge> per node;
 *   - path length is kept incrementally instead of recomputed from scratch every step;
 *   - seed_mode 1 seeds one mt19937 per walk (tlo_walk_seed) instead of sharing one stream.
 */
#ifndef TELOMERI_ORACLE_H
#define TELOMERI_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { TLO_MODE_MIN = 0, TLO_MODE_MAX = 1, TLO_MODE_AVG = 2, TLO_MODE_SUM = 3 };  /* OverlapGraph.hpp:56-65 */
enum { TLO_METRIC_ES = 0, TLO_METRIC_OS = 1, TLO_METRIC_ES_SQRT = 2, TLO_METRIC_OS_SQRT = 3 }; /* Utils.hpp:43-45 */
enum { TLO_SEED_SHARED = 0, TLO_SEED_PER_WALK = 1 };

typedef struct {
    int32_t mode;
    uint64_t min_overlap_length;
    float min_overlap_percentage;
    uint64_t max_overhang_length;
    float max_overhang_percentage;
} tlo_filter_params;                       /* OverlapGraph.hpp:67-78 */

typedef struct {
    int32_t rebuild_attempts;
    int32_t backtrack_attempts;
    int64_t len_threshold;
} tlo_walk_params;                         /* PathManager.hpp:27-31 */

typedef struct {
    uint64_t units;      /* walk units executed */
    uint64_t steps;      /* selection scans (MC) / ranked steps (DET) */
    uint64_t evals;      /* adjacency entries examined by those scans */
    uint64_t draws;      /* uniform doubles drawn */
    uint64_t accepted;   /* paths appended */
} tlo_counters;

/* Read-only view of the graph as CSR; pointers stay valid until the graph is changed or freed. */
typedef struct {
    uint32_t n_nodes;
    uint64_t n_adj;
    const uint64_t *row_ptr;       /* [n_nodes+1] */
    const uint32_t *nbr;           /* q_index of the entry (the neighbour); owner = t_index */
    const uint32_t *q_start, *q_end, *t_start, *t_end;
    const float *os, *si, *es;
    const uint8_t *strand;
    const uint32_t *node_len;
    const uint8_t *node_anchor;
    uint64_t n_edges;              /* edges_.size() == kept records */
} tlo_csr;

/* Packed Path set: path i covers nodes[off[i] .. off[i+1]); slots[k] is the
 * position inside nodes[k]'s list of the edge to nodes[k+1] (0xFFFFFFFF at a path's last node). */
typedef struct {
    uint64_t n_paths;
    uint64_t *unit;                /* walk unit that produced the path (ascending) */
    int64_t *length;               /* Path::length() */
    uint64_t *off;                 /* [n_paths+1] */
    uint32_t *nodes;
    uint32_t *slots;
} tlo_paths;

/* Per-unit outcome, for parity checks against the device path. */
typedef struct {
    uint8_t status;                /* 0 abandoned, 1 accepted, 2 reached an anchor but length < 0 (not appended) */
    uint32_t n_nodes;              /* nodes on the path when the walk ended */
    uint32_t steps;
    uint64_t evals;
    uint32_t draws;
    int64_t length;                /* last computed Path::length() */
} tlo_unit_stat;

typedef struct tlo_graph tlo_graph;

tlo_graph *tlo_graph_new(void);
void tlo_graph_free(tlo_graph *g);

/* a2 + a3: filter decision and SI / OS / ES per record (scores are written for every record). */
void tlo_filter_score(const tlo_filter_params *p, size_t n, const int32_t *q_len, const int32_t *q_start,
                      const int32_t *q_end, const int32_t *t_len, const int32_t *t_start, const int32_t *t_end,
                      const int32_t *matches, const uint8_t *same_name, uint8_t *keep, float *si, float *os,
                      float *es);

/* OverlapGraph::load on a PAF text file; returns 0 on success, -1 if the file cannot be opened.
 * records_out (may be NULL) receives the number of records parsed ("ctr", OverlapGraph.cpp:128). */
int tlo_graph_load_paf(tlo_graph *g, const char *path, int anchors, const tlo_filter_params *p,
                       uint32_t test_load_num, uint64_t *records_out);

/* The same on integer columns: names are external ids, name_is_ctg[id] = "name starts with ctg". */
int tlo_graph_add_records(tlo_graph *g, int anchors, const tlo_filter_params *p, size_t n, const int32_t *q_id,
                          const int32_t *t_id, const uint8_t *name_is_ctg, const int32_t *q_len,
                          const int32_t *q_start, const int32_t *q_end, const int32_t *t_len,
                          const int32_t *t_start, const int32_t *t_end, const int32_t *matches,
                          const int32_t *strand);

/* Adopt a finished CSR (bench.py's cpu_baseline leg: the CSR the device built, copied to the host). */
int tlo_graph_from_csr(tlo_graph *g, uint32_t n_nodes, uint64_t n_adj, const uint64_t *row_ptr, const uint32_t *nbr,
                       const uint32_t *q_start, const uint32_t *t_start, const float *os, const float *es,
                       const uint32_t *node_len, const uint8_t *node_anchor);

int tlo_graph_csr(tlo_graph *g, tlo_csr *out);
const char *tlo_graph_node_name(tlo_graph *g, uint32_t node);   /* text mode only, else "" */
int64_t tlo_graph_node_ext_id(tlo_graph *g, uint32_t node);      /* column mode only, else -1 */

/* Walk units.  MC: unit = anchor_rank * rebuild_attempts + r.  DET: units enumerate, anchor by anchor
 * in node order, the first-edge positions the reference's loop visits (it stops after the first
 * edge that leads straight to an anchor, PathManager.cpp:224-229). */
uint64_t tlo_mc_units(tlo_graph *g, const tlo_walk_params *p);
uint64_t tlo_det_units(tlo_graph *g);

uint32_t tlo_walk_seed(int metric, uint32_t anchor_node, uint32_t r);

/* Run units [unit_begin, unit_end).  stats (may be NULL) has unit_end-unit_begin entries.
 * threads > 1 is only allowed for seed_mode PER_WALK (MC) and for DET.  Returns NULL on error. */
tlo_paths *tlo_walk_mc(tlo_graph *g, int metric, const tlo_walk_params *p, int seed_mode, uint64_t unit_begin,
                       uint64_t unit_end, int threads, tlo_counters *ctr, tlo_unit_stat *stats);
tlo_paths *tlo_walk_det(tlo_graph *g, int metric, uint64_t unit_begin, uint64_t unit_end, int threads,
                        tlo_counters *ctr, tlo_unit_stat *stats);
void tlo_paths_free(tlo_paths *p);

/* First n outputs of std::mt19937(seed), and n doubles of uniform_real_distribution<double>(0,1). */
void tlo_mt19937_raw(uint32_t seed, size_t n, uint32_t *out);
void tlo_mt19937_canonical(uint32_t seed, size_t n, double *out);

#ifdef __cplusplus
}
#endif
#endif

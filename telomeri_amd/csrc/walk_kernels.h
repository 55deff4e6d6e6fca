// telomeri_amd/csrc/walk_kernels.h -- device side of the contig-to-contig walks (gfx950).
//
// One 64-lane wavefront per live walk (north_star): the neighbour list of the current node is
// read as coalesced 8-byte {neighbour|anchor, metric} entries, lane l holding K consecutive entries
// (K = 4 / 6 / 8 by list length: aligned 16-byte loads, the list read once per step); the
// forced-anchor test, the visited test, the roulette prefix sums (MC) and the rank arg-max (DET)
// are wave ballots / DPP scans / DPP reductions.  Waves are persistent and pull walk units from one
// atomic cursor per XCD, so divergent walk lengths do not idle the chip; every wave leaves when all
// cursors have passed the end of the work list.  No MFMA: integer / fp32 / fp64 work on HBM/L2 gathers.
//
//   MC  : PathManager::buildMonteCarlo     reference src/PathManager.cpp:12-177
//   DET : PathManager::buildDeterministic  reference src/PathManager.cpp:179-386
//   length bookkeeping : Path::updateLength  reference src/Path.cpp:6-16
//   RNG : std::mt19937 + generate_canonical<double,53>  (libstdc++ 11, bits/random.tcc:3348-3380)
#ifndef TLM_WALK_KERNELS_H
#define TLM_WALK_KERNELS_H

#include "tlm_common.h"

// Outcome of one walk unit (device -> host), one record per unit of a call.
struct UnitOut {
    uint64_t evals;
    uint64_t off;          // arena offset of the accepted path
    int64_t len;           // last Path::length()
    uint32_t nnodes, steps, draws;
    uint32_t status;       // ST_*
};

// Kernel arguments.  Kept small on purpose: every pointer is two SGPRs that stay live across the walk loop.
struct WalkArgs {
    // graph (tlm_graph, walk layout: rows padded to 8 entries, nodes in locality order)
    const uint2 *rows;              // per node {row start, degree | order-safety flags << 24}
    const uint2 *ents;              // w_ent_es or w_ent_os
    const uint2 *qt;                // per entry {q_start, t_start}: the length increment of Path::updateLength
    const uint32_t *met;            // compact rows: fp32 bits of the metric per entry (w_m_es or w_m_os) ...
    const uint16_t *d16;            // ... and neighbour - owner + 32768 per entry; same index space as `ents`
    const uint32_t *met1;           // slot 1
    const uint32_t *node_len;       // reference order
    const uint32_t *old_of_new;     // reference node index of a walk-order node (results are reported in those)
    uint32_t n_nodes;
    uint32_t flag_bit;              // 1 << metric
    int32_t metric;
    // a launch may carry the same units under a second metric ("slot" 1): work item w = slot * n_units + unit
    const uint2 *ents1;
    uint32_t flag_bit1;
    int32_t metric1;
    uint32_t n_units;               // units per slot
    // units
    const uint32_t *anchors;        // reference node index of each anchor (seeds, first node's length)
    const uint32_t *w_anchors;      // walk-order index of each anchor
    uint32_t n_anchors;
    const uint64_t *det_off;
    uint64_t unit_begin;
    uint32_t n_work;
    uint32_t chunk;                 // consecutive work items that share a start anchor: dealt to one XCD's queue
    const uint32_t *work_list;      // unit - unit_begin per work item, or null = identity
    int32_t rebuild_attempts, backtrack_attempts;
    int64_t len_threshold;
    // per-wave scratch: [wave][3][cap] = path nodes, path edge indices, mark log
    uint32_t *scratch;
    uint32_t cap;
    uint32_t *bitmaps;
    uint32_t bm_words;              // 32-bit words per wave: the N-bit visited map, then WALK_RNG_WORDS of generator state
    // arena cursor (u64) at cursor[2..3]; launch span at [6..9]; ACCOUNT builds: layout bytes (u64) at [10..11] and ten
    // u64 step-class counters at [12..31]; one work
    // cursor per XCD at cursor[WALK_QUEUE_BASE + WALK_QUEUE_STRIDE * x]
    uint32_t *cursor;
    UnitOut *units;
    uint2 *arena;                   // accepted paths: {reference node, edge slot}
    uint64_t ar_cap;
    // host-mapped word, or null: set to 1 by the workgroup that finds every workgroup of the launch started (counted in
    // cursor[1]); the host queues the deterministic launch of tlm_walk_builders only then
    uint32_t *resident_flag;
    // sizes the checked build (-DTLM_CHECKED) compares indices with; unused kernel arguments cost no registers
    uint64_t n_ent;                 // entries of the walk-layout arrays (ents / met / d16 / qt): zero block + padded rows
    uint32_t n_work_all;            // entries of `units`
    // (ABI 4) the units of the call as an ascending LIST instead of the range [unit_begin, unit_begin + n_units): unit of
    // local index ui = unit_ids[ui]; null = the range.  Read once per walk, where the unit becomes (anchor, trial / edge).
    const uint64_t *unit_ids;
};

constexpr int WALK_RNG_WORDS = 640;             // raw std::mt19937 state of a wave (624 words), kept in HBM between twists
constexpr int WALK_XCDS = 8;                    // MI355X: 8 accelerator dies, each with its own L2
constexpr int WALK_QUEUE_BASE = 32;             // u32 index of XCD 0's work cursor; one 64-byte line per cursor
constexpr int WALK_QUEUE_STRIDE = 16;
constexpr int WALK_CURSOR_WORDS = WALK_QUEUE_BASE + WALK_QUEUE_STRIDE * WALK_XCDS;
#ifndef TLM_WALK_BLOCK
#define TLM_WALK_BLOCK 256
#endif
constexpr int WALK_BLOCK = TLM_WALK_BLOCK;      // independent waves of a workgroup (4 by default)
constexpr int WALK_WAVES_PER_BLOCK = WALK_BLOCK / 64;

// resident workgroups per CU of the kernel (occupancy query): the persistent grid is sized from it
int tlm_walk_blocks_per_cu(bool is_mc, bool sqrt_metric, bool lite, bool account);
// account: the build of the kernel that also sums the layout bytes its steps ask for into cursor[10..11] (tlm_counters.layout_bytes)
// -> 0; the checked build (-DTLM_CHECKED) waits for the launch and returns 1, with the error text set, when a kernel formed
// an index outside one of its arrays (the access itself was skipped)
// `lite`: no list of more than 512 entries, no order-unsafe row for the launch's metrics (the host has looked)
int tlm_launch_walk_mc(hipStream_t s, unsigned blocks, bool sqrt_metric, const WalkArgs &a, bool account, bool lite);
int tlm_launch_walk_det(hipStream_t s, unsigned blocks, bool sqrt_metric, const WalkArgs &a, bool account, bool lite);

#endif

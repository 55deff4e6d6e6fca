// telomeri_amd/csrc/walk_kernels.hip -- see walk_kernels.h.
#include "walk_kernels.h"
#include <type_traits>

// Diagnostic build (make EXTRA=-DTLM_TIMING): shader-clock cycles per phase of a Monte-Carlo step, summed over all
// waves of a launch and printed on stderr by tlm_launch_walk_mc.  The phases are separated by full vmcnt waits, so
// the build is slower than the product and its results are for reading, not for shipping.
#ifdef TLM_TIMING
__device__ unsigned long long tlm_timing[8];
#endif

// Checked build (make EXTRA=-DTLM_CHECKED, a separate library that tests load through TLM_LIB): every index the walk kernels
// form into a global array -- list entries, row records, {q_start, t_start}, the wave's path / log / visited-map scratch,
// the arena, the unit records -- is compared with its array's size first; an index outside is NOT dereferenced (a GPU
// memory fault can take the whole host's GPUs down), it is recorded (first offender: site, index, limit) and the launch
// is reported as failed by the host.  The product build compiles the checks away.
#ifdef TLM_CHECKED
struct ChkLimits {
    unsigned long long n_ent;      // entries of the walk-layout arrays (rows, zero block and padding included)
    unsigned n_nodes, n_anchors;
};
__device__ ChkLimits tlm_chk;
__device__ unsigned long long tlm_oob[4];       // first offender: site, index, limit; [3] = offences counted
#ifndef TLM_CHECK_REPORT_INLINE
#define TLM_CHECK_REPORT_INLINE __noinline__
#endif
__device__ TLM_CHECK_REPORT_INLINE void tlm_oob_report(unsigned site, unsigned long long idx, unsigned long long lim) {
    if (atomicCAS(&tlm_oob[0], 0ull, (unsigned long long) site) == 0ull) {
        tlm_oob[1] = idx;
        tlm_oob[2] = lim;
    }
    atomicAdd(&tlm_oob[3], 1ull);
}
#ifdef TLM_CHECK_OFF        /* bisecting the checked build itself */
#define TLM_IN(site, idx, lim) true
#else
#ifndef TLM_CHECK_SITES
#define TLM_CHECK_SITES 0xFFFFFFFFu
#endif
#define TLM_IN(site, idx, lim) \
    ((((TLM_CHECK_SITES) >> (site)) & 1u) == 0u || (unsigned long long) (idx) < (unsigned long long) (lim) ? true : (tlm_oob_report((site), (idx), (lim)), false))
#endif
#define TLM_N_ENT (tlm_chk.n_ent)
#define TLM_N_NODES (tlm_chk.n_nodes)
#else
#define TLM_IN(site, idx, lim) true
#define TLM_N_ENT 0ull
#define TLM_N_NODES 0u
#endif
enum {      // sites of the checked build
    CHK_ROWS = 1, CHK_LIST8 = 2, CHK_LISTC = 3, CHK_CHUNK = 4, CHK_SERIAL = 5, CHK_BACKPROBE = 6, CHK_QT = 7, CHK_PATH = 8,
    CHK_LOG = 9, CHK_MAP = 10, CHK_ARENA = 11, CHK_UNIT = 12, CHK_NODE_TABLE = 13, CHK_ANCHOR = 14, CHK_BAND = 15
};

namespace {

struct StepClock {
#ifdef TLM_TIMING
    uint64_t t, acc[5], steps;
    __device__ __forceinline__ void start() {
        for (int i = 0; i < 5; i++) acc[i] = 0;
        steps = 0;
        t = __builtin_amdgcn_s_memtime();
    }
    __device__ __forceinline__ void mark(int phase, bool wait_loads) {     // closes `phase`
#ifdef TLM_TIMING_ROW1
        if (wait_loads && phase == 0) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");      // leave the youngest load out
        else
#endif
        if (wait_loads) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint64_t now = __builtin_amdgcn_s_memtime();
        acc[phase] += now - t;
        t = now;
    }
    __device__ __forceinline__ void flush(int lane) {
        if (lane == 0) {
            for (int i = 0; i < 5; i++) atomicAdd(&tlm_timing[i], (unsigned long long) acc[i]);
            atomicAdd(&tlm_timing[5], (unsigned long long) steps);
        }
    }
#else
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int, bool) {}
    __device__ __forceinline__ void flush(int) {}
#endif
};
enum { CLK_ROW_WAIT = 0, CLK_LIST_WAIT = 1, CLK_SELECT = 2, CLK_ADVANCE = 3, CLK_OTHER = 4 };

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) v); }

// Kernel arguments that only the start and the end of a walk need (work cursor, anchor tables, unit records, arena ...)
// are read from the kernarg segment where they are used, through a pointer the optimiser cannot see through: loads
// from it cannot be hoisted above the walk loop, so they do not occupy scalar registers while a walk is stepping.
// WalkArgs is the only kernel parameter (offset 0 of the segment).
// ONLY FOR CODE THAT IS INLINED INTO THE KERNEL: in a function that stays a real call the intrinsic below yields a NULL
// pointer (a callee is not handed the kernarg segment), and every "cold" argument is then read from address 0 -- round 3's
// kernel that "died on a null address" when a check made mc_walk too large for the inliner, and round 4's first far filter
// (profiles/experiments/r04_far_filter_bisect.md).  Every function that calls cold_args() is __forceinline__ (mc_walk,
// det_walk, finish_unit, next_work, the kernels themselves), and tests/test_capi_library.py::test_cold_args_only_in_kernels
// reads the ISA: no function other than the kernels may contain the marker of the asm statement below.
typedef __attribute__((address_space(4))) const WalkArgs ColdArgs;
__device__ __forceinline__ ColdArgs *cold_args() {
    uint64_t p = (uint64_t) __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("; cold_args" : "+s"(p));
    return (ColdArgs *) p;
}

// A {u32, u32} record of a table that no kernel writes, at a wave-uniform index: through the constant address space the
// load becomes s_load_dwordx2 -- the scalar cache path, results in scalar registers, counted by lgkmcnt: it neither
// occupies vector registers around the loop nor takes part in the in-order return of the vector loads.
__device__ __forceinline__ uint2 scalar_load_u2(const uint2 *table, uint32_t index) {
    const uint64_t v = ((__attribute__((address_space(4))) const uint64_t *) table)[index];
    return make_uint2((uint32_t) v, (uint32_t) (v >> 32));
}

__device__ __forceinline__ uint2 row_record(const uint2 *rows, uint32_t node) {
    if (!TLM_IN(CHK_ROWS, node, TLM_N_NODES)) return make_uint2(0u, 0u);
    return scalar_load_u2(rows, node);
}
__device__ __forceinline__ uint2 qt_record(const uint2 *qt, uint32_t e) {
    if (!TLM_IN(CHK_QT, e, TLM_N_ENT)) return make_uint2(0u, 0u);
    return scalar_load_u2(qt, e);
}

// What the step loop itself reads (stays in scalar registers for the whole walk).
struct HotArgs {
    const uint2 *rows;
    const uint2 *ents;
    const uint2 *qt;
    const uint32_t *met;       // compact rows: metric bits ...
    const uint16_t *d16;       // ... and 16-bit relative neighbour ids
    uint32_t cap;
    uint32_t flag_bit;
    int32_t backtrack_attempts;
    int64_t len_threshold;
};

// Orders a wave's own LDS / global traffic between its lanes (lane 0 writes, all lanes read):
// the wave executes in lockstep and its memory instructions issue in order, so this only has to
// stop the compiler from moving accesses across it.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------
// std::mt19937 + uniform_real_distribution<double>(0,1), one generator per wave.
//
// A walk consumes one double per step, and a double is two raw words, tempered, combined and scaled.  That work is
// done once per twist for all 312 doubles of a generator period, by all 64 lanes: the wave's LDS block (2496 bytes)
// holds the 312 ready doubles, a draw is one 8-byte LDS read.  The raw state is only needed by the next twist (every
// 312 draws) and sits in the wave's HBM scratch in between.

typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(1))) uint32_t glb_u32;

__device__ __forceinline__ uint32_t walk_seed_dev(int metric, uint32_t anchor_node, uint32_t r) {
    // == tlm_walk_seed() / tlo_walk_seed()
    return 5489u + 0x9E3779B9u * (anchor_node + 1u) + 0x85EBCA6Bu * r + 0xC2B2AE35u * (uint32_t) metric;
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// In-place regeneration of the raw state in LDS, in chunks of 64 words.  Word i needs old mt[i], old mt[i+1] and
// mt[(i+397)%624], which is an OLD word for i < 227 (it lies >= 6 chunks ahead) and a NEW word for i >= 227 (it
// lies >= 3 chunks behind); word 623 needs the new mt[0].  Within a chunk all reads are issued before the chunk's
// writes.  Then: the new raw state goes to HBM, and every pair of words becomes its double, in place --
// uniform_real_distribution<double>(0,1)(gen): two draws, low word first, RN(x0 + x1 * 2^32) / 2^64, clamped
// below 1 (libstdc++ generate_canonical<double, 53>).
__device__ __forceinline__ void mt_twist_lds(lds_u32 *mt, glb_u32 *raw, int lane) {
    for (int base = 0; base < 624; base += 64) {
        const int i = base + lane;
        uint32_t v = 0;
        if (i < 624) {
            const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
            v = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        wave_sync();
        if (i < 624) mt[i] = v;
        wave_sync();
    }
    for (int j = lane; j < 312; j += 64) {
        const uint32_t r0 = mt[2 * j], r1 = mt[2 * j + 1];
        raw[2 * j] = r0;
        raw[2 * j + 1] = r1;
        const double sum = (double) mt_temper(r0) + (double) mt_temper(r1) * 4294967296.0;
        double ret = sum * 0x1p-64;                   // exact scaling
        if (ret >= 1.0) ret = 0x1.fffffffffffffp-1;   // nextafter(1, 0)
        mt[2 * j] = (uint32_t) __double2loint(ret);
        mt[2 * j + 1] = (uint32_t) __double2hiint(ret);
    }
    wave_sync();
}

// Real calls (once per walk / every 312 draws): their registers and address arithmetic stay out of the step loop.
// Arguments by value, pointers typed by address space so that the bodies keep ds_* / global_* instructions.
__device__ __noinline__ void mt_seed(lds_u32 *mt, glb_u32 *raw, int lane, uint32_t seed) {
    // mt[i] = 1812433253 * (mt[i-1] ^ (mt[i-1] >> 30)) + i : a serial chain, computed redundantly
    // by every lane (wave-uniform); lane i&63 keeps word i and the wave stores 64 words at a time.
    uint32_t x = seed, mine = 0;
    for (int i = 0; i < 624; i++) {
        if ((i & 63) == lane) mine = x;
        if ((i & 63) == 63 || i == 623) {
            const int base = i & ~63;
            if (base + lane <= i) mt[base + lane] = mine;
        }
        x = 1812433253u * (x ^ (x >> 30)) + (uint32_t) (i + 1);
    }
    wave_sync();    // LDS is ordered per wave; this only keeps the compiler from caching or moving the accesses
    mt_twist_lds(mt, raw, lane);    // std::mt19937 twists before its first output
}

__device__ __noinline__ void mt_twist(lds_u32 *mt, glb_u32 *raw, int lane) {
    for (int i = lane; i < 624; i += 64) mt[i] = raw[i];
    wave_sync();
    mt_twist_lds(mt, raw, lane);
}

// the generator of a walk: the LDS block of ready doubles, the raw state in HBM, the index of the next double
struct Rng {
    uint32_t *mt;
    uint32_t *raw;
    int idx;
};

__device__ __forceinline__ double mt_canonical(Rng &g, int lane) {
    if (__builtin_expect(g.idx >= 312, 0)) {
        mt_twist((lds_u32 *) g.mt, (glb_u32 *) g.raw, lane);
        g.idx = 0;
    }
    const uint2 w = *reinterpret_cast<const uint2 *>(g.mt + 2 * g.idx);
    g.idx++;
    return __hiloint2double((int) w.y, (int) w.x);
}

// ---------------------------------------------------------------------------------------------

// (the ids that reach it are neighbour ids of the graph, validated when a neighbour is CHOSEN (row_record) and by the graph
// tests; round 3 could not put an index check here -- see visited_test for what that was)
__device__ __forceinline__ bool bm_test(const uint32_t *bm, uint32_t n) {
    // (the map is GLOBAL memory and is read as such: see visited_test)
    return (((__attribute__((address_space(1))) const uint32_t *) bm)[n >> 5] >> (n & 31)) & 1u;
}

// Visited set of a Monte-Carlo walk.  Node ids are walk-order indices (graph build: nearest anchor, BFS level), so
// the nodes a walk meets lie in a band of ids around its start anchor.  The band [lo, lo + WIN_BITS) lives in LDS
// (one bit per node, cleared at walk start): a probe of a neighbour inside it is an LDS read, no global round trip.
// Ids outside the band use the wave's N-bit map in HBM, exactly as the deterministic kernel does for every id; the
// split is by id, so the set is exact whatever the numbering looks like.
constexpr uint32_t WIN_BITS = 1u << 14;                 // 2 KB of LDS per wave
constexpr uint32_t WIN_WORDS = WIN_BITS / 32;
constexpr uint32_t AFTER_SLOTS = 8;                     // closing sums of the first 8 chunks (2048 entries) of a long list
struct Visited {
    uint32_t *win;      // 2 KB-aligned in LDS
    uint32_t lo;
    uint32_t *bm;
};

// LDS byte address of the band (the low 32 bits of a generic pointer into LDS are its LDS offset)
__device__ __forceinline__ uint32_t lds_addr(const uint32_t *p) {
    return (uint32_t) (uintptr_t) (__attribute__((address_space(3))) const uint32_t *) p;
}

// The slow-path probe of one id (a lane whose entry holds an anchor, an id outside the band, the serial and backtracking
// scans).  Band (LDS) and map (global memory) are read THROUGH THEIR OWN ADDRESS SPACES.  Round 3's form read both through
// generic pointers; the optimiser turned "band or map" into ONE flat load through a pointer picked from a two-entry table it
// kept in scratch, and every attempt to put a third outcome into the far branch (round 3: an index check; round 4: a
// filter in front of the map) came out of that transformation as a kernel whose walks differed from run to run -- although
// the far branch was never reached (counted: 0 calls on the graph that showed it,
// tests/test_gpu_parity.py::test_walks_across_lane_widths).  With typed address spaces there is no pointer to choose
// between: a ds_read under one predicate, a global_load under another, no scratch table, 250 flat loads in the file
// instead of 984 (profiles/experiments/r04_far_filter_bisect.md).
__device__ __forceinline__ bool visited_test(const Visited &v, uint32_t n) {      // slow paths only
    const uint32_t d = n - v.lo;
    if (d < WIN_BITS) {
        const uint32_t w = ((__attribute__((address_space(3))) const uint32_t *) (uintptr_t) lds_addr(v.win))[d >> 5];
        return ((w >> (d & 31u)) & 1u) != 0;
    }
    return bm_test(v.bm, n);
}

__device__ __forceinline__ int64_t edge_delta(const uint2 *qt_tab, uint32_t e) {
    if (!TLM_IN(CHK_QT, e, TLM_N_ENT)) return 0;
    const uint2 qt = qt_tab[e];
    return (int64_t) qt.x - (int64_t) qt.y;                    // Path.cpp:14
}

// ---------------------------------------------------------------------------------------------
// Wave-level primitives on DPP (data-parallel primitives: lane permutes inside the VALU, no LDS
// round trip).  gfx9 controls: row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143.

__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int l) {
    return (uint32_t) __builtin_amdgcn_readlane((int) v, l);
}

__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                            __builtin_amdgcn_readlane(__double2loint(v), l));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {   // lanes without a source read 0.0
    if (ROW_MASK == 0xf) {      // every row written: bound_ctrl supplies the zeros, no `old` register to initialise
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
        return __hiloint2double(hi, lo);
    }
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

// inclusive prefix sum over the 64 lanes: 4 shifts inside each row of 16, then the row totals
__device__ __forceinline__ double wave_incl_scan_f64(double v) {
    v += dpp_f64<0x111, 0xf>(v);
    v += dpp_f64<0x112, 0xf>(v);
    v += dpp_f64<0x114, 0xf>(v);
    v += dpp_f64<0x118, 0xf>(v);
    v += dpp_f64<0x142, 0xa>(v);     // lane 15 of rows 0 / 2 -> rows 1 / 3
    v += dpp_f64<0x143, 0xc>(v);     // lane 31 -> rows 2 and 3
    return v;
}

struct WaveScratch {
    uint32_t *pn;     // path: nodes
    uint32_t *pe;     // path: global CSR index of each edge
    uint32_t *lg;     // every node whose visited bit was set (to clear the bitmap afterwards)
    uint32_t *bm;     // N-bit visited set, all zero between walks
    uint32_t *rng;    // raw generator state (Monte-Carlo)
};

// End of a walk: publish the unit's outcome, copy an accepted path into the arena, clear the bitmap.
__device__ __forceinline__ void finish_unit(const WaveScratch &ws, int lane, uint32_t ui, uint8_t status, uint32_t m,
                                   uint32_t nlog, int64_t len_final, uint32_t steps, uint64_t evals, uint32_t draws) {
    ColdArgs *ca = cold_args();
    const uint32_t n_out = m + 1;
    uint64_t off = 0;
    if (status == ST_ACCEPTED) {
        unsigned long long o = 0;
        if (lane == 0) o = atomicAdd(reinterpret_cast<unsigned long long *>(ca->cursor + 2), (unsigned long long) n_out);
        off = __shfl(o, 0);
        if (off + n_out > ca->ar_cap) {
            status = ST_ARENA_OVF;
        } else {
            uint2 *arena = ca->arena;
            const uint32_t *old_of_new = ca->old_of_new;
            const uint2 *rows = ca->rows;
            for (uint32_t i = lane; i < n_out; i += 64) {
                if (!TLM_IN(CHK_PATH, i, ca->cap)) continue;
                const uint32_t node = ws.pn[i];
                if (!TLM_IN(CHK_NODE_TABLE, node, TLM_N_NODES) || !TLM_IN(CHK_ARENA, off + i, ca->ar_cap)) continue;
                arena[off + i] = make_uint2(old_of_new[node],                         // reference index
                                            (i < m) ? (ws.pe[i] - rows[node].x) : 0xFFFFFFFFu);
            }
        }
    }
    if (lane == 0) {
        UnitOut u;
        u.evals = evals;
        u.off = off;
        u.len = len_final;
        u.nnodes = n_out;
        u.steps = steps;
        u.draws = draws;
        u.status = status;
        if (TLM_IN(CHK_UNIT, ui, ca->n_work_all)) ca->units[ui] = u;
    }
    // the wave's HBM map goes back to all-zero: single marks by node id, flushed bands (band_slide) by their first word
    const uint32_t n_words = ca->bm_words - WALK_RNG_WORDS;
    for (uint32_t base = 0; base < nlog; base += 64) {
        const bool valid = base + lane < nlog && TLM_IN(CHK_LOG, base + lane, ca->cap);
        const uint32_t e = valid ? ws.lg[base + lane] : 0u;
        if (valid && !(e >> 31) && TLM_IN(CHK_MAP, e >> 5, n_words)) ws.bm[e >> 5] = 0u;
        uint64_t bands = __ballot(valid && (e >> 31));
        while (bands) {
            const int j = __ffsll((unsigned long long) bands) - 1;
            bands &= bands - 1;
            const uint32_t first = readlane_u32(e, j) & 0x7FFFFFFFu;
            for (uint32_t k = lane; k < WIN_WORDS; k += 64)
                if (first + k < n_words) ws.bm[first + k] = 0u;
        }
    }
    wave_sync();
}

// ---------------------------------------------------------------------------------------------
// One Monte-Carlo trial: PathManager.cpp:30-160.
//
// The selection scan of one step (:53-115) on a wave, BLOCKED layout: lane l owns K consecutive
// entries K*l .. K*l+K-1 of the list (K/2 aligned 16-byte loads: rows start on 64-byte boundaries and
// are padded to 8 entries), K = 4 for lists up to 256 entries, K = 8 up to 512.  List order is
// lane-major, so the prefix of entry (l, k) is
//     [exclusive wave scan of the lane totals](l) + [running sum inside lane l],
// i.e. ONE 64-lane scan per step, everything else is lane-local and the list is read once.  On a safe
// node (all metrics >= 0, every partial sum exact in a double -- node flag from the graph build) any
// association of the additions gives the reference's bits, and the non-decreasing prefix lets the
// second pass (:107-114) look only at the one lane whose range contains `random`.  Unsafe nodes take
// select_serial, which adds one candidate after the other in list order.
//
// No per-entry validity tests: row padding entries name the row's own node (visited while the walk
// stands on it, so never a candidate; anchor bit clear, metric 0), and lanes past the end of the list
// re-read the list's first entries and have their sum forced to zero.

// x & ~mask in ONE instruction (v_bfi_b32: (mask & 0) | (~mask & x)); left to itself the compiler splits some of these into
// v_xor + v_and around the wait for the list (-0.6 % of the Monte-Carlo launch, round 3)
__device__ __forceinline__ uint32_t and_not(uint32_t x, uint32_t mask) {
    uint32_t r;
    asm("v_bfi_b32 %0, %1, 0, %2" : "=v"(r) : "v"(mask), "v"(x));
    return r;
}

struct Selection {
    int pick;            // position in the list, -1 = none
    uint32_t nbr;        // neighbour at `pick`
    uint32_t anchor;     // its anchor flag (0 / 1: 32-bit on purpose -- a bool carried around in scalar registers is a 64-bit lane mask)
    uint32_t forced;     // taken by the forced-anchor rule (:57-60)
    uint32_t ncand;      // unvisited neighbours seen (0 with pick < 0 = dead end)
    uint32_t evals;      // adjacency entries examined
};

template <bool SQRT>
__device__ __forceinline__ double metric_f64(uint32_t bits) {
    float mf = __uint_as_float(bits);
    if (SQRT) mf = sqrtf(mf);                           // Utils::getMetric, Utils.hpp:47-60
    return (double) mf;
}

struct Chunk {           // one lane's 4 entries of a 256-entry chunk (deterministic kernel, backtracking)
    uint32_t nb[4];      // neighbour | anchor << 31
    uint32_t mb[4];      // fp32 bits of the metric
};

__device__ __forceinline__ Chunk load_chunk(const uint2 *ents, uint32_t row, uint32_t base, uint32_t deg, int lane) {
    // row is a multiple of 8 entries and base + 4*lane of 4: 32-byte aligned, two dwordx4 loads.  Lanes whose four
    // entries all lie past the degree load nothing; inside a lane, entries past the degree are row padding, masked
    // by position and never used.
    Chunk c;
    if (base + 4u * (uint32_t) lane < deg && TLM_IN(CHK_CHUNK, (uint64_t) row + base + 4u * (uint32_t) lane + 3u, TLM_N_ENT)) {
        const uint4 *p = reinterpret_cast<const uint4 *>(ents + row + base + 4u * (uint32_t) lane);
        const uint4 lo = p[0], hi = p[1];
        c.nb[0] = lo.x; c.mb[0] = lo.y; c.nb[1] = lo.z; c.mb[1] = lo.w;
        c.nb[2] = hi.x; c.mb[2] = hi.y; c.nb[3] = hi.z; c.mb[3] = hi.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) c.nb[k] = c.mb[k] = 0u;
    }
    return c;
}

// K consecutive entries per lane starting at list position base + K*lane; lanes past the end of the list (`act`
// false) re-read the chunk's first entries (same cache line as lane 0) -- their sums are forced to zero below.
template <int K>
__device__ __forceinline__ bool lane_load(const uint2 *ents, uint32_t row, uint32_t base, uint32_t deg, int lane,
                                          uint32_t (&nb)[K], uint32_t (&mb)[K]) {
    const uint32_t first = base + (uint32_t) K * (uint32_t) lane;
    const bool act = first < deg;
#ifdef TLM_ABLATE_HALFLINES /* timing-only build (results are wrong): lanes 32..63 re-read the addresses of lanes 0..31, so
                               a step fetches HALF the 128-byte lines of its list with the same instruction stream */
    const uint4 *p = reinterpret_cast<const uint4 *>(ents + row + (act ? base + (uint32_t) K * (uint32_t) (lane & 31) : base));
#else
    const uint4 *p = reinterpret_cast<const uint4 *>(ents + row + (act ? first : base));
#endif
    // (an EMPTY row at the very end of the arrays starts where they end: its inactive lanes read "lane 0's entries" there,
    // K entries of the 256 zero entries of slack behind the last row -- the one legitimate use of that slack)
    if (!TLM_IN(CHK_LIST8, (uint64_t) row + (act ? first : base) + (uint32_t) K - 1u, TLM_N_ENT + (deg == 0u ? (uint32_t) K : 0u))) {
#pragma unroll
        for (int k = 0; k < K; k++) nb[k] = mb[k] = 0u;
        return false;
    }
#pragma unroll
    for (int j = 0; j < K / 2; j++) {
        const uint4 v = p[j];      // (non-temporal loads were tried: 120 -> 179 ms, the lists do get reused in L2)
        nb[2 * j] = v.x; mb[2 * j] = v.y; nb[2 * j + 1] = v.z; mb[2 * j + 1] = v.w;
    }
    return act;
}

// The same K entries of a COMPACT row (row flag bit 29): 4-byte metric + 2-byte neighbour id relative to the row's own
// node -- 6 bytes per eval instead of 8, 11 instead of 15 cache lines for a list of 236 entries.  m[k] = metric bits of
// entry K*lane + k, w[j] = the ids of entries K*lane + 2j (low half) and 2j + 1.  Lanes past the end of the list read
// entries 0.. of the arrays: a block of 256 zero entries in front of the first row (metric 0, relative id 32768 = the
// row's own node, which is visited while the walk stands on it) -- they need no masking afterwards.
template <int K>
__device__ __forceinline__ void lane_load_cw(const uint32_t *met, const uint16_t *d16, uint32_t row, uint32_t deg, int lane,
                                             uint32_t (&mb)[K], uint32_t (&w)[K / 2]) {
    const uint32_t first = (uint32_t) K * (uint32_t) lane;
    const uint32_t off = first < deg ? row + first : 0u;     // rows start at multiples of 8 entries: 32-byte / 16-byte aligned
    struct alignas(K == 6 ? 8 : 16) Met { uint32_t v[K]; };
    struct alignas(K == 6 ? 4 : K == 12 ? 8 : 2 * K) Ids { uint32_t v[K / 2]; };
    if (!TLM_IN(CHK_LISTC, (uint64_t) off + (uint32_t) K - 1u, TLM_N_ENT)) {
#pragma unroll
        for (int k = 0; k < K; k++) mb[k] = 0u;
#pragma unroll
        for (int j = 0; j < K / 2; j++) w[j] = 0x80008000u;
        return;
    }
    const Met m = *reinterpret_cast<const Met *>(met + off);
    const Ids ww = *reinterpret_cast<const Ids *>(d16 + off);
#pragma unroll
    for (int k = 0; k < K; k++) mb[k] = m.v[k];
#pragma unroll
    for (int j = 0; j < K / 2; j++) w[j] = ww.v[j];
}

// ids relative to the visited band (`d` = id - band base), which is what the probe wants: id = d + lo
template <int K>
__device__ __forceinline__ void unpack_rel(const uint32_t (&w)[K / 2], uint32_t rel_base, uint32_t (&d)[K]) {
#pragma unroll
    for (int j = 0; j < K / 2; j++) {
        d[2 * j] = (w[j] & 0xFFFFu) + rel_base;         // (v_add_u32 with an SDWA word select)
        d[2 * j + 1] = (w[j] >> 16) + rel_base;
    }
}

template <int K>
__device__ __forceinline__ bool lane_load_c(const uint32_t *met, const uint16_t *d16, uint32_t row, uint32_t deg, int lane,
                                            uint32_t rel_base, uint32_t (&d)[K], uint32_t (&mb)[K]) {
    uint32_t w[K / 2];
    lane_load_cw<K>(met, d16, row, deg, lane, mb, w);
    unpack_rel<K>(w, rel_base, d);
    return (uint32_t) K * (uint32_t) lane < deg;
}

// visited masks for band-relative ids (compact rows hold no anchors: every id is a plain node id)
template <int K>
__device__ __forceinline__ void lane_probe_rel(const Visited &vs, const uint32_t (&d)[K], uint32_t (&vis)[K]) {
    uint32_t any = 0u;
#ifdef TLM_ABLATE_NOBM
#pragma unroll
    for (int k = 0; k < K; k++) vis[k] = 0u;
    (void) any;
#else
    const uint32_t wb = lds_addr(vs.win);
#pragma unroll
    for (int k = 0; k < K; k++) {
        any |= d[k];
        static_assert(WIN_WORDS == 512u, "the bit-field width below is log2(WIN_WORDS)");
        uint32_t addr;
        asm("v_bfe_u32 %0, %1, 5, 9\n\tv_lshl_or_b32 %0, %0, 2, %2" : "=v"(addr) : "v"(d[k]), "s"(wb));
        vis[k] = *(__attribute__((address_space(3))) const uint32_t *) (uintptr_t) addr;
    }
#pragma unroll
    for (int k = 0; k < K; k++) vis[k] = (uint32_t) __builtin_amdgcn_sbfe((int) vis[k], d[k], 1u);   // lo is a multiple of 32
    if (__builtin_expect(__ballot(any >= WIN_BITS) != 0, 0)) {    // rare: a neighbour outside the band
#ifdef TLM_TIMING
        if ((threadIdx.x & 63) == 0) atomicAdd(&tlm_timing[6], 1ull);
#endif
#pragma unroll
        for (int k = 0; k < K; k++)
            if (d[k] >= WIN_BITS) vis[k] = bm_test(vs.bm, d[k] + vs.lo) ? ~0u : 0u;
    }
#endif
}

// forced anchor edge: the first entry, in list order, whose neighbour is an anchor other than `start` (:57-60).
// Returns its position relative to the chunk (lane*K+k) or -1.  Padding entries carry no anchor bit; an inactive
// lane repeats lane 0's entries, and lane 0 wins the find-first.
template <int K>
__device__ __forceinline__ int forced_in_lanes(const uint32_t (&nb)[K], uint32_t start, uint32_t &nbr_out) {
    int best = -1;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const uint64_t fm = __ballot((nb[k] >> 31) && (nb[k] & 0x7FFFFFFFu) != start);
        if (fm) {
            const int fl = __ffsll((unsigned long long) fm) - 1;
            const int pos = fl * K + k;
            if (best < 0 || pos < best) {
                best = pos;
                nbr_out = readlane_u32(nb[k], fl) & 0x7FFFFFFFu;
            }
        }
    }
    return best;
}

// visited masks of a lane's K entries (all ones = visited): K LDS reads in flight at once, no branches.  The anchor
// bit is left in the id on purpose: such an entry lands outside the band and takes the exact path below, as does
// every id that really lies outside the band.
template <int K>
__device__ __forceinline__ void lane_probe(const Visited &vs, const uint32_t (&nb)[K], uint32_t (&vis)[K]) {
    uint32_t any = 0u;
#ifdef TLM_ABLATE_NOBM      /* timing-only build: no visited probes (results are wrong) */
#pragma unroll
    for (int k = 0; k < K; k++) vis[k] = 0u;
    (void) any;
#else
    const uint32_t wb = lds_addr(vs.win);                             // multiple of 2048: `|` instead of `+` (v_and_or_b32)
#pragma unroll
    for (int k = 0; k < K; k++) {
        const uint32_t d = nb[k] - vs.lo;
        any |= d;
        // word (d >> 5) mod 512 of the band: always inside it, whatever d is
        // two instructions (the compiler's own form, shift + mask + or, takes three: a VOP3 cannot carry the literal)
        static_assert(WIN_WORDS == 512u, "the bit-field width below is log2(WIN_WORDS)");
        uint32_t addr;
        asm("v_bfe_u32 %0, %1, 5, 9\n\tv_lshl_or_b32 %0, %0, 2, %2" : "=v"(addr) : "v"(d), "s"(wb));
        vis[k] = *(__attribute__((address_space(3))) const uint32_t *) (uintptr_t) addr;
    }
    // bit (id - lo) & 31 == id & 31 (lo is a multiple of 32; v_bfe_i32 reads 5 bits of the offset): 0 or all ones
#pragma unroll
    for (int k = 0; k < K; k++) vis[k] = (uint32_t) __builtin_amdgcn_sbfe((int) vis[k], nb[k], 1u);
    if (__builtin_expect(__ballot(any >= WIN_BITS) != 0, 0)) {    // rare: an anchor neighbour, or a neighbour outside the band
#ifdef TLM_TIMING
        if ((threadIdx.x & 63) == 0) atomicAdd(&tlm_timing[7], 1ull);
#endif
#pragma unroll
        for (int k = 0; k < K; k++)
            if (nb[k] - vs.lo >= WIN_BITS) vis[k] = visited_test(vs, nb[k] & 0x7FFFFFFFu) ? ~0u : 0u;
    }
#endif
}

// Selection among the K*64 entries held in registers.  `mb` arrives masked (0 where the entry is not a candidate),
// `va` is the AND of the lane's visited masks (all ones = no candidate in this lane).
// REL: `nb` holds band-relative ids of a compact row (no anchor bits): neighbour = nb + vs.lo.
template <int K, bool SQRT, bool REL = false>
__device__ __forceinline__ void select_in_regs(Selection &s, const Visited &vs, Rng &rng, int lane,
                                               uint32_t base, bool act, const uint32_t (&nb)[K], const uint32_t (&mb)[K],
                                               uint32_t va, uint32_t &draws) {
    double ls = metric_f64<SQRT>(mb[0]);
#pragma unroll
    for (int k = 1; k < K; k++) ls += metric_f64<SQRT>(mb[k]);                     // :64
    if (!act) ls = 0.0;
#ifdef TLM_ABLATE_NOSCAN    /* timing-only build: no prefix scan (results are wrong) */
    const double incl = ls * (double) (lane + 1);
#else
    const double incl = wave_incl_scan_f64(ls);
#endif
    const double total = readlane_f64(incl, 63);
    const bool has = act && va != ~0u;
    if (__builtin_expect(total == 0.0, 0)) {                                       // dead end, or candidates of metric 0
        if (!__ballot(has)) return;                                                // :72
    }
    s.ncand = 1;
    const double random = mt_canonical(rng, lane) * total;                // :104
    draws++;
    int hl, myk = K - 1;
    uint32_t mynb = nb[K - 1];
    if (__builtin_expect(random == 0.0, 0)) {
        // `sum >= 0` holds at the first candidate of the list (:110); rare enough to probe again
        hl = __ffsll((unsigned long long) __ballot(has)) - 1;
        uint32_t vis[K], nb2[K];
#pragma unroll
        for (int k = 0; k < K; k++) {       // opaque copy: nothing of the first probe is kept alive for this path
            nb2[k] = nb[k];
            asm volatile("" : "+v"(nb2[k]));
        }
        if (REL) lane_probe_rel<K>(vs, nb2, vis);
        else lane_probe<K>(vs, nb2, vis);
#pragma unroll
        for (int k = K - 2; k >= 0; k--)
            if (vis[k] == 0u) {
                myk = k;
                mynb = nb[k];
            }
    } else {
        // the first lane whose inclusive prefix reaches `random` holds the hit; inside it the first entry whose
        // running prefix reaches it.  Entries that are not candidates add +0 and so never are the first to reach it.
        const uint64_t hm = __ballot(incl >= random);
        if (!hm) return;                               // not reachable for finite sums (u < 1)
        hl = __ffsll((unsigned long long) hm) - 1;
        double pk[K - 1];
        double p = incl - ls;                           // exclusive prefix of the lane (exact)
#pragma unroll
        for (int k = 0; k < K - 1; k++) {
            p += metric_f64<SQRT>(mb[k]);               // :109
            pk[k] = p;
        }
#pragma unroll
        for (int k = K - 2; k >= 0; k--)
            if (pk[k] >= random) {                      // :110, first k wins
                myk = k;
                mynb = nb[k];
            }
    }
    const int k = __builtin_amdgcn_readlane(myk, hl);
    const uint32_t e = readlane_u32(mynb, hl);
    s.pick = (int) base + hl * K + k;
    s.nbr = REL ? e + vs.lo : e & 0x7FFFFFFFu;
    s.anchor = REL ? 0u : e >> 31;
}

// A list of at most 64*K entries: one load, one visited probe, ONE scan (it yields the total and every lane's
// prefix), no second pass over memory.
template <int K, bool SQRT>
__device__ __forceinline__ Selection select_short(const HotArgs &a, const Visited &vs, Rng &rng, int lane,
                                                  uint32_t row, uint32_t deg, uint32_t start, bool anchor_nbr,
                                                  uint32_t &draws, StepClock &clk) {
    Selection s{-1, 0u, 0u, 0u, 0u, deg};
    uint32_t nb[K], mb[K], vis[K];
    const bool act = lane_load<K>(a.ents, row, 0u, deg, lane, nb, mb);
    clk.mark(CLK_LIST_WAIT, true);
    if (__builtin_expect(anchor_nbr, 0)) {
        uint32_t fn = 0;
        const int fpos = forced_in_lanes<K>(nb, start, fn);
        if (fpos >= 0) {
            s.pick = fpos;
            s.nbr = fn;
            s.anchor = s.forced = 1u;
            s.evals = (uint32_t) fpos + 1u;
            return s;
        }
    }
    lane_probe<K>(vs, nb, vis);
    uint32_t va = ~0u;
#pragma unroll
    for (int k = 0; k < K; k++) {
        mb[k] &= ~vis[k];
        va &= vis[k];
    }
    select_in_regs<K, SQRT>(s, vs, rng, lane, 0u, act, nb, mb, va, draws);
    return s;
}

// The same on a compact row (no anchor in the list, so no forced-anchor test; `node` = the row's own node), with what
// the compact form allows left out of the common path (the kernel is bound by vector-instruction issue):
//   * no lane is masked: lanes past the list hold zero entries (lane_load_cw), padding names the row's own node;
//   * the running sums inside a lane are the partial sums of the lane's own total, kept from the summation: the hit is
//     the first entry whose partial sum reaches `random - (exclusive prefix of the lane)`.  On an order-safe row (the only
//     rows that come here) that difference is exact: every prefix is a multiple of the row's quantum q with total <
//     2^52 q, so ulp(random) < q and random - prefix has no bit below ulp(random) -- the comparison is the reference's
//     `sum >= random` (:110), bit for bit;
//   * "any candidate at all?" is only asked when the total is zero, by probing again;
//   * the chosen neighbour is read from the packed id words (K / 2 readlanes and scalar selects, no per-lane chain).
template <int K, bool SQRT>
__device__ __forceinline__ Selection select_short_c(const HotArgs &a, const Visited &vs, Rng &rng, int lane, uint32_t row,
                                                    uint32_t deg, uint32_t node, uint32_t &draws, StepClock &clk) {
    Selection s{-1, 0u, 0u, 0u, 0u, deg};
    uint32_t w[K / 2], mb[K];
    lane_load_cw<K>(a.met, a.d16, row, deg, lane, mb, w);
    clk.mark(CLK_LIST_WAIT, true);
    const uint32_t rel_base = node - vs.lo - 32768u;
    {
        uint32_t d[K], vis[K];
        unpack_rel<K>(w, rel_base, d);
        lane_probe_rel<K>(vs, d, vis);
#pragma unroll
        for (int k = 0; k < K; k++) mb[k] = and_not(mb[k], vis[k]);
    }
    // ps[k] = candidates' metrics of the lane's entries 0..k (:64); kept across the scan for the shorter lists, summed
    // again afterwards for 8 and 12 entries per lane (registers)
    constexpr bool KEEP = K <= 6;
    double ps[K];
    ps[0] = metric_f64<SQRT>(mb[0]);
#pragma unroll
    for (int k = 1; k < K; k++) ps[k] = ps[k - 1] + metric_f64<SQRT>(mb[k]);
    const double ls = ps[K - 1];
#ifdef TLM_ABLATE_NOSCAN
    const double incl = ls * (double) (lane + 1);
#else
    const double incl = wave_incl_scan_f64(ls);
#endif
    const double total = readlane_f64(incl, 63);
    // the unvisited entries of a lane, asked for on the two rare paths only
    auto unvisited = [&](uint32_t (&vis)[K]) {
        uint32_t d[K];
        unpack_rel<K>(w, rel_base, d);
#pragma unroll
        for (int k = 0; k < K; k++) asm volatile("" : "+v"(d[k]));     // opaque: nothing of the first probe is kept alive
        lane_probe_rel<K>(vs, d, vis);
        uint32_t va = ~0u;
#pragma unroll
        for (int k = 0; k < K; k++) va &= vis[k];
        return va != ~0u;
    };
    if (__builtin_expect(total == 0.0, 0)) {            // dead end, or candidates of metric 0
        uint32_t vis[K];
        if (!__ballot(unvisited(vis))) return s;        // :72
    }
    s.ncand = 1;
    const double random = mt_canonical(rng, lane) * total;     // :104
    draws++;
    int hl, myk = K - 1;
    if (__builtin_expect(random == 0.0, 0)) {
        // `sum >= 0` holds at the first candidate of the list (:110)
        uint32_t vis[K];
        hl = __ffsll((unsigned long long) __ballot(unvisited(vis))) - 1;
#pragma unroll
        for (int k = K - 2; k >= 0; k--)
            if (vis[k] == 0u) myk = k;
    } else {
        // the first lane whose inclusive prefix reaches `random` holds the hit
        const uint64_t hm = __ballot(incl >= random);
        if (!hm) return s;                              // not reachable for finite sums (u < 1)
        hl = __ffsll((unsigned long long) hm) - 1;
        const double rr = random - (incl - ls);
        if (!KEEP) {
#pragma unroll
            for (int k = 0; k < K; k++) asm volatile("" : "+v"(mb[k]));        // (a second summation, not kept sums)
            ps[0] = metric_f64<SQRT>(mb[0]);
#pragma unroll
            for (int k = 1; k < K - 1; k++) ps[k] = ps[k - 1] + metric_f64<SQRT>(mb[k]);
        }
#pragma unroll
        for (int k = K - 2; k >= 0; k--)
            if (ps[k] >= rr) myk = k;                   // :110, first k wins; entries that are no candidates add +0
    }
    const uint32_t k = (uint32_t) __builtin_amdgcn_readlane(myk, hl);
    uint32_t raw = readlane_u32(w[0], hl);
#pragma unroll
    for (int j = 1; j < K / 2; j++) {
        const uint32_t rj = readlane_u32(w[j], hl);
        if ((k >> 1) == (uint32_t) j) raw = rj;
    }
    const uint32_t rel = (k & 1u) ? raw >> 16 : raw & 0xFFFFu;
    s.pick = hl * K + (int) k;
    s.nbr = node + rel - 32768u;
    return s;
}

template <bool SQRT>
__device__ __forceinline__ Selection select_serial(const HotArgs &a, const Visited &vs, Rng &rng, int lane, uint32_t row,
                                                   uint32_t deg, uint32_t start, uint32_t &draws, bool have_u, double u_given);

// Lists longer than 512 entries, chunks of 512.  Pass 1 (:53-67): total over the unvisited neighbours; the running
// sum after each of the first AFTER_SLOTS chunks is parked in LDS: the prefix is non-decreasing, so pass 2
// (:107-114) only has to look into the one chunk where it crosses `random`.
template <bool SQRT>
__device__ __forceinline__ Selection select_long(const HotArgs &a, const Visited &vs, Rng &rng, double *after,
                                              int lane, uint32_t row, uint32_t deg, uint32_t start, bool anchor_nbr,
                                              uint32_t &draws) {
    constexpr int K = 8;
    constexpr uint32_t C = 64u * K;
    Selection s{-1, 0u, 0u, 0u, 0u, deg};
    double total = 0.0;
    uint32_t ci = 0;
    bool any_cand = false;
    for (uint32_t base = 0; base < deg; base += C, ci++) {
        uint32_t nb[K], mb[K], vis[K];
        const bool act = lane_load<K>(a.ents, row, base, deg, lane, nb, mb);
        if (anchor_nbr) {
            uint32_t fn = 0;
            const int fpos = forced_in_lanes<K>(nb, start, fn);
            if (fpos >= 0) {
                s.pick = (int) base + fpos;
                s.nbr = fn;
                s.anchor = s.forced = 1u;
                s.evals = (uint32_t) s.pick + 1u;
                return s;
            }
        }
        lane_probe<K>(vs, nb, vis);
        double ls = 0.0;
        uint32_t va = ~0u;
#pragma unroll
        for (int k = 0; k < K; k++) {
            ls += metric_f64<SQRT>(mb[k] & ~vis[k]);
            va &= vis[k];
        }
        if (!act) ls = 0.0;
        if (__ballot(act && va != ~0u)) any_cand = true;
        total += readlane_f64(wave_incl_scan_f64(ls), 63);
        if (ci < AFTER_SLOTS && lane == 0) after[ci] = total;
    }
    if (!any_cand) return s;                                                        // :72 dead end
    s.ncand = 1;
    wave_sync();                                                                    // `after` is read by every lane below
    const double u = mt_canonical(rng, lane);
    draws++;
    const double random = u * total;                                                // :104
    if (random == 0.0) return select_serial<SQRT>(a, vs, rng, lane, row, deg, start, draws, true, u);
    double carry = 0.0;
    ci = 0;
    for (uint32_t base = 0; base < deg; base += C, ci++) {
        if (ci < AFTER_SLOTS) {
            const double closing = after[ci];
            if (closing < random) {          // the hit is further down the list
                carry = closing;
                continue;
            }
        }
        uint32_t nb[K], mb[K], vis[K];
        const bool act = lane_load<K>(a.ents, row, base, deg, lane, nb, mb);
        lane_probe<K>(vs, nb, vis);
        double ls = 0.0;
#pragma unroll
        for (int k = 0; k < K; k++) {
            mb[k] &= ~vis[k];
            ls += metric_f64<SQRT>(mb[k]);
        }
        if (!act) ls = 0.0;
        const double incl = wave_incl_scan_f64(ls) + carry;
        carry = readlane_f64(incl, 63);
        const uint64_t hm = __ballot(incl >= random);
        if (!hm) continue;
        const int hl = __ffsll((unsigned long long) hm) - 1;
        int myk = K - 1;
        uint32_t mynb = nb[K - 1];
        double pk[K - 1];
        double p = incl - ls;
#pragma unroll
        for (int k = 0; k < K - 1; k++) {
            p += metric_f64<SQRT>(mb[k]);
            pk[k] = p;
        }
#pragma unroll
        for (int k = K - 2; k >= 0; k--)
            if (pk[k] >= random) {
                myk = k;
                mynb = nb[k];
            }
        const int k = __builtin_amdgcn_readlane(myk, hl);
        const uint32_t e = readlane_u32(mynb, hl);
        s.pick = (int) base + hl * K + k;
        s.nbr = e & 0x7FFFFFFFu;
        s.anchor = e >> 31;
        return s;
    }
    return s;
}

// Unsafe node (a metric is negative / not finite, or the sums are not exact in a double): the additions are made
// one candidate after the other in list order, as PathManager.cpp:64,109 makes them.  Rare; never on real overlaps.
// `have_u`: the step's draw was already made by the caller (select_long, when u * total came out as 0).
template <bool SQRT>
__device__ __forceinline__ Selection select_serial(const HotArgs &a, const Visited &vs, Rng &rng, int lane,
                                                uint32_t row, uint32_t deg, uint32_t start, uint32_t &draws, bool have_u,
                                                double u_given) {
    Selection s{-1, 0u, 0u, 0u, 0u, deg};
    double total = 0.0;
    for (int pass = 0; pass < 2; pass++) {
        double run = 0.0;
        double random = 0.0;
        if (pass == 1) {
            if (s.ncand == 0) return s;
            if (have_u) {
                random = u_given * total;
            } else {
                random = mt_canonical(rng, lane) * total;                        // :104
                draws++;
            }
        }
        for (uint32_t base = 0; base < deg; base += 64) {
            const uint32_t i = base + lane;
            const bool valid = i < deg && TLM_IN(CHK_SERIAL, (uint64_t) row + i, TLM_N_ENT);
            const uint2 ent = valid ? a.ents[row + i] : make_uint2(0u, 0u);
            const uint32_t nbr = ent.x & 0x7FFFFFFFu;
            if (pass == 0) {
                const uint64_t fm = __ballot(valid && (ent.x >> 31) && nbr != start);     // :57
                if (fm) {
                    const int fl = __ffsll((unsigned long long) fm) - 1;
                    s.pick = (int) base + fl;
                    s.nbr = readlane_u32(nbr, fl);
                    s.anchor = s.forced = 1u;
                    s.evals = (uint32_t) s.pick + 1u;
                    return s;
                }
            }
            const bool cand = valid && !visited_test(vs, valid ? nbr : vs.lo);       // :63
            uint64_t cm = __ballot(cand);
            if (pass == 0) s.ncand += (uint32_t) __popcll((unsigned long long) cm);
            const double mv = cand ? metric_f64<SQRT>(ent.y) : 0.0;
            while (cm) {
                const int j = __ffsll((unsigned long long) cm) - 1;
                cm &= cm - 1;
                run += readlane_f64(mv, j);                                           // :64 / :109
                if (pass == 1 && run >= random) {                                     // :110
                    s.pick = (int) base + j;
                    s.nbr = readlane_u32(nbr, j);
                    s.anchor = readlane_u32(ent.x, j) >> 31;
                    return s;
                }
            }
        }
        total = run;
    }
    return s;
}

// The LDS band FOLLOWS the walk.  Ids are walk-order indices: (nearest anchor, BFS level), so a walk is among nearby ids
// while it stays in the neighbourhood of one anchor -- and jumps to another id range when it crosses into the next
// anchor's neighbourhood, which every accepted walk does once (anchors are numbered in node order, not along the genome).
// A band fixed at the start anchor left a third of all steps probing the HBM map (measured).  So: when the
// current node is further than BAND_SLACK from the band's centre, the band's words are written to the wave's HBM map
// (the flushed range is logged, the map is cleared from the log when the walk ends) and the band is re-loaded around the
// current node.  The set stays exact at every moment: band ids in LDS, all others in HBM.
constexpr uint32_t BAND_SLACK = WIN_BITS / 2 - 1024;

__device__ __forceinline__ uint32_t band_base_for(uint32_t n) { return n > WIN_BITS / 2 ? (n - WIN_BITS / 2) & ~31u : 0u; }

__device__ __noinline__ void band_slide(lds_u32 *win, glb_u32 *bm, uint32_t n_words, uint32_t old_lo, uint32_t new_lo, int lane) {
    const uint32_t ow = old_lo >> 5, nw = new_lo >> 5;
    // the band is the truth for its ids (it was loaded from these very words, or both were zero): written back whole,
    // so that a flag the deterministic walk has CLEARED in the band is cleared in the map too
    for (uint32_t i = lane; i < WIN_WORDS; i += 64)
        if (ow + i < n_words) bm[ow + i] = win[i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the two ranges may overlap: stores first, then the loads
    for (uint32_t i = lane; i < WIN_WORDS; i += 64) win[i] = nw + i < n_words ? bm[nw + i] : 0u;
    wave_sync();
}

// The path under construction (edge i, node i + 1) goes to the wave's HBM scratch -- but not one store per step: on
// gfx9 loads and stores share one outstanding-operation counter and return out of order with respect to each other,
// so a wave with a store in flight has to wait for ALL its memory operations before it may touch a loaded value.
// The last PATH_RING entries live in LDS and are written out 32 at a time, by 64 lanes, in two store instructions.
constexpr uint32_t PATH_RING = 32;
struct PathRing {
    uint32_t *ring;     // LDS: [0, 32) edge indices, [32, 64) nodes
    uint32_t base;      // index of the first edge held in the ring
    uint32_t cnt;       // edges held
#ifdef TLM_CHECKED
    uint32_t cap;       // entries of the wave's path arrays
#endif
};

__device__ __forceinline__ void ring_flush(PathRing &pr, const WaveScratch &ws, int lane) {
    if (pr.cnt != 0u) {
        const uint32_t j = (uint32_t) lane & (PATH_RING - 1u);
        if (j < pr.cnt && TLM_IN(CHK_PATH, pr.base + j + 1u, pr.cap)) {
            const uint32_t v = pr.ring[lane];
            if (lane < (int) PATH_RING) ws.pe[pr.base + j] = v;
            else ws.pn[pr.base + j + 1u] = v;
        }
        pr.base += pr.cnt;
        pr.cnt = 0u;
    }
    wave_sync();
}

// before the scratch is READ (backtracking, the end of the walk): everything out, and arrived
__device__ __forceinline__ void ring_sync(PathRing &pr, const WaveScratch &ws, int lane) {
    ring_flush(pr, ws, lane);
    // the wave's own stores, then its own loads: in order through the same L1 / L2 once the stores are acknowledged
    // (no cache maintenance: an agent-scope fence here invalidates the die's L2 under every other walk)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ACCOUNT builds of the walk kernels (never the timed ones; bench.py runs one launch of them beside the timed steps) add up
// the bytes a step's loads ASK the layout for: the list as it is read -- ceil(deg / K) * K entries of 6 bytes (compact row)
// or 8, K = the entries a lane owns for this list length; a chunked long list: all of it once plus one chunk of 512 again
// -- plus the 8-byte row record and the 8-byte {q_start, t_start} of the chosen edge.  Summed per launch in cursor[10..11].
__device__ __forceinline__ uint32_t step_layout_bytes(uint32_t deg, bool compact) {
    if (deg > 768u || (!compact && deg > 512u)) return ((deg + 7u) & ~7u) * 8u + 512u * 8u + 16u;
    const uint32_t k = deg <= 256u ? 4u : deg <= 384u ? 6u : deg <= 512u ? 8u : 12u;
    return ((deg + k - 1u) / k) * k * (compact ? 6u : 8u) + 16u;
}

// ... and which form of the selection every step took (ACCOUNT builds; cursor[12..31] = 10 u64 counters):
// 0-3 compact row with 4 / 6 / 8 / 12 entries per lane, 4-6 8-byte row with 4 / 6 / 8, 7 8-byte row with 12 (deterministic
// kernel only), 8 chunked long list, 9 serial (order-unsafe row)
constexpr int STEP_CLASSES = 10;
__device__ __forceinline__ int step_class(uint32_t deg, bool compact, bool safe, bool is_det) {
    if (!safe) return 9;
    if (deg > 768u || (!compact && !is_det && deg > 512u)) return 8;
    const int k = deg <= 256u ? 0 : deg <= 384u ? 1 : deg <= 512u ? 2 : 3;
    return (compact ? 0 : 4) + k;
}

// LITE: the graph has no list of more than 512 entries and no order-unsafe row (the host has looked: tlm_graph_info) -- the
// selections for longer lists and the serial one are not compiled in.  The kernel sits at the edge of its register budget
// (64 vector / 78 scalar at 8 waves per SIMD): code that never runs still shapes the allocation of the code that does.
template <bool SQRT, bool ACCOUNT, bool LITE>
__device__ __forceinline__ void mc_walk(const HotArgs &a, const WaveScratch &ws, uint32_t *mt, uint32_t *win, double *after,
                        uint32_t *ring, int lane, uint32_t ui, uint32_t start, uint32_t seed, int64_t len0,
                        uint32_t *acct) {
    // `start` and every node id below are walk-order indices
    Rng rng{mt, ws.rng, 0};
    mt_seed((lds_u32 *) mt, (glb_u32 *) ws.rng, lane, seed);
    Visited vs;
    vs.win = win;
    vs.lo = start > WIN_BITS / 2 ? (start - WIN_BITS / 2) & ~31u : 0u;
    vs.bm = ws.bm;
    for (uint32_t i = lane; i < WIN_WORDS; i += 64) win[i] = 0u;
    PathRing pr{ring, 0u, 0u};
#ifdef TLM_CHECKED
    pr.cap = a.cap;
#endif
    uint32_t n = start;
    uint32_t m = 0;          // p.edges_.size()
    uint32_t nlog = 0;       // marks that went to the HBM map (ids outside the LDS band)
    int64_t acc = 0;         // sum of the increments of edges 0..m-2 (all but the last pushed one)
    uint32_t pend_qs = 0, pend_ts = 0;   // q_start / t_start of the last pushed edge, in flight
    uint32_t have_pend = 0u;   // (32-bit on purpose, see Selection)
    int backtracks = 0;
    uint32_t steps = 0, draws = 0;
    uint64_t evals = 0;
    uint64_t lbytes = 0;     // ACCOUNT only
    bool acceptable = false;
    uint8_t status = ST_ABANDONED;
    if (lane == 0) {
        ws.pn[0] = start;                               // :39
    }
    wave_sync();
    if (lane == 0) {                                    // :33, start lies inside the band by construction
        const uint32_t d = start - vs.lo;
        win[d >> 5] |= 1u << (d & 31u);
    }
    wave_sync();
    uint2 rec = row_record(a.rows, n);
    StepClock clk;
    clk.start();
    for (;;) {                                          // :48
        {   // keep the band around the walk (rare: about once per walk)
            const uint32_t want = band_base_for(n);
            if (__builtin_expect(want - vs.lo + BAND_SLACK > 2u * BAND_SLACK, 0)) {
                if (nlog + 1 > a.cap) {
                    status = ST_SCRATCH_OVF;
                    break;
                }
                if (lane == 0 && TLM_IN(CHK_LOG, nlog, a.cap)) ws.lg[nlog] = 0x80000000u | (vs.lo >> 5);
                nlog++;
                band_slide((lds_u32 *) win, (glb_u32 *) ws.bm, cold_args()->bm_words - WALK_RNG_WORDS, vs.lo, want, lane);
                vs.lo = want;
#ifdef TLM_TIMING
                if (lane == 0) atomicAdd(&tlm_timing[4], 1ull << 32);
#endif
            }
        }
        clk.mark(CLK_ROW_WAIT, true);
        const uint32_t row = rfl(rec.x);
        const uint32_t deg = rfl(rec.y) & 0xFFFFFFu;
        const bool safe = ((rfl(rec.y) >> 24) & a.flag_bit) != 0;
        const bool anchor_nbr = ((rfl(rec.y) >> 28) & 1u) != 0;
        const bool compact = ((rfl(rec.y) >> 29) & 1u) != 0;
        // ---- selection scan (:53-115)
        const Selection s =
            LITE ? (compact ? (deg <= 256u   ? select_short_c<4, SQRT>(a, vs, rng, lane, row, deg, n, draws, clk)
                               : deg <= 384u ? select_short_c<6, SQRT>(a, vs, rng, lane, row, deg, n, draws, clk)
                                             : select_short_c<8, SQRT>(a, vs, rng, lane, row, deg, n, draws, clk))
                            : (deg <= 256u   ? select_short<4, SQRT>(a, vs, rng, lane, row, deg, start, anchor_nbr, draws, clk)
                               : deg <= 384u ? select_short<6, SQRT>(a, vs, rng, lane, row, deg, start, anchor_nbr, draws, clk)
                                             : select_short<8, SQRT>(a, vs, rng, lane, row, deg, start, anchor_nbr, draws, clk)))
            : !safe ? select_serial<SQRT>(a, vs, rng, lane, row, deg, start, draws, false, 0.0)
            : __builtin_expect(compact && deg <= 768u, 1)
                ? (deg <= 256u   ? select_short_c<4, SQRT>(a, vs, rng, lane, row, deg, n, draws, clk)
                   : deg <= 384u ? select_short_c<6, SQRT>(a, vs, rng, lane, row, deg, n, draws, clk)
                   : deg <= 512u ? select_short_c<8, SQRT>(a, vs, rng, lane, row, deg, n, draws, clk)
                                 : select_short_c<12, SQRT>(a, vs, rng, lane, row, deg, n, draws, clk))
            : deg <= 256u ? select_short<4, SQRT>(a, vs, rng, lane, row, deg, start, anchor_nbr, draws, clk)
            : deg <= 384u ? select_short<6, SQRT>(a, vs, rng, lane, row, deg, start, anchor_nbr, draws, clk)
            : deg <= 512u ? select_short<8, SQRT>(a, vs, rng, lane, row, deg, start, anchor_nbr, draws, clk)
                          : select_long<SQRT>(a, vs, rng, after, lane, row, deg, start, anchor_nbr, draws);
        evals += s.evals;
        steps++;
        if (ACCOUNT) {
            lbytes += step_layout_bytes(deg, compact);
            if (lane == 0) atomicAdd(acct + step_class(deg, compact, safe, false), 1u);
        }
        clk.mark(CLK_SELECT, false);
#ifdef TLM_TIMING
        clk.steps++;
        if (anchor_nbr) clk.acc[CLK_OTHER]++;       // steps on rows that hold an anchor
#endif
        if (__builtin_expect(s.pick < 0, 0)) {
            if (s.ncand != 0) break;    // no prefix reached `random` (NaN / negative metrics): the reference
                                        // dereferences a null Edge* there; oracle and kernel abandon the walk
            // :72 dead end
            if (backtracks < a.backtrack_attempts && m > 1) {    // :76
                ring_sync(pr, ws, lane);                // the pops below read the path from the scratch
                bool t = false;
                do {                                    // :78-88
                    // pop edge m-1: it was the pending one (never added); the new last edge m-2 leaves the sum
                    m--;
                    have_pend = 0u;
                    if (m >= 1) acc -= edge_delta(a.qt, rfl(ws.pe[m - 1]));
                    const uint32_t back = rfl(ws.pn[m]);
                    const uint2 brec = row_record(a.rows, back);
                    const uint32_t brow = rfl(brec.x);
                    const uint32_t bdeg = rfl(brec.y) & 0xFFFFFFu;
                    for (uint32_t base = 0; base < bdeg && !t; base += 64) {
                        const uint32_t i = base + lane;
                        const bool valid = i < bdeg && TLM_IN(CHK_BACKPROBE, (uint64_t) brow + i, TLM_N_ENT);
                        const uint32_t nbr = valid ? (a.ents[brow + i].x & 0x7FFFFFFFu) : 0u;
                        if (__ballot(valid && !visited_test(vs, valid ? nbr : vs.lo))) t = true;
                    }
                } while (m != 0 && !t);
                pr.base = m;                            // the ring is empty; the next edge pushed is edge m again
                if (m == 0) break;                      // :90-95
                ++backtracks;
                n = rfl(ws.pn[m]);                      // :98
                {   // the new last edge's increment becomes the pending one again
                    const uint32_t le = rfl(ws.pe[m - 1]);
                    const uint2 qt = qt_record(a.qt, le);
                    pend_qs = qt.x;
                    pend_ts = qt.y;
                    have_pend = 1u;
                }
                rec = row_record(a.rows, n);
                continue;
            }
            break;                                      // :101
        }
        // ---- advance (:118-123)
        rec = row_record(a.rows, s.nbr);        // the next step's row record: requested before this step's bookkeeping
        if (__builtin_expect(m + 2 > a.cap || nlog + 1 > a.cap, 0)) {
            status = ST_SCRATCH_OVF;
            break;
        }
        const uint32_t eidx = row + (uint32_t) s.pick;
        const uint32_t md = s.nbr - vs.lo;
        const bool mark_far = md >= WIN_BITS && !s.forced;     // a forced pick ends the walk: nothing probes it again
        if (lane == 0) {
            pr.ring[pr.cnt] = eidx;                                             // edge m
            pr.ring[PATH_RING + pr.cnt] = s.nbr;                                // node m + 1
            if (md < WIN_BITS) atomicOr(win + (md >> 5), 1u << (md & 31u));     // :123 (ds_or_b32)
            else if (mark_far && TLM_IN(CHK_MAP, s.nbr, TLM_N_NODES) && TLM_IN(CHK_LOG, nlog, a.cap)) {
                ws.bm[s.nbr >> 5] |= 1u << (s.nbr & 31);
                ws.lg[nlog] = s.nbr;
            }
        }
        if (mark_far) nlog++;
        // :125 -- Path::length() after this push excludes the new last edge (Path.cpp:12-15): it is len0 plus the
        // increments of all EARLIER edges.  The increment of the edge pushed in the previous step was requested
        // then (pend_*) and is only consumed here, a whole step later, so its latency is never waited for.
        // (readfirstlane: the compiler takes the loaded words for per-lane values, and a loop whose exit depends on
        // them for a divergent loop -- every loop-carried scalar would then live in a vector register)
        if (have_pend) acc += (int64_t) rfl(pend_qs) - (int64_t) rfl(pend_ts);
        const int64_t len = len0 + acc;
#ifdef TLM_ABLATE_NOQT      /* timing-only build: no {q_start, t_start} load (results are wrong) */
        pend_qs = 8u;
        pend_ts = 3u;
#else
        {
            const uint2 qt = qt_record(a.qt, eidx);
            pend_qs = qt.x;
            pend_ts = qt.y;
        }
#endif
        have_pend = 1u;
        m++;
        n = s.nbr;
        wave_sync();
        if (++pr.cnt == PATH_RING) ring_flush(pr, ws, lane);
        clk.mark(CLK_ADVANCE, false);
        if (len <= 0) break;                            // :126-128
        if (s.anchor) {                                 // :131-134
            acceptable = true;
            break;
        }
        if (len >= a.len_threshold) break;              // :137-142
    }
    clk.flush(lane);
    ring_sync(pr, ws, lane);
    // acc holds the increments of edges 0..m-2 (the last pushed edge is still pending): exactly Path::length()
    const int64_t len_final = len0 + acc;
    if (acceptable && status != ST_SCRATCH_OVF) status = len_final < 0 ? ST_NEGATIVE : ST_ACCEPTED;   // :146-155
    if (ACCOUNT && lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(cold_args()->cursor + 10), (unsigned long long) lbytes);
    finish_unit(ws, lane, ui, status, m, nlog, len_final, steps, evals, draws);
}

// ---------------------------------------------------------------------------------------------
// One deterministic walk from (start, first edge): PathManager.cpp:210-384.
//
// The reference copies the list and stable_sorts it descending by metric (:289-296), then takes the
// first rank k in [skip_n_best, deg-2] whose neighbour is unvisited (:299-315).  Rank order is
// (metric desc, list position asc).  With ord = an order-preserving u32 image of the metric:
//   rank 0      = the largest ord, at its smallest position;
//   rank deg-1  = the smallest ord, at its largest position (never eligible, :302);
//   the wanted entry = the largest ord among the unvisited entries other than rank deg-1 and, after a go-back
//   (skip_n_best == 1), rank 0 -- at its smallest position.
// A wave finds each with a 32-bit DPP max/min (six v_max_u32_dpp) and a find-first over ballots.

__device__ __forceinline__ uint32_t det_ord(uint32_t bits) {
    if (bits == 0x80000000u) bits = 0u;                               // -0 == +0 for operator>
    return (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
}

template <bool SQRT>
__device__ __forceinline__ uint32_t det_ord_of(uint32_t mb) {
    float mf = __uint_as_float(mb);
    if (SQRT) mf = sqrtf(mf);
    return det_ord(__float_as_uint(mf));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v, uint32_t identity) {       // lanes without a source read `identity`
    return (uint32_t) __builtin_amdgcn_update_dpp((int) identity, (int) v, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {      // result wave-uniform
    v = max(v, dpp_u32<0x111, 0xf>(v, 0u));
    v = max(v, dpp_u32<0x112, 0xf>(v, 0u));
    v = max(v, dpp_u32<0x114, 0xf>(v, 0u));
    v = max(v, dpp_u32<0x118, 0xf>(v, 0u));
    v = max(v, dpp_u32<0x142, 0xa>(v, 0u));
    v = max(v, dpp_u32<0x143, 0xc>(v, 0u));
    return readlane_u32(v, 63);
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, dpp_u32<0x111, 0xf>(v, ~0u));
    v = min(v, dpp_u32<0x112, 0xf>(v, ~0u));
    v = min(v, dpp_u32<0x114, 0xf>(v, ~0u));
    v = min(v, dpp_u32<0x118, 0xf>(v, ~0u));
    v = min(v, dpp_u32<0x142, 0xa>(v, ~0u));
    v = min(v, dpp_u32<0x143, 0xc>(v, ~0u));
    return readlane_u32(v, 63);
}

// Which of a lane's 4 entries may be taken (bit k): inside the list, unvisited, not rank deg-1, not the excluded
// rank 0.  Visited probes as in the Monte-Carlo kernel (LDS band, exact HBM map for ids outside it).
__device__ __forceinline__ uint32_t det_eligible(const Chunk &c, const Visited &vs, uint32_t base, uint32_t deg, int lane,
                                                 uint32_t worst_i, uint32_t excluded_i) {
    uint32_t vis[4];
    lane_probe<4>(vs, c.nb, vis);
    uint32_t m = 0u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t pos = base + 4u * (uint32_t) lane + (uint32_t) k;
        if (pos < deg && vis[k] == 0u && pos != worst_i && pos != excluded_i) m |= 1u << k;       // :308
    }
    return m;
}

struct DetPick {
    bool go_back;           // no eligible entry (:318)
    uint32_t chosen;        // list position of the wanted entry
    uint32_t chosen_nbr;    // its neighbour
    bool chosen_anchor;     // ... and whether that is an anchor
    uint32_t pushed;        // :340-347: the FIRST entry in list order with the same (q_index, t_index) is pushed
};

// The common case of a ranked step, at a third of the vector instructions of the general form below.  When every metric
// of the list has its sign bit clear (ES = fabsf(..) always; OS unless the record is degenerate) the fp32 BIT PATTERNS
// order exactly as operator> orders the floats (+NaN included, as det_ord places it), so no order image is needed; and
// the two exclusions of :299-308 are checked AFTER the choice instead of being searched for first:
//   * rank deg-1 (smallest metric, last position): the chosen entry is not it whenever some entry of the list has a
//     strictly smaller metric -- looked for only in lanes whose K entries all lie inside the list (no validity masks);
//     if none is found the general form decides;
//   * rank 0 after a go-back (skip_n_best == 1): found as the first position of the wave maximum; if the choice is that
//     entry it is knocked out and the choice repeated.
// No validity masks are needed for the choice itself: lanes past the list repeat lane 0's entries (same keys at larger
// positions), and the partial lane's tail is row padding (walk_layout_kernel: the row's own node, metric +0, padded so
// that a lane's K entries never reach the next row) -- the weakest possible candidate at positions >= deg: it is
// chosen only when no real entry is eligible, which is the go-back.
// Returns false when the general form has to decide (r untouched).
template <int K, bool SQRT, bool REL>
__device__ __forceinline__ bool det_select_fast(const uint32_t (&nb)[K], const uint32_t (&mb)[K], const uint32_t (&vis)[K],
                                                int lane, uint32_t deg, uint32_t skip_n_best, uint32_t lo, DetPick &r) {
    uint32_t ob[K], orr = 0u, lmin = ~0u, lall = 0u;
#pragma unroll
    for (int k = 0; k < K; k++) {
        ob[k] = SQRT ? __float_as_uint(sqrtf(__uint_as_float(mb[k]))) : mb[k];
        orr |= SQRT ? (ob[k] | mb[k]) : ob[k];
        lmin = min(lmin, ob[k]);
        lall = max(lall, ob[k]);
    }
    if (__builtin_expect(__ballot((int) orr < 0) != 0, 0)) return false;
    if ((int) deg - K * lane < K) lmin = ~0u;
    const uint32_t omin_full = wave_min_u32(lmin);
    // Round 4: NO ARRAY OF KEYS.  Positions are lane-major (lane l owns K*l .. K*l+K-1), so the first position of the wave's
    // largest key lies in the FIRST LANE that holds it, at that lane's smallest k.  Every lane therefore keeps, while its
    // entries go by, its largest key, the first k that has it and the neighbour there (a compare and three selects per
    // entry, on the vector side); one DPP maximum, one ballot and two v_readlane make the choice.  Before: key[K] beside
    // nb[K] / mb[K] / vis[K] -- 48 values per lane at K = 12 -- and a ballot, a find-first, a v_readlane and a scalar
    // minimum PER k for the first position, again for rank 0 and again for the pushed entry: ~270 scalar instructions per
    // step on the chr1-shaped graph, scalar pipe 83 % busy (profiles/r03_chr1_summary.json).
    // Rank 0 (:299: skipped for the rest of the walk after a go-back) = the first position of the list's largest metric,
    // visited or not.  It is found FIRST and taken out of the candidates, instead of choosing, comparing and choosing again.
    uint32_t xk = 0xFFu;                                        // this lane's excluded k (0xFF: none)
    const bool skip = skip_n_best != 0u;
    if (skip) {
        const uint32_t omax = wave_max_u32(lall);
        uint32_t kb = 0xFFu;
#pragma unroll
        for (int k = K - 1; k >= 0; k--) kb = ob[k] == omax ? (uint32_t) k : kb;
        const uint64_t bm = __ballot(kb != 0xFFu);              // never empty
        const int bfl = (__ffsll((unsigned long long) bm) - 1) & 63;
        if (lane == bfl) xk = kb;
    }
    const bool past = (uint32_t) (K * lane) >= deg;             // (the copies of lane 0's entries go with an excluded entry)
    uint32_t lc = 0u, ksel = 0u, nsel = 0u;
    auto stream = [&](auto excl) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            uint32_t key = (ob[k] | 0x80000000u) & ~vis[k];     // 0: visited
            if (decltype(excl)::value) key = (past || xk == (uint32_t) k) ? 0u : key;
            const bool g = key > lc;                            // strictly: the first k keeps it
            lc = g ? key : lc;
            ksel = g ? (uint32_t) k : ksel;
            nsel = g ? nb[k] : nsel;
        }
    };
    if (skip) stream(std::true_type{});
    else stream(std::false_type{});
    const uint32_t oc = wave_max_u32(lc);
    if (oc == 0u) {                                             // nothing unvisited: :318
        r.go_back = true;
        return true;
    }
    const uint64_t hm = __ballot(lc == oc);
    const int fl = (__ffsll((unsigned long long) hm) - 1) & 63;
    const uint32_t p = (uint32_t) K * (uint32_t) fl + readlane_u32(ksel, fl);
    const uint32_t e = readlane_u32(nsel, fl);
    if (p >= deg) {                                             // ... but row padding
        r.go_back = true;
        return true;
    }
    if (__builtin_expect(!(omin_full < (oc & 0x7FFFFFFFu)), 0)) return false;      // may be rank deg-1
    r.chosen = p;
    r.chosen_nbr = REL ? e + lo : e & 0x7FFFFFFFu;
    r.chosen_anchor = REL ? false : (e >> 31) != 0;
    // ---- :340-347 (the anchor bit belongs to the neighbour: equal words <=> equal neighbours); the chosen entry itself
    // matches, so the ballot is never empty
    uint32_t kp = 0xFFu;
#pragma unroll
    for (int k = K - 1; k >= 0; k--) kp = nb[k] == e ? (uint32_t) k : kp;
    const uint64_t pm = __ballot(kp != 0xFFu);
    const int pfl = (__ffsll((unsigned long long) pm) - 1) & 63;
    r.pushed = min(r.pushed, (uint32_t) K * (uint32_t) pfl + readlane_u32(kp, pfl));
    return true;
}

template <bool SQRT>
__device__ __noinline__ DetPick det_select_long_general(const uint2 *ents_tab, uint32_t *vs_win, uint32_t vs_lo, uint32_t *vs_bm, int lane,
                                                        uint32_t row, uint32_t deg, uint32_t skip_n_best);

// Lists of up to 64*K entries, K consecutive entries per lane, everything in registers: one read of the list per step.
// REL: the row is read in its compact form (metric + 16-bit relative id; `nb` then holds ids relative to the band base
// and the list holds no anchor): `node` = the row's own node.
template <int K, bool SQRT, bool REL = false>
__device__ __forceinline__ DetPick det_select_regs(const WalkArgs &a, const Visited &vs, int lane, uint32_t row,
                                                   uint32_t deg, uint32_t skip_n_best, uint32_t node) {
    DetPick r{false, 0xFFFFFFFFu, 0u, false, 0xFFFFFFFFu};
    uint32_t nb[K], mb[K], vis[K];
    // (lanes past the list repeat lane 0's entries, or read the zero block in front of the compact rows)
    if (REL) lane_load_c<K>(a.met, a.d16, row, deg, lane, node - vs.lo - 32768u, nb, mb);
    else lane_load<K>(a.ents, row, 0u, deg, lane, nb, mb);
    if (REL) lane_probe_rel<K>(vs, nb, vis);
    else lane_probe<K>(vs, nb, vis);
    if (det_select_fast<K, SQRT, REL>(nb, mb, vis, lane, deg, skip_n_best, vs.lo, r)) return r;
    // A negative metric in the list, or a choice that may be rank deg-1: the general form decides.  It reads the list again,
    // as 8-byte rows (every row has them), in a function of its own: nothing of this step has to stay in registers for it
    // (round 4; before, the general form ran on nb / mb / vis here, which kept all three arrays alive to the end of the
    // common path above).
    return det_select_long_general<SQRT>(a.ents, vs.win, vs.lo, vs.bm, lane, row, deg, skip_n_best);
}

// Longer lists, the general form: chunks of 256 entries, the first one kept in registers, the others re-read pass by pass
// (five passes over the list).  Since round 4 only the fallback of det_select_long below: lists with a negative metric, and
// the case where the choice may be rank deg-1.
template <bool SQRT>
__device__ __noinline__ DetPick det_select_long_general(const uint2 *ents_tab, uint32_t *vs_win, uint32_t vs_lo, uint32_t *vs_bm, int lane,
                                                        uint32_t row, uint32_t deg, uint32_t skip_n_best) {
    struct { const uint2 *ents; } a{ents_tab};
    Visited vs;
    vs.win = vs_win;
    vs.lo = vs_lo;
    vs.bm = vs_bm;
    bool go_back = false;
    uint32_t chosen = 0;
    // the first 256 entries stay in registers through all passes; longer lists re-read the rest (L1 / L2)
    const Chunk c0 = load_chunk(a.ents, row, 0u, deg, lane);
    uint32_t o0[4];
#pragma unroll
    for (int k = 0; k < 4; k++) o0[k] = det_ord_of<SQRT>(c0.mb[k]);
    uint32_t chosen_nbr = 0;
    bool chosen_anchor = false;
    {
        // ---- rank 0 and rank deg-1 (:289-296)
        uint32_t lmax = 0u, lmin = ~0u;
        for (uint32_t base = 0; base < deg; base += 256) {
            const Chunk c = base == 0 ? c0 : load_chunk(a.ents, row, base, deg, lane);
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (base + 4u * (uint32_t) lane + (uint32_t) k < deg) {
                    const uint32_t o = base == 0 ? o0[k] : det_ord_of<SQRT>(c.mb[k]);
                    lmax = max(lmax, o);
                    lmin = min(lmin, o);
                }
        }
        const uint32_t omax = wave_max_u32(lmax), omin = wave_min_u32(lmin);
        uint32_t best_i = 0xFFFFFFFFu, worst_i = 0u;
        for (uint32_t base = 0; base < deg; base += 256) {
            const Chunk c = base == 0 ? c0 : load_chunk(a.ents, row, base, deg, lane);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool valid = base + 4u * (uint32_t) lane + (uint32_t) k < deg;
                const uint32_t o = base == 0 ? o0[k] : det_ord_of<SQRT>(c.mb[k]);
                const uint64_t bm = __ballot(valid && o == omax), wm = __ballot(valid && o == omin);
                if (bm) best_i = min(best_i, base + 4u * (uint32_t) (__ffsll((unsigned long long) bm) - 1) + (uint32_t) k);
                if (wm) worst_i = max(worst_i, base + 4u * (uint32_t) (63 - __clzll((unsigned long long) wm)) + (uint32_t) k);
            }
        }
        const uint32_t excluded_i = skip_n_best != 0 ? best_i : 0xFFFFFFFFu;        // after a go-back (:299)
        // ---- the best eligible entry (:299-315)
        uint32_t lc = 0u, elig0 = 0u;
        bool any = false;
        for (uint32_t base = 0; base < deg; base += 256) {
            const Chunk c = base == 0 ? c0 : load_chunk(a.ents, row, base, deg, lane);
            const uint32_t em = det_eligible(c, vs, base, deg, lane, worst_i, excluded_i);
            if (base == 0) elig0 = em;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if ((em >> k) & 1u) lc = max(lc, base == 0 ? o0[k] : det_ord_of<SQRT>(c.mb[k]));
            any |= em != 0u;
        }
        if (!__ballot(any)) {
            go_back = true;                         // :318
        } else {
            const uint32_t oc = wave_max_u32(lc);
            chosen = 0xFFFFFFFFu;
            for (uint32_t base = 0; base < deg && chosen == 0xFFFFFFFFu; base += 256) {
                const Chunk c = base == 0 ? c0 : load_chunk(a.ents, row, base, deg, lane);
                const uint32_t em = base == 0 ? elig0 : det_eligible(c, vs, base, deg, lane, worst_i, excluded_i);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint64_t hm = __ballot(((em >> k) & 1u) && (base == 0 ? o0[k] : det_ord_of<SQRT>(c.mb[k])) == oc);
                    if (hm) {
                        const int fl = __ffsll((unsigned long long) hm) - 1;
                        const uint32_t p = base + 4u * (uint32_t) fl + (uint32_t) k;
                        if (p < chosen) {
                            chosen = p;
                            const uint32_t e = readlane_u32(c.nb[k], fl);
                            chosen_nbr = e & 0x7FFFFFFFu;
                            chosen_anchor = (e >> 31) != 0;
                        }
                    }
                }
            }
        }
    }
    // :340-347: the FIRST entry in list order with the same (q_index, t_index) is pushed
    uint32_t pushed = chosen;
    for (uint32_t base = 0; base < deg; base += 256) {
        const Chunk c = base == 0 ? c0 : load_chunk(a.ents, row, base, deg, lane);
        uint32_t best = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t pos = base + 4u * (uint32_t) lane + (uint32_t) k;
            const uint64_t hm = __ballot(pos < deg && (c.nb[k] & 0x7FFFFFFFu) == chosen_nbr);
            if (hm) {
                const uint32_t p = base + 4u * (uint32_t) (__ffsll((unsigned long long) hm) - 1) + (uint32_t) k;
                best = p < best ? p : best;
            }
        }
        if (best != 0xFFFFFFFFu) {
            pushed = best;
            break;
        }
    }
    return DetPick{go_back, chosen, chosen_nbr, chosen_anchor, pushed};
}

// Longer lists (round 4): det_select_fast's reasoning, chunk by chunk.  On the hostile graphs 8 % of the deterministic steps
// stand on lists of more than 768 entries -- hubs of up to 22 000 -- and carry 58 % of the evals; the general form above reads
// such a list five times (HBM traffic of the launch: 2.0x the bytes its steps ask for, profiles/r04_hard_summary.json).
// With every metric's sign bit clear the fp32 bit patterns order as operator> does, and the two exclusions of :299-308 are
// checked after the choice, so:
//   pass A (whole list, 512 entries a time, probes included): the largest unvisited key WITH its first position and
//          neighbour -- the choice (:299-315) --, the smallest metric of the list and, after a go-back, the first position
//          of the largest (rank 0: if that IS the choice, pass A again without it);
//   pass C (up to the first hit, never past the choice): the first entry with the chosen neighbour (:340-347).
// (The first form of this round had a pass B between them that re-read the list up to the first hit to find the position.)
// Everything else -- a negative metric anywhere, a choice that may be rank deg-1 -- goes to the general form.
template <bool SQRT>
__device__ __forceinline__ DetPick det_select_long(const uint2 *ents_tab, const Visited &vs, int lane, uint32_t row, uint32_t deg,
                                                   uint32_t skip_n_best) {
    constexpr int K = 8;
    constexpr uint32_t C = 64u * K;
    auto ord_bits = [](uint32_t mb) { return SQRT ? __float_as_uint(sqrtf(__uint_as_float(mb))) : mb; };
    // ---- pass A (the only pass with probes): every lane keeps, over ALL chunks, its largest unvisited key with the first
    // position that has it and the neighbour there (det_select_fast's running choice; a lane meets its positions in
    // increasing order, so "strictly larger" keeps the first), the extrema of the metric and -- after a go-back -- the first
    // position of its largest metric.  The wave's choice is then the smallest kept position among the lanes that hold the
    // wave maximum: no second pass over the list to find it (round 4, second form; the first re-read the list up to the hit).
    uint32_t excluded = 0xFFFFFFFFu;                    // position knocked out after a go-back (rank 0), if it was the choice
    uint32_t oc = 0u, omin = ~0u, omax = 0u, best_i = 0xFFFFFFFFu;
    const bool skip = skip_n_best != 0u;
    for (int round = 0; round < 2; round++) {
        uint32_t lc = 0u, psel = 0xFFFFFFFFu, nsel = 0u, lmin = ~0u, lall = 0u, pall = 0xFFFFFFFFu, orr = 0u;
        for (uint32_t base = 0; base < deg; base += C) {
            uint32_t nb[K], mb[K], vis[K];
            lane_load<K>(ents_tab, row, base, deg, lane, nb, mb);
            lane_probe<K>(vs, nb, vis);
#pragma unroll
            for (int k = 0; k < K; k++) {
                const uint32_t pos = base + (uint32_t) K * (uint32_t) lane + (uint32_t) k;
                const uint32_t ob = ord_bits(mb[k]);
                if (pos < deg) {
                    orr |= SQRT ? (ob | mb[k]) : ob;
                    lmin = min(lmin, ob);
                    if (skip && round == 0 && ob > lall) pall = pos;
                    lall = max(lall, ob);
                    const uint32_t key = pos != excluded ? (ob | 0x80000000u) & ~vis[k] : 0u;
                    if (key > lc) {
                        lc = key;
                        psel = pos;
                        nsel = nb[k];
                    }
                }
            }
        }
        if (__builtin_expect(__ballot((int) orr < 0) != 0, 0)) return det_select_long_general<SQRT>(ents_tab, vs.win, vs.lo, vs.bm, lane, row, deg, skip_n_best);
        oc = wave_max_u32(lc);
        if (round == 0) {
            omin = wave_min_u32(lmin);
            if (skip) {
                omax = wave_max_u32(lall);
                best_i = wave_min_u32(lall == omax ? pall : 0xFFFFFFFFu);          // rank 0: first position of the largest metric
            }
        }
        if (oc == 0u) return DetPick{true, 0u, 0u, false, 0u};                      // :318: nothing unvisited
        const uint32_t p = wave_min_u32(lc == oc ? psel : 0xFFFFFFFFu);
        const uint64_t hm = __ballot(lc == oc && psel == p);
        const uint32_t e = readlane_u32(nsel, (__ffsll((unsigned long long) hm) - 1) & 63);
        if (round == 0 && skip && p == best_i) {        // the choice is rank 0, skipped for the rest of the walk (:299): once more without it
            excluded = best_i;
            continue;
        }
        if (p >= deg) return DetPick{true, 0u, 0u, false, 0u};
        if (__builtin_expect(!(omin < (oc & 0x7FFFFFFFu)), 0))                       // may be rank deg-1: the general form decides
            return det_select_long_general<SQRT>(ents_tab, vs.win, vs.lo, vs.bm, lane, row, deg, skip_n_best);
        // ---- pass C (:340-347): the first entry, in list order, that names the chosen neighbour (the anchor bit belongs to the
        // neighbour: equal words <=> equal neighbours); it lies at or before the choice
        uint32_t pushed = p;
        for (uint32_t base = 0; base <= p; base += C) {
            uint32_t nb[K], mb[K];
            lane_load<K>(ents_tab, row, base, deg, lane, nb, mb);
            uint32_t best = 0xFFFFFFFFu, kp = 0xFFu;    // (lane-major inside a chunk: the first lane with a match, its smallest k)
#pragma unroll
            for (int k = K - 1; k >= 0; k--) {
                const uint32_t pos = base + (uint32_t) K * (uint32_t) lane + (uint32_t) k;
                kp = (pos < deg && nb[k] == e) ? (uint32_t) k : kp;
            }
            const uint64_t hm = __ballot(kp != 0xFFu);
            if (hm) {
                const int fl = __ffsll((unsigned long long) hm) - 1;
                best = base + (uint32_t) K * (uint32_t) fl + readlane_u32(kp, fl);
            }
            if (best != 0xFFFFFFFFu) {
                pushed = best;
                break;
            }
        }
        return DetPick{false, p, e & 0x7FFFFFFFu, (e >> 31) != 0, pushed};
    }
    return DetPick{true, 0u, 0u, false, 0u};            // (not reached: the second round either chooses or goes back)
}

// set / clear a visited flag (lane 0 only)
__device__ __forceinline__ void visited_set(const Visited &vs, uint32_t n) {
    const uint32_t d = n - vs.lo;
    if (d < WIN_BITS) atomicOr(vs.win + (d >> 5), 1u << (d & 31u));
    else if (TLM_IN(CHK_MAP, n, TLM_N_NODES)) vs.bm[n >> 5] |= 1u << (n & 31);
}

__device__ __forceinline__ void visited_clear(const Visited &vs, uint32_t n) {
    const uint32_t d = n - vs.lo;
    if (d < WIN_BITS) atomicAnd(vs.win + (d >> 5), ~(1u << (d & 31u)));
    else if (TLM_IN(CHK_MAP, n, TLM_N_NODES)) vs.bm[n >> 5] &= ~(1u << (n & 31));
}

template <bool SQRT, bool ACCOUNT, bool LITE>
__device__ __forceinline__ void det_walk(const WalkArgs &a, const WaveScratch &ws, uint32_t *win, uint32_t *ring, int lane, uint32_t ui,
                         uint32_t start, uint32_t start_ref, uint32_t first_slot, uint32_t *acct) {
    Visited vs;
    vs.win = win;
    vs.lo = start > WIN_BITS / 2 ? (start - WIN_BITS / 2) & ~31u : 0u;
    vs.bm = ws.bm;
    for (uint32_t i = lane; i < WIN_WORDS; i += 64) win[i] = 0u;
    const uint32_t first = rfl(row_record(a.rows, start).x) + first_slot;
    const uint2 ent0 = TLM_IN(CHK_LIST8, first, TLM_N_ENT) ? a.ents[first] : make_uint2(start, 0u);
    const uint32_t second = rfl(ent0.x & 0x7FFFFFFFu);
    const bool second_anchor = rfl(ent0.x >> 31) != 0;
    const int64_t len0 = TLM_IN(CHK_NODE_TABLE, start_ref, TLM_N_NODES) ? (int64_t) a.node_len[start_ref] : 0;
    uint32_t m = 1, nlog = 0;           // nlog: marks that went to the HBM map
    PathRing pr{ring, 1u, 0u};          // edge 0 and nodes 0, 1 are written directly below
#ifdef TLM_CHECKED
    pr.cap = a.cap;
#endif
    uint32_t steps = 0;
    uint64_t evals = 0;
    uint64_t lbytes = 0;     // ACCOUNT only
    uint8_t status = ST_ABANDONED;
    wave_sync();
    if (lane == 0) {
        ws.pn[0] = start;                                   // :216
        ws.pe[0] = first;                                   // :217
        ws.pn[1] = second;                                  // :222
        visited_set(vs, start);                             // :215
    }
    wave_sync();
    if (lane == 0) {
        visited_set(vs, second);                            // :221
        if (second - vs.lo >= WIN_BITS && TLM_IN(CHK_LOG, 0u, a.cap)) ws.lg[0] = second;
    }
    if (second - vs.lo >= WIN_BITS) nlog = 1;               // start lies inside the band by construction
    wave_sync();
    bool all_ok = false;
    if (second_anchor) {
        status = ST_ACCEPTED;                               // :224-229: appended without a length check
    } else {
        int step_index = 0;
        uint32_t skip_n_best = 0;
        bool went_back = false;
        uint32_t node = second;
        uint2 rec = row_record(a.rows, node);
        for (;;) {                                          // :243
            {   // keep the band around the walk (see band_slide)
                const uint32_t want = band_base_for(node);
                if (__builtin_expect(want - vs.lo + BAND_SLACK > 2u * BAND_SLACK, 0)) {
                    if (nlog + 1 > a.cap) {
                        status = ST_SCRATCH_OVF;
                        break;
                    }
                    if (lane == 0 && TLM_IN(CHK_LOG, nlog, a.cap)) ws.lg[nlog] = 0x80000000u | (vs.lo >> 5);
                    nlog++;
                    band_slide((lds_u32 *) win, (glb_u32 *) ws.bm, a.bm_words - WALK_RNG_WORDS, vs.lo, want, lane);
                    vs.lo = want;
                }
            }
            const uint32_t row = rfl(rec.x);
            const uint32_t deg = rfl(rec.y) & 0xFFFFFFu;
            bool go_back = false;
            DetPick pk{true, 0u, 0u, false, 0u};            // :245, :267: nothing to rank
            if (deg != 0 && skip_n_best < deg - 1) {
                evals += deg;
                steps++;
                const bool compact = ((rfl(rec.y) >> 29) & 1u) != 0;
                if (ACCOUNT) {
                    // (no {q_start, t_start} inside the loop; lists of 513..768 entries are read once here even as 8-byte rows)
                    lbytes += (!compact && deg > 512u && deg <= 768u ? ((deg + 11u) / 12u) * 96u + 16u : step_layout_bytes(deg, compact)) - 8u;
                    if (lane == 0) atomicAdd(acct + step_class(deg, compact, true, true), 1u);
                }
                if (LITE)       // (no list of more than 512 entries in this graph: the host has looked)
                    pk = compact ? (deg <= 256u   ? det_select_regs<4, SQRT, true>(a, vs, lane, row, deg, skip_n_best, node)
                                    : deg <= 384u ? det_select_regs<6, SQRT, true>(a, vs, lane, row, deg, skip_n_best, node)
                                                  : det_select_regs<8, SQRT, true>(a, vs, lane, row, deg, skip_n_best, node))
                                 : (deg <= 256u   ? det_select_regs<4, SQRT>(a, vs, lane, row, deg, skip_n_best, node)
                                    : deg <= 384u ? det_select_regs<6, SQRT>(a, vs, lane, row, deg, skip_n_best, node)
                                                  : det_select_regs<8, SQRT>(a, vs, lane, row, deg, skip_n_best, node));
                else
                pk = compact && deg <= 768u
                         ? (deg <= 256u   ? det_select_regs<4, SQRT, true>(a, vs, lane, row, deg, skip_n_best, node)
                            : deg <= 384u ? det_select_regs<6, SQRT, true>(a, vs, lane, row, deg, skip_n_best, node)
                            : deg <= 512u ? det_select_regs<8, SQRT, true>(a, vs, lane, row, deg, skip_n_best, node)
                                          : det_select_regs<12, SQRT, true>(a, vs, lane, row, deg, skip_n_best, node))
                     : deg <= 256u ? det_select_regs<4, SQRT>(a, vs, lane, row, deg, skip_n_best, node)
                     : deg <= 384u ? det_select_regs<6, SQRT>(a, vs, lane, row, deg, skip_n_best, node)
                     : deg <= 512u ? det_select_regs<8, SQRT>(a, vs, lane, row, deg, skip_n_best, node)
                     : deg <= 768u ? det_select_regs<12, SQRT>(a, vs, lane, row, deg, skip_n_best, node)
                                   : det_select_long<SQRT>(a.ents, vs, lane, row, deg, skip_n_best);
            }
            go_back = pk.go_back;
            const uint32_t chosen_nbr = pk.chosen_nbr, pushed = pk.pushed;
            const bool chosen_anchor = pk.chosen_anchor;
            if (go_back) {
                if (step_index == 0) break;                 // :246-248,268-271,319-322
                if (went_back) break;                       // :251-253,274-276,325-327
                went_back = true;
                step_index -= 1;
                skip_n_best += 1;
                ring_sync(pr, ws, lane);                    // the pop reads the path from the scratch
                m--;                                        // pop node + edge
                pr.base = m;
                node = rfl(ws.pn[m]);
                if (lane == 0) visited_clear(vs, node);     // :262,284,335
                wave_sync();
                rec = row_record(a.rows, node);
                continue;
            }
            rec = row_record(a.rows, chosen_nbr);           // the next step's row record, requested before the bookkeeping
            if (m + 2 > a.cap || nlog + 1 > a.cap) {
                status = ST_SCRATCH_OVF;
                break;
            }
            const uint32_t eidx = row + pushed;
            const bool far_mark = node - vs.lo >= WIN_BITS;
            if (lane == 0) {
                pr.ring[pr.cnt] = eidx;                     // edge m
                pr.ring[PATH_RING + pr.cnt] = chosen_nbr;   // node m + 1
                visited_set(vs, node);                      // :355 marks the node being left
                if (far_mark && TLM_IN(CHK_LOG, nlog, a.cap)) ws.lg[nlog] = node;
            }
            if (far_mark) nlog++;
            m++;
            node = chosen_nbr;                              // :356
            wave_sync();
            if (++pr.cnt == PATH_RING) ring_flush(pr, ws, lane);
            if (chosen_anchor) {                            // :364-367
                all_ok = true;
                break;
            }
            step_index += 1;
        }
    }
    ring_sync(pr, ws, lane);
    // Path::length() (Path.cpp:6-16): the first node plus the increments of all edges but the last.  Nothing in the
    // walk depends on it (:372-384 look at it once, at the end), so it is summed here, from the finished path, by all
    // lanes -- not edge by edge inside the step loop.
    int64_t part = 0;
    for (uint32_t i = lane; i + 1u < m; i += 64)
        if (TLM_IN(CHK_PATH, i, a.cap)) part += edge_delta(a.qt, ws.pe[i]);
    for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
    const int64_t len_final = len0 + (int64_t) __shfl(part, 0);
    if (all_ok && status != ST_SCRATCH_OVF) status = len_final < 0 ? ST_NEGATIVE : ST_ACCEPTED;   // :372-384
    if (ACCOUNT && lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(a.cursor + 10), (unsigned long long) lbytes);
    finish_unit(ws, lane, ui, status, m, nlog, len_final, steps, evals, 0);
}

// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ WaveScratch wave_scratch(const WalkArgs &a, uint32_t wave) {
    WaveScratch ws;
    ws.pn = a.scratch + (size_t) wave * 3u * a.cap;
    ws.pe = ws.pn + a.cap;
    ws.lg = ws.pe + a.cap;
    ws.bm = a.bitmaps + (size_t) wave * a.bm_words;
    ws.rng = ws.bm + (a.bm_words - WALK_RNG_WORDS);
    return ws;
}

// First workgroup in / last workgroup out on the device's constant-rate clock (100 MHz), in cursor[6..9]: what a launch
// took once it had compute units, however long it waited for them.
__device__ __forceinline__ void stamp_span(uint32_t *cursor, bool leaving) {
    if (threadIdx.x == 0) {
        const unsigned long long t = wall_clock64();
        if (leaving) atomicMax(reinterpret_cast<unsigned long long *>(cursor + 8), t);
        else atomicMin(reinterpret_cast<unsigned long long *>(cursor + 6), t);
    }
}

// Persistent waves with one work queue per XCD.  The work items are cut into chunks of `chunk` consecutive items (the
// trials of one start anchor under one metric) and chunk j belongs to XCD j % 8: the waves of an XCD walk the same
// neighbourhood at the same time, so its L2 (4 MB, private to the XCD) holds the rows they share.  A wave pulls from
// its own XCD's cursor; when that queue is dry it moves on to the next XCD's and never comes back.  Cursors only
// grow, so every wave passes all eight queues and leaves: the grid drains.  Which XCD runs a unit has no influence
// on its result.
struct QueueState {
    uint32_t home;      // XCC_ID of this wave
    uint32_t passed;    // queues found dry so far
};

__device__ __forceinline__ QueueState queue_init() {
    QueueState q;
    // HW_REG_XCC_ID (id 20), bits [3:0]: s_getreg_b32 operand = id | offset << 6 | (size - 1) << 11
    q.home = (uint32_t) __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & (WALK_XCDS - 1);
    q.passed = 0;
    return q;
}

__device__ __forceinline__ bool next_work(int lane, QueueState &q, uint32_t &ui) {
    ColdArgs *ca = cold_args();
    const uint32_t n_work = ca->n_work, chunk = ca->chunk;
    uint32_t *cur = ca->cursor;
    while (q.passed < WALK_XCDS) {
        const uint32_t x = (q.home + q.passed) & (WALK_XCDS - 1);
        uint32_t p = 0;
        if (lane == 0) p = atomicAdd(cur + WALK_QUEUE_BASE + WALK_QUEUE_STRIDE * x, 1u);
        p = rfl(p);
        const uint64_t w = ((uint64_t) (p / chunk) * WALK_XCDS + x) * chunk + p % chunk;
        if (w < n_work) {
            const uint32_t *wl = ca->work_list;
            ui = wl ? rfl(wl[w]) : (uint32_t) w;
            return true;
        }
        q.passed++;
    }
    return false;
}

// work item -> (metric slot, unit); returns the arguments specialised for the slot (deterministic kernel)
__device__ __forceinline__ WalkArgs slot_args(const WalkArgs &a, uint32_t cu, uint32_t &ui) {
    WalkArgs b = a;
    const uint32_t slot = cu >= a.n_units ? 1u : 0u;      // at most two slots
    ui = cu - slot * a.n_units;
    if (slot) {
        b.ents = a.ents1;
        b.met = a.met1;
        b.flag_bit = a.flag_bit1;
        b.metric = a.metric1;
    }
    return b;
}

// waves per SIMD the Monte-Carlo kernel is compiled for (8 = the architecture's maximum: 64 vector / 78 scalar registers)
#ifndef TLM_MC_WAVES
#define TLM_MC_WAVES 8
#endif
template <bool SQRT, bool ACCOUNT = false, bool LITE = false>
__global__ __launch_bounds__(WALK_BLOCK, TLM_MC_WAVES) void walk_mc_kernel(WalkArgs a) {
    __shared__ __attribute__((aligned(WIN_WORDS * 4))) uint32_t win_s[WALK_WAVES_PER_BLOCK][WIN_WORDS];
    __shared__ uint32_t mt_s[WALK_WAVES_PER_BLOCK][624];
    __shared__ double after_s[WALK_WAVES_PER_BLOCK][AFTER_SLOTS];
    __shared__ uint32_t ring_s[WALK_WAVES_PER_BLOCK][2 * PATH_RING];
    __shared__ uint32_t acct_s[ACCOUNT ? WALK_WAVES_PER_BLOCK : 1][STEP_CLASSES];      // ACCOUNT builds only
    const int lane = threadIdx.x & 63;
    const int wib = (int) rfl(threadIdx.x >> 6);      // wave-uniform: keeps the per-wave pointers in SGPRs
    const WaveScratch ws = wave_scratch(a, blockIdx.x * WALK_WAVES_PER_BLOCK + wib);
    if (ACCOUNT) {
        if (lane < STEP_CLASSES) acct_s[wib][lane] = 0u;
        wave_sync();
    }
    if (threadIdx.x == 0) {      // "all workgroups are resident" for a host that wants to queue another launch behind this one
        ColdArgs *ca = cold_args();
        uint32_t *flag = ca->resident_flag;
        if (flag && atomicAdd(ca->cursor + 1, 1u) + 1u == gridDim.x) {
            __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    uint32_t cu;
    QueueState q = queue_init();
    while (next_work(lane, q, cu)) {
        // work item -> (metric slot, unit) -> (anchor, trial); everything but the step loop's own arguments is read
        // from the kernarg segment here and dropped again
        ColdArgs *ca = cold_args();
        const uint32_t n_units = ca->n_units;
        const bool slot1 = cu >= n_units;                  // at most two slots
        const uint32_t ui = cu - (slot1 ? n_units : 0u);
        const uint64_t *ids = ca->unit_ids;
        const uint64_t unit = ids ? ((__attribute__((address_space(4))) const uint64_t *) ids)[ui] : ca->unit_begin + ui;
        const uint64_t rb = (uint64_t) ca->rebuild_attempts;
        const uint32_t a_rank = (uint32_t) (unit / rb);
        const uint32_t r = (uint32_t) (unit % rb);
        if (!TLM_IN(CHK_ANCHOR, a_rank, ca->n_anchors)) continue;
        const uint32_t start = rfl(ca->w_anchors[a_rank]);
        const uint32_t start_ref = rfl(ca->anchors[a_rank]);        // the anchor's reference index
        const int metric = slot1 ? ca->metric1 : ca->metric;
        HotArgs h;
        h.rows = a.rows;
        h.ents = slot1 ? ca->ents1 : a.ents;
        h.qt = a.qt;
        h.met = slot1 ? ca->met1 : a.met;
        h.d16 = a.d16;
        h.cap = a.cap;
        h.flag_bit = slot1 ? ca->flag_bit1 : a.flag_bit;
        h.backtrack_attempts = a.backtrack_attempts;
        h.len_threshold = a.len_threshold;
        const int64_t len0 = (int64_t) rfl(ca->node_len[start_ref]);
        mc_walk<SQRT, ACCOUNT, LITE>(h, ws, mt_s[wib], win_s[wib], after_s[wib], ring_s[wib], lane, cu, start,
                      walk_seed_dev(metric, start_ref, r), len0, ACCOUNT ? acct_s[wib] : nullptr);
    }
    if (ACCOUNT) {
        wave_sync();
        if (lane < STEP_CLASSES)
            atomicAdd(reinterpret_cast<unsigned long long *>(cold_args()->cursor + 12) + lane, (unsigned long long) acct_s[wib][lane]);
    }
}

// 6 waves per SIMD (80 vector registers): measured 9.9 ms against 10.2 ms at 5 and 9.96 ms at 8 (A. thaliana config)
#ifndef TLM_DET_WAVES
#define TLM_DET_WAVES 6
#endif
#ifndef TLM_DET_LITE_WAVES
#define TLM_DET_LITE_WAVES 8
#endif
template <bool SQRT, bool ACCOUNT = false, bool LITE = false>
__global__ __launch_bounds__(WALK_BLOCK, LITE ? TLM_DET_LITE_WAVES : TLM_DET_WAVES) void walk_det_kernel(WalkArgs a) {
    __shared__ __attribute__((aligned(WIN_WORDS * 4))) uint32_t win_s[WALK_WAVES_PER_BLOCK][WIN_WORDS];
    __shared__ uint32_t ring_s[WALK_WAVES_PER_BLOCK][2 * PATH_RING];
    __shared__ uint32_t acct_s[ACCOUNT ? WALK_WAVES_PER_BLOCK : 1][STEP_CLASSES];      // ACCOUNT builds only
    const int lane = threadIdx.x & 63;
    const int wib = (int) rfl(threadIdx.x >> 6);      // wave-uniform: keeps the per-wave pointers in SGPRs
    const WaveScratch ws = wave_scratch(a, blockIdx.x * WALK_WAVES_PER_BLOCK + wib);
    if (ACCOUNT) {
        if (lane < STEP_CLASSES) acct_s[wib][lane] = 0u;
        wave_sync();
    }
    stamp_span(a.cursor, false);
    uint32_t cu;
    QueueState q = queue_init();
    while (next_work(lane, q, cu)) {
        uint32_t ui;
        const WalkArgs b = slot_args(a, cu, ui);
        const uint64_t *ids = cold_args()->unit_ids;        // (from the kernarg segment: not a register across the walk)
        const uint64_t unit = ids ? ((__attribute__((address_space(4))) const uint64_t *) ids)[ui] : a.unit_begin + ui;
        uint32_t lo = 0, hi = a.n_anchors;          // largest k with det_off[k] <= unit
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) / 2;
            if (a.det_off[mid] <= unit) lo = mid;
            else hi = mid;
        }
        det_walk<SQRT, ACCOUNT, LITE>(b, ws, win_s[wib], ring_s[wib], lane, cu, rfl(a.w_anchors[lo]), rfl(a.anchors[lo]),
                       (uint32_t) (unit - a.det_off[lo]), ACCOUNT ? acct_s[wib] : nullptr);
    }
    if (ACCOUNT) {
        wave_sync();
        if (lane < STEP_CLASSES)
            atomicAdd(reinterpret_cast<unsigned long long *>(a.cursor + 12) + lane, (unsigned long long) acct_s[wib][lane]);
    }
    __syncthreads();
    stamp_span(a.cursor, true);
}

}  // namespace

int tlm_walk_blocks_per_cu(bool is_mc, bool sqrt_metric, bool lite, bool account) {
    int n = 0;
    hipError_t e;
    auto ask = [&](auto kernel) { return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, WALK_BLOCK, 0); };
    if (account) {          // (accounting launches are never LITE)
        if (is_mc) e = sqrt_metric ? ask(walk_mc_kernel<true, true>) : ask(walk_mc_kernel<false, true>);
        else e = sqrt_metric ? ask(walk_det_kernel<true, true>) : ask(walk_det_kernel<false, true>);
    } else if (lite) {
        if (is_mc) e = sqrt_metric ? ask(walk_mc_kernel<true, false, true>) : ask(walk_mc_kernel<false, false, true>);
        else e = sqrt_metric ? ask(walk_det_kernel<true, false, true>) : ask(walk_det_kernel<false, false, true>);
    } else {
        if (is_mc) e = sqrt_metric ? ask(walk_mc_kernel<true>) : ask(walk_mc_kernel<false>);
        else e = sqrt_metric ? ask(walk_det_kernel<true>) : ask(walk_det_kernel<false>);
    }
    if (e != hipSuccess || n < 1) n = 1;
    return n;
}

#ifdef TLM_CHECKED
static void chk_arm(hipStream_t s, const WalkArgs &a) {
    ChkLimits lim{a.n_ent, a.n_nodes, a.n_anchors};
    if (const char *e = getenv("TLM_CHECK_SHRINK"))        // the check's own test: pretend the list arrays are shorter
        lim.n_ent = (unsigned long long) ((double) lim.n_ent * atof(e));
    unsigned long long z[4] = {0, 0, 0, 0};
    (void) hipStreamSynchronize(s);
    const hipError_t e1 = hipMemcpyToSymbol(HIP_SYMBOL(tlm_chk), &lim, sizeof(lim));
    const hipError_t e2 = hipMemcpyToSymbol(HIP_SYMBOL(tlm_oob), z, sizeof(z));
    if (getenv("TLM_CHECK_VERBOSE"))
        fprintf(stderr, "[checked build] limits: %llu entries, %u nodes, %u anchors, cap %u, map words %u (%s, %s)\n", lim.n_ent, lim.n_nodes,
                lim.n_anchors, a.cap, a.bm_words, hipGetErrorString(e1), hipGetErrorString(e2));
}
static int chk_collect(hipStream_t s, const char *kernel) {
    unsigned long long h[4] = {0, 0, 0, 0};
    (void) hipStreamSynchronize(s);
    const hipError_t e = hipMemcpyFromSymbol(h, HIP_SYMBOL(tlm_oob), sizeof(h));
    if (getenv("TLM_CHECK_VERBOSE"))
        fprintf(stderr, "[checked build] %s: read-back %s; site %llu index %llu limit %llu count %llu\n", kernel, hipGetErrorString(e), h[0],
                h[1], h[2], h[3]);
    if (e != hipSuccess) {
        tlm_set_error("checked build: cannot read the offence record: %s", hipGetErrorString(e));
        return 1;
    }
    if (!h[0]) return 0;
    tlm_set_error("checked build: %s formed an index outside its array %llu time(s); first: site %llu, index %llu, limit %llu",
                  kernel, h[3], h[0], h[1], h[2]);
    return 1;
}
#endif

int tlm_launch_walk_mc(hipStream_t s, unsigned blocks, bool sqrt_metric, const WalkArgs &a, bool account, bool lite) {
#ifdef TLM_CHECKED
    chk_arm(s, a);
#endif
    if (account) {      // same walks, plus the layout-byte count (never the timed launch)
        if (sqrt_metric) hipLaunchKernelGGL((walk_mc_kernel<true, true>), dim3(blocks), dim3(WALK_BLOCK), 0, s, a);
        else hipLaunchKernelGGL((walk_mc_kernel<false, true>), dim3(blocks), dim3(WALK_BLOCK), 0, s, a);
#ifdef TLM_CHECKED
        return chk_collect(s, "walk_mc_kernel (accounting)");
#endif
        return 0;
    }
#ifdef TLM_TIMING
    {
        unsigned long long z[8] = {0};
        (void) hipMemcpyToSymbol(HIP_SYMBOL(tlm_timing), z, sizeof(z));
    }
#endif
    if (lite) {
        if (sqrt_metric) hipLaunchKernelGGL((walk_mc_kernel<true, false, true>), dim3(blocks), dim3(WALK_BLOCK), 0, s, a);
        else hipLaunchKernelGGL((walk_mc_kernel<false, false, true>), dim3(blocks), dim3(WALK_BLOCK), 0, s, a);
    } else if (sqrt_metric) hipLaunchKernelGGL(walk_mc_kernel<true>, dim3(blocks), dim3(WALK_BLOCK), 0, s, a);
    else hipLaunchKernelGGL(walk_mc_kernel<false>, dim3(blocks), dim3(WALK_BLOCK), 0, s, a);
#ifdef TLM_TIMING
    {
        unsigned long long h[8] = {0};
        (void) hipStreamSynchronize(s);
        (void) hipMemcpyFromSymbol(h, HIP_SYMBOL(tlm_timing), sizeof(h));
        if (h[5])
            fprintf(stderr, "[tlm timing] %llu steps (%.1f %% on rows that hold an anchor; probes outside the LDS band on %.1f %% "
                            "of the steps: %.1f %% compact rows + %.1f %% other rows); cycles per step: row-record wait "
                            "%.0f, list wait %.0f, selection %.0f, bookkeeping %.0f; band moves %llu\n", h[5],
                    100.0 * (h[4] & 0xFFFFFFFFull) / h[5],
                    100.0 * (h[6] + h[7]) / h[5], 100.0 * h[6] / h[5], 100.0 * h[7] / h[5],
                    (double) h[0] / h[5], (double) h[1] / h[5], (double) h[2] / h[5], (double) h[3] / h[5], h[4] >> 32);
    }
#endif
#ifdef TLM_CHECKED
    return chk_collect(s, "walk_mc_kernel");
#endif
    return 0;
}

int tlm_launch_walk_det(hipStream_t s, unsigned blocks, bool sqrt_metric, const WalkArgs &a, bool account, bool lite) {
#ifdef TLM_CHECKED
    chk_arm(s, a);
#endif
    if (account) {
        if (sqrt_metric) hipLaunchKernelGGL((walk_det_kernel<true, true>), dim3(blocks), dim3(WALK_BLOCK), 0, s, a);
        else hipLaunchKernelGGL((walk_det_kernel<false, true>), dim3(blocks), dim3(WALK_BLOCK), 0, s, a);
#ifdef TLM_CHECKED
        return chk_collect(s, "walk_det_kernel (accounting)");
#endif
        return 0;
    }
    if (lite) {
        if (sqrt_metric) hipLaunchKernelGGL((walk_det_kernel<true, false, true>), dim3(blocks), dim3(WALK_BLOCK), 0, s, a);
        else hipLaunchKernelGGL((walk_det_kernel<false, false, true>), dim3(blocks), dim3(WALK_BLOCK), 0, s, a);
    } else if (sqrt_metric) hipLaunchKernelGGL(walk_det_kernel<true>, dim3(blocks), dim3(WALK_BLOCK), 0, s, a);
    else hipLaunchKernelGGL(walk_det_kernel<false>, dim3(blocks), dim3(WALK_BLOCK), 0, s, a);
#ifdef TLM_CHECKED
    return chk_collect(s, "walk_det_kernel");
#endif
    return 0;
}

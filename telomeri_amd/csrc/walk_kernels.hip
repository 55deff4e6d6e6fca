// telomeri_amd/csrc/walk_kernels.hip -- see walk_kernels.h.
#include "walk_kernels.h"

namespace {

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) v); }

// Orders a wave's own LDS / global traffic between its lanes (lane 0 writes, all lanes read):
// the wave executes in lockstep and its memory instructions issue in order, so this only has to
// stop the compiler from moving accesses across it.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------
// std::mt19937, one generator per wave, state in LDS (624 words), all lanes hold the same index.

__device__ __forceinline__ uint32_t walk_seed_dev(int metric, uint32_t anchor_node, uint32_t r) {
    // == tlm_walk_seed() / tlo_walk_seed()
    return 5489u + 0x9E3779B9u * (anchor_node + 1u) + 0x85EBCA6Bu * r + 0xC2B2AE35u * (uint32_t) metric;
}

__device__ inline void mt_seed(volatile uint32_t *mt, int lane, uint32_t seed) {
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
}

__device__ inline void mt_twist(volatile uint32_t *mt, int lane) {
    // In-place regeneration in chunks of 64 words.  Word i needs old mt[i], old mt[i+1] and
    // mt[(i+397)%624], which is an OLD word for i < 227 (it lies >= 6 chunks ahead) and a NEW word
    // for i >= 227 (it lies >= 3 chunks behind); word 623 needs the new mt[0].  Within a chunk all
    // reads are issued before the chunk's writes.
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
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// uniform_real_distribution<double>(0,1)(gen): two draws, low word first, RN(x0 + x1*2^32) / 2^64,
// clamped below 1.  The index is always even, so a pair never straddles a twist.
__device__ inline double mt_canonical(volatile uint32_t *mt, int &idx, int lane) {
    if (idx >= 624) {
        mt_twist(mt, lane);
        idx = 0;
    }
    const uint32_t y0 = mt_temper(mt[idx]), y1 = mt_temper(mt[idx + 1]);
    idx += 2;
    const double sum = (double) y0 + (double) y1 * 4294967296.0;
    double ret = sum * 0x1p-64;                   // exact scaling
    if (ret >= 1.0) ret = 0x1.fffffffffffffp-1;   // nextafter(1, 0)
    return ret;
}

// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ bool bm_test(const uint32_t *bm, uint32_t n) { return (bm[n >> 5] >> (n & 31)) & 1u; }

__device__ __forceinline__ int64_t edge_delta(const WalkArgs &a, uint32_t e) {
    return (int64_t) a.q_start[e] - (int64_t) a.t_start[e];    // Path.cpp:14
}

// Inclusive prefix, in list order, of the candidates' metrics of one 64-entry chunk, continuing
// `carry`.  safe: every partial sum is exact in a double, so the log-step scan equals the
// reference's sequential `sum += metric` bit for bit; otherwise the additions are done one
// candidate after the other in list order, exactly as PathManager.cpp:64,109 does them.
__device__ inline double chunk_scan(double mv, uint64_t cm, double &carry, bool safe, int lane) {
    double incl;
    if (safe) {
        incl = mv;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        incl += carry;
        carry = __shfl(incl, 63);
    } else {
        double run = carry;
        incl = 0.0;
        uint64_t mm = cm;
        while (mm) {
            const int j = __ffsll((unsigned long long) mm) - 1;
            mm &= mm - 1;
            run += __shfl(mv, j);
            if (lane == j) incl = run;
        }
        carry = run;
    }
    return incl;
}

__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint64_t o = __shfl_xor((unsigned long long) v, d);
        v = o > v ? o : v;
    }
    return v;
}

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint64_t o = __shfl_xor((unsigned long long) v, d);
        v = o < v ? o : v;
    }
    return v;
}

struct WaveScratch {
    uint32_t *pn;     // path: nodes
    uint32_t *pe;     // path: global CSR index of each edge
    uint32_t *lg;     // every node whose visited bit was set (to clear the bitmap afterwards)
    uint32_t *bm;     // N-bit visited set, all zero between walks
};

// End of a walk: publish the unit's outcome, copy an accepted path into the arena, clear the bitmap.
__device__ inline void finish_unit(const WalkArgs &a, const WaveScratch &ws, int lane, uint32_t ui, uint8_t status,
                                   uint32_t m, uint32_t nlog, int64_t len_final, uint32_t steps, uint64_t evals,
                                   uint32_t draws) {
    const uint32_t n_out = m + 1;
    uint64_t off = 0;
    if (status == ST_ACCEPTED) {
        unsigned long long o = 0;
        if (lane == 0) o = atomicAdd(a.ar_cursor, (unsigned long long) n_out);
        off = __shfl(o, 0);
        if (off + n_out > a.ar_cap) {
            status = ST_ARENA_OVF;
        } else {
            for (uint32_t i = lane; i < n_out; i += 64) {
                const uint32_t node = ws.pn[i];
                a.ar_nodes[off + i] = node;
                a.ar_slots[off + i] = (i < m) ? (ws.pe[i] - a.row_ptr[node]) : 0xFFFFFFFFu;
            }
        }
    }
    if (lane == 0) {
        a.u_status[ui] = status;
        a.u_nnodes[ui] = n_out;
        a.u_len[ui] = len_final;
        a.u_steps[ui] = steps;
        a.u_evals[ui] = evals;
        a.u_draws[ui] = draws;
        a.u_off[ui] = off;
    }
    for (uint32_t i = lane; i < nlog; i += 64) ws.bm[ws.lg[i] >> 5] = 0u;
    wave_sync();
}

// ---------------------------------------------------------------------------------------------
// One Monte-Carlo trial: PathManager.cpp:30-160.

template <bool SQRT>
__device__ void mc_walk(const WalkArgs &a, const WaveScratch &ws, volatile uint32_t *mt, int lane, uint32_t ui,
                        uint32_t start, uint32_t r) {
    int mti = 624;
    mt_seed(mt, lane, walk_seed_dev(a.metric, start, r));
    uint32_t n = start;
    uint32_t m = 0;          // p.edges_.size()
    uint32_t nlog = 1;
    int64_t acc = 0;         // sum of the increments of ALL m edges
    const int64_t len0 = (int64_t) a.node_len[start];
    int backtracks = 0;
    uint32_t steps = 0, draws = 0;
    uint64_t evals = 0;
    bool acceptable = false;
    uint8_t status = ST_ABANDONED;
    if (lane == 0) {
        ws.pn[0] = start;                               // :39
        ws.bm[start >> 5] |= 1u << (start & 31);        // :33
        ws.lg[0] = start;
    }
    wave_sync();
    for (;;) {                                          // :48
        const uint32_t row = rfl(a.row_ptr[n]);
        const uint32_t deg = rfl(a.row_ptr[n + 1]) - row;
        const bool safe = (rfl(a.node_flags[n]) & a.flag_bit) != 0;
        // ---- selection scan (:53-67): forced anchor edge, else sum over unvisited neighbours
        int forced = -1;
        uint32_t pick_nbr = 0;
        bool pick_anchor = false;
        double total = 0.0;
        uint32_t ncand = 0;
        for (uint32_t base = 0; base < deg; base += 64) {
            const uint32_t i = base + lane;
            const bool valid = i < deg;
            const uint2 ent = valid ? a.ents[row + i] : make_uint2(0u, 0u);
            const uint32_t nbr = ent.x & 0x7FFFFFFFu;
            const uint64_t fm = __ballot(valid && (ent.x >> 31) && nbr != start);     // :57
            if (fm) {
                const int fl = __ffsll((unsigned long long) fm) - 1;
                forced = (int) base + fl;
                pick_nbr = __shfl(nbr, fl);
                pick_anchor = true;
                break;
            }
            const bool cand = valid && !bm_test(ws.bm, nbr);                         // :63
            const uint64_t cm = __ballot(cand);
            if (cm) {
                float mf = __uint_as_float(ent.y);
                if (SQRT) mf = sqrtf(mf);
                const double mv = cand ? (double) mf : 0.0;
                (void) chunk_scan(mv, cm, total, safe, lane);                         // :64
                ncand += (uint32_t) __popcll((unsigned long long) cm);
            }
        }
        evals += forced >= 0 ? (uint64_t) (forced + 1) : (uint64_t) deg;
        steps++;
        int pick = -1;
        if (forced >= 0) {
            pick = forced;                              // :70-71
        } else if (ncand == 0) {                        // :72 dead end
            if (backtracks < a.backtrack_attempts && m > 1) {    // :76
                bool t = false;
                do {                                    // :78-88
                    m--;
                    acc -= edge_delta(a, rfl(ws.pe[m]));
                    const uint32_t back = rfl(ws.pn[m]);
                    const uint32_t brow = rfl(a.row_ptr[back]);
                    const uint32_t bdeg = rfl(a.row_ptr[back + 1]) - brow;
                    for (uint32_t base = 0; base < bdeg && !t; base += 64) {
                        const uint32_t i = base + lane;
                        const bool valid = i < bdeg;
                        const uint32_t nbr = valid ? (a.ents[brow + i].x & 0x7FFFFFFFu) : 0u;
                        if (__ballot(valid && !bm_test(ws.bm, nbr))) t = true;
                    }
                } while (m != 0 && !t);
                if (m == 0) break;                      // :90-95
                ++backtracks;
                n = rfl(ws.pn[m]);                      // :98
                continue;
            }
            break;                                      // :101
        } else {
            const double random = mt_canonical(mt, mti, lane) * total;               // :104
            draws++;
            double carry = 0.0;                         // :107
            for (uint32_t base = 0; base < deg; base += 64) {
                const uint32_t i = base + lane;
                const bool valid = i < deg;
                const uint2 ent = valid ? a.ents[row + i] : make_uint2(0u, 0u);
                const uint32_t nbr = ent.x & 0x7FFFFFFFu;
                const bool cand = valid && !bm_test(ws.bm, nbr);
                const uint64_t cm = __ballot(cand);
                if (!cm) continue;
                float mf = __uint_as_float(ent.y);
                if (SQRT) mf = sqrtf(mf);
                const double mv = cand ? (double) mf : 0.0;
                const double incl = chunk_scan(mv, cm, carry, safe, lane);            // :109
                const uint64_t hm = __ballot(cand && incl >= random);                // :110
                if (hm) {
                    const int hl = __ffsll((unsigned long long) hm) - 1;
                    pick = (int) base + hl;
                    pick_nbr = __shfl(nbr, hl);
                    pick_anchor = (__shfl(ent.x, hl) >> 31) != 0;
                    break;
                }
            }
            // no prefix reached `random` (NaN / negative metrics): the reference dereferences a null
            // Edge* there; the oracle and this kernel abandon the walk instead.
            if (pick < 0) break;
        }
        // ---- advance (:118-123)
        if (m + 2 > a.cap || nlog + 1 > a.cap) {
            status = ST_SCRATCH_OVF;
            break;
        }
        const uint32_t eidx = row + (uint32_t) pick;
        if (lane == 0) {
            ws.pe[m] = eidx;
            ws.pn[m + 1] = pick_nbr;
            ws.bm[pick_nbr >> 5] |= 1u << (pick_nbr & 31);
            ws.lg[nlog] = pick_nbr;
        }
        nlog++;
        const int64_t len = len0 + acc;                 // :125 -- the new last edge is excluded (Path.cpp:12-15)
        acc += edge_delta(a, eidx);
        m++;
        n = pick_nbr;
        wave_sync();
        if (len <= 0) break;                            // :126-128
        if (pick_anchor) {                              // :131-134
            acceptable = true;
            break;
        }
        if (len >= a.len_threshold) break;              // :137-142
    }
    const int64_t len_final = len0 + acc - (m ? edge_delta(a, rfl(ws.pe[m - 1])) : 0);
    if (acceptable && status != ST_SCRATCH_OVF) status = len_final < 0 ? ST_NEGATIVE : ST_ACCEPTED;   // :146-155
    finish_unit(a, ws, lane, ui, status, m, nlog, len_final, steps, evals, draws);
}

// ---------------------------------------------------------------------------------------------
// One deterministic walk from (start, first edge): PathManager.cpp:210-384.
//
// The reference copies the list and stable_sorts it descending by metric (:289-296), then takes the
// first rank k in [skip_n_best, deg-2] whose neighbour is unvisited (:299-315).  Rank order is
// (metric desc, list position asc); so with key = (orderable(metric) << 32) | ~position, rank 0 is
// the maximum key, rank deg-1 the minimum key, and the wanted entry is the maximum key among the
// unvisited entries other than rank deg-1 and, after a go-back (skip_n_best == 1), rank 0.

__device__ __forceinline__ uint64_t det_key(uint32_t bits, uint32_t i) {
    if (bits == 0x80000000u) bits = 0u;                               // -0 == +0 for operator>
    const uint32_t ord = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
    return ((uint64_t) ord << 32) | (uint64_t) (0xFFFFFFFFu - i);
}

template <bool SQRT>
__device__ void det_walk(const WalkArgs &a, const WaveScratch &ws, int lane, uint32_t ui, uint32_t start,
                         uint32_t first_slot) {
    const uint32_t first = rfl(a.row_ptr[start]) + first_slot;
    const uint2 ent0 = a.ents[first];
    const uint32_t second = rfl(ent0.x & 0x7FFFFFFFu);
    const bool second_anchor = rfl(ent0.x >> 31) != 0;
    const int64_t len0 = (int64_t) a.node_len[start];
    uint32_t m = 1, nlog = 2;
    int64_t acc = edge_delta(a, first);
    uint32_t steps = 0;
    uint64_t evals = 0;
    uint8_t status = ST_ABANDONED;
    if (lane == 0) {
        ws.pn[0] = start;                                   // :216
        ws.pe[0] = first;                                   // :217
        ws.pn[1] = second;                                  // :222
        ws.bm[start >> 5] |= 1u << (start & 31);            // :215
        ws.lg[0] = start;
    }
    wave_sync();
    if (lane == 0) {
        ws.bm[second >> 5] |= 1u << (second & 31);          // :221
        ws.lg[1] = second;
    }
    wave_sync();
    bool all_ok = false;
    if (second_anchor) {
        status = ST_ACCEPTED;                               // :224-229: appended without a length check
    } else {
        int step_index = 0;
        uint32_t skip_n_best = 0;
        bool went_back = false;
        uint32_t node = second;
        for (;;) {                                          // :243
            const uint32_t row = rfl(a.row_ptr[node]);
            const uint32_t deg = rfl(a.row_ptr[node + 1]) - row;
            bool go_back = false;
            uint32_t chosen = 0;
            if (deg == 0 || skip_n_best >= deg - 1) {       // :245, :267
                go_back = true;
            } else {
                evals += deg;
                steps++;
                uint64_t kmax = 0, kmin = ~0ull;
                for (uint32_t base = 0; base < deg; base += 64) {
                    const uint32_t i = base + lane;
                    if (i < deg) {
                        float mf = __uint_as_float(a.ents[row + i].y);
                        if (SQRT) mf = sqrtf(mf);
                        const uint64_t k = det_key(__float_as_uint(mf), i);
                        kmax = k > kmax ? k : kmax;
                        kmin = k < kmin ? k : kmin;
                    }
                }
                kmax = wave_max_u64(kmax);
                kmin = wave_min_u64(kmin);
                const uint32_t best_i = 0xFFFFFFFFu - (uint32_t) kmax;     // rank 0
                const uint32_t worst_i = 0xFFFFFFFFu - (uint32_t) kmin;    // rank deg-1: never eligible (:302)
                uint64_t kc = 0;
                for (uint32_t base = 0; base < deg; base += 64) {
                    const uint32_t i = base + lane;
                    if (i < deg) {
                        const uint2 ent = a.ents[row + i];
                        const bool elig = i != worst_i && !(skip_n_best != 0 && i == best_i)
                                          && !bm_test(ws.bm, ent.x & 0x7FFFFFFFu);           // :308
                        if (elig) {
                            float mf = __uint_as_float(ent.y);
                            if (SQRT) mf = sqrtf(mf);
                            const uint64_t k = det_key(__float_as_uint(mf), i);
                            kc = k > kc ? k : kc;
                        }
                    }
                }
                kc = wave_max_u64(kc);
                if (kc == 0) go_back = true;                // :318
                else chosen = 0xFFFFFFFFu - (uint32_t) kc;
            }
            if (go_back) {
                if (step_index == 0) break;                 // :246-248,268-271,319-322
                if (went_back) break;                       // :251-253,274-276,325-327
                went_back = true;
                step_index -= 1;
                skip_n_best += 1;
                m--;                                        // pop node + edge
                acc -= edge_delta(a, rfl(ws.pe[m]));
                node = rfl(ws.pn[m]);
                if (lane == 0) ws.bm[node >> 5] &= ~(1u << (node & 31));     // :262,284,335
                wave_sync();
                continue;
            }
            const uint2 cent = a.ents[row + chosen];
            const uint32_t chosen_nbr = rfl(cent.x & 0x7FFFFFFFu);
            const bool chosen_anchor = rfl(cent.x >> 31) != 0;
            // :340-347: the FIRST entry in list order with the same (q_index, t_index) is pushed
            uint32_t pushed = chosen;
            for (uint32_t base = 0; base < deg; base += 64) {
                const uint32_t i = base + lane;
                const uint64_t hm = __ballot(i < deg && (a.ents[row + (i < deg ? i : 0)].x & 0x7FFFFFFFu) == chosen_nbr);
                if (hm) {
                    pushed = base + (uint32_t) (__ffsll((unsigned long long) hm) - 1);
                    break;
                }
            }
            if (m + 2 > a.cap || nlog + 1 > a.cap) {
                status = ST_SCRATCH_OVF;
                break;
            }
            const uint32_t eidx = row + pushed;
            if (lane == 0) {
                ws.pe[m] = eidx;
                ws.pn[m + 1] = chosen_nbr;
                ws.bm[node >> 5] |= 1u << (node & 31);      // :355 marks the node being left
                ws.lg[nlog] = node;
            }
            nlog++;
            acc += edge_delta(a, eidx);
            m++;
            node = chosen_nbr;                              // :356
            wave_sync();
            if (chosen_anchor) {                            // :364-367
                all_ok = true;
                break;
            }
            step_index += 1;
        }
    }
    const int64_t len_final = len0 + acc - edge_delta(a, rfl(ws.pe[m - 1]));
    if (all_ok && status != ST_SCRATCH_OVF) status = len_final < 0 ? ST_NEGATIVE : ST_ACCEPTED;   // :372-384
    finish_unit(a, ws, lane, ui, status, m, nlog, len_final, steps, evals, 0);
}

// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ WaveScratch wave_scratch(const WalkArgs &a, uint32_t wave) {
    WaveScratch ws;
    ws.pn = a.sc_nodes + (size_t) wave * a.cap;
    ws.pe = a.sc_edges + (size_t) wave * a.cap;
    ws.lg = a.sc_log + (size_t) wave * a.cap;
    ws.bm = a.bitmaps + (size_t) wave * a.bm_words;
    return ws;
}

// Persistent waves: every wave pulls the next work item until the cursor passes n_work; the cursor
// only grows, so every wave reaches its exit and the grid drains.
__device__ __forceinline__ bool next_work(const WalkArgs &a, int lane, uint32_t &ui) {
    uint32_t w = 0;
    if (lane == 0) w = atomicAdd(a.cursor, 1u);
    w = rfl(w);
    if (w >= a.n_work) return false;
    ui = a.work_list ? rfl(a.work_list[w]) : w;
    return true;
}

template <bool SQRT>
__global__ __launch_bounds__(WALK_BLOCK) void walk_mc_kernel(WalkArgs a) {
    __shared__ uint32_t mt_s[WALK_WAVES_PER_BLOCK][624];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const WaveScratch ws = wave_scratch(a, blockIdx.x * WALK_WAVES_PER_BLOCK + wib);
    uint32_t ui;
    while (next_work(a, lane, ui)) {
        const uint64_t unit = a.unit_begin + ui;
        const uint32_t a_rank = (uint32_t) (unit / (uint64_t) a.rebuild_attempts);
        const uint32_t r = (uint32_t) (unit % (uint64_t) a.rebuild_attempts);
        mc_walk<SQRT>(a, ws, mt_s[wib], lane, ui, rfl(a.anchors[a_rank]), r);
    }
}

template <bool SQRT>
__global__ __launch_bounds__(WALK_BLOCK) void walk_det_kernel(WalkArgs a) {
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const WaveScratch ws = wave_scratch(a, blockIdx.x * WALK_WAVES_PER_BLOCK + wib);
    uint32_t ui;
    while (next_work(a, lane, ui)) {
        const uint64_t unit = a.unit_begin + ui;
        uint32_t lo = 0, hi = a.n_anchors;          // largest k with det_off[k] <= unit
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) / 2;
            if (a.det_off[mid] <= unit) lo = mid;
            else hi = mid;
        }
        det_walk<SQRT>(a, ws, lane, ui, rfl(a.anchors[lo]), (uint32_t) (unit - a.det_off[lo]));
    }
}

}  // namespace

void tlm_launch_walk_mc(hipStream_t s, unsigned blocks, bool sqrt_metric, const WalkArgs &a) {
    if (sqrt_metric) hipLaunchKernelGGL(walk_mc_kernel<true>, dim3(blocks), dim3(WALK_BLOCK), 0, s, a);
    else hipLaunchKernelGGL(walk_mc_kernel<false>, dim3(blocks), dim3(WALK_BLOCK), 0, s, a);
}

void tlm_launch_walk_det(hipStream_t s, unsigned blocks, bool sqrt_metric, const WalkArgs &a) {
    if (sqrt_metric) hipLaunchKernelGGL(walk_det_kernel<true>, dim3(blocks), dim3(WALK_BLOCK), 0, s, a);
    else hipLaunchKernelGGL(walk_det_kernel<false>, dim3(blocks), dim3(WALK_BLOCK), 0, s, a);
}

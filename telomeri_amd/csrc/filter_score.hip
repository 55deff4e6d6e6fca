// telomeri_amd/csrc/filter_score.hip -- per-record PAF filter + SI/OS/ES (gfx950).
//
// Replaces OverlapGraph::filter (reference src/OverlapGraph.cpp:156-202) and the score lines of
// OverlapGraph::buildFrom (src/OverlapGraph.cpp:221-231 with the helpers at :297-324).
//
// One thread per record, SoA int32 columns: every wave-load is 64 consecutive dwords (256 B),
// fully coalesced; no reuse, so no LDS.  HBM-bound: 37 B read (9 int32 columns + 1 strand byte is
// not read here: 36 B) and 13 B written per record.  fp32 arithmetic in the reference's order,
// compiled with -ffp-contract=off and correctly rounded divides, so bit patterns equal x86-64.
#include "tlm_common.h"

struct FilterConsts {
    int mode;
    float min_ol;    // (float) min_overlap_length      (usual arithmetic conversions, :195)
    float min_olp;
    float max_oh;    // (float) max_overhang_length     (:200)
    float max_ohp;
};

__global__ __launch_bounds__(256) void filter_score_kernel(FilterConsts c, uint64_t n, const int32_t *__restrict__ q_id,
                                                           const int32_t *__restrict__ t_id,
                                                           const int32_t *__restrict__ q_len,
                                                           const int32_t *__restrict__ q_start,
                                                           const int32_t *__restrict__ q_end,
                                                           const int32_t *__restrict__ t_len,
                                                           const int32_t *__restrict__ t_start,
                                                           const int32_t *__restrict__ t_end,
                                                           const int32_t *__restrict__ matches,
                                                           uint8_t *__restrict__ keep, float *__restrict__ si,
                                                           float *__restrict__ os, float *__restrict__ es,
                                                           unsigned long long *__restrict__ n_kept) {
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int ql = q_len[i], qs = q_start[i], qe = q_end[i];
        const int tl = t_len[i], ts = t_start[i], te = t_end[i];
        const int qol = qe - qs;          // getQueryOverlapLength   :297
        const int tol = te - ts;          // getTargetOverlapLength  :301
        const int qoh = ql - qe;          // getQueryOverhangLength  :305
        const int toh = ts;               // getTargetOverhangLength :309
        float overlap_length, overhang_length, total_length;
        bool ok = true;
        switch (c.mode) {                 // :170-193
            case TLM_MODE_MIN:
                overlap_length = (float) min(qol, tol);
                overhang_length = (float) min(qoh, toh);
                total_length = (float) min(ql, tl);
                break;
            case TLM_MODE_MAX:
                overlap_length = (float) max(qol, tol);
                overhang_length = (float) max(qoh, toh);
                total_length = (float) max(ql, tl);
                break;
            case TLM_MODE_AVG:
                overlap_length = (float) (qol + tol) / 2.0f;
                overhang_length = (float) (qoh + toh) / 2.0f;
                total_length = (float) (ql + tl) / 2.0f;
                break;
            case TLM_MODE_SUM:
                overlap_length = (float) (qol + tol);
                overhang_length = (float) (qoh + toh);
                total_length = (float) (ql + tl);
                break;
            default:
                overlap_length = overhang_length = total_length = 0.0f;
                ok = false;
        }
        if (q_id[i] == t_id[i]) ok = false;                                               // :157
        if (overlap_length < c.min_ol || (overlap_length / total_length) < c.min_olp) ok = false;   // :195-198
        if (!(overhang_length <= c.max_oh && (overhang_length / overlap_length) <= c.max_ohp)) ok = false;  // :200-201
        // scores, for every record (:221-231)
        const float SI = (float) matches[i] / (float) min(ql, tl);                         // :321-324
        const float OS = (float) (qol + tol) / 2.0f * SI;                                  // :228
        const float ES = fabsf(OS + (float) qs / 2.0f - (float) (qoh + toh) / 2.0f);       // :231
        if (keep) keep[i] = ok ? 1 : 0;
        if (si) si[i] = SI;
        if (os) os[i] = OS;
        if (es) es[i] = ES;
        mine += ok ? 1 : 0;
    }
    if (n_kept) {
        // one atomic per wave: ballot-free wave sum through DPP-less shuffles is overkill here
        for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d);
        if ((threadIdx.x & 63) == 0 && mine) atomicAdd(n_kept, mine);
    }
}

void tlm_launch_filter_score(hipStream_t s, const tlm_filter_params &p, const tlm_records &r, uint8_t *keep, float *si,
                             float *os, float *es, unsigned long long *n_kept_dev) {
    if (r.n == 0) return;
    FilterConsts c;
    c.mode = p.mode;
    c.min_ol = (float) p.min_overlap_length;
    c.min_olp = p.min_overlap_percentage;
    c.max_oh = (float) p.max_overhang_length;
    c.max_ohp = p.max_overhang_percentage;
    uint64_t blocks = (r.n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;   // grid-stride the rest: 16 blocks per CU
    hipLaunchKernelGGL(filter_score_kernel, dim3((unsigned) blocks), dim3(256), 0, s, c, r.n, r.q_id, r.t_id, r.q_len,
                       r.q_start, r.q_end, r.t_len, r.t_start, r.t_end, r.matches, keep, si, os, es, n_kept_dev);
}

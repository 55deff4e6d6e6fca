// telomeri_amd/csrc/pathset_ops.hip -- post-passes on packed Path sets, on the device.
//
//   tlm_pathset_concat        paths_ after several builder calls on one PathManager   reference Main.cpp:237-243
//   tlm_pathset_filter_unique PathManager::filterUnique                                reference PathManager.cpp:432-438
//                             (std::unique with Path::operator==, Path.cpp:46-49: node sequence only,
//                              ADJACENT duplicates only)
//   tlm_pathset_anchor_pairs  PathManager::getPathsBetweenAnchors                      reference PathManager.cpp:626-645
//                             (std::map keyed by (front Node*, back Node*): buckets in (front, back) index order,
//                              inside a bucket in paths_ order)
// They run where the paths already are: a rank filters and buckets its shard before the one all-gather, and the
// host receives the Path set already deduplicated.  HBM-bound streaming passes over the packed arrays.
#include <algorithm>
#include <cstring>
#include <vector>

#include "tlm_common.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace {

// path-set buffers and temporaries come from the library's device cache (device_pool.hip)
template <class T> int dmalloc(T **p, size_t n, hipStream_t s) {
    return tlm_pool_alloc((void **) p, (n ? n : 1) * sizeof(T), s);
}

struct Tmp {      // a temporary of one call: handed back in stream order
    void *p = nullptr;
    ~Tmp() { tlm_pool_free(p); }
};

int alloc_pathset(tlm_ctx *ctx, uint64_t n_paths, uint64_t n_nodes, tlm_pathset **out) {
    tlm_pathset *ps = new tlm_pathset();
    ps->ctx = ctx;
    ps->device = ctx->device;
    ps->n_paths = n_paths;
    ps->n_nodes_total = n_nodes;
    hipStream_t s = ctx->stream;
    if (dmalloc(&ps->unit, n_paths, s) || dmalloc(&ps->length, n_paths, s) || dmalloc(&ps->off, n_paths + 1, s)
        || dmalloc(&ps->nodes, n_nodes, s) || dmalloc(&ps->slots, n_nodes, s)) {
        tlm_pathset_destroy(ps);
        return TLM_EHIP;
    }
    *out = ps;
    return TLM_OK;
}

__global__ void rebase_kernel(uint64_t n, const uint64_t *__restrict__ src_unit, const uint64_t *__restrict__ src_off,
                              uint64_t tag, uint64_t node_base, uint64_t *__restrict__ dst_unit, uint64_t *__restrict__ dst_off) {
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        dst_unit[i] = src_unit[i] | tag;
        dst_off[i] = src_off[i] + node_base;
    }
}

// keep[i] = 0 iff path i has the same node sequence as path i-1 (one wave per path)
__global__ __launch_bounds__(256) void dup_flags_kernel(uint64_t n_paths, const uint64_t *__restrict__ off,
                                                        const uint32_t *__restrict__ nodes, uint8_t *__restrict__ keep) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    for (uint64_t i = wave; i < n_paths; i += n_waves) {
        bool same = false;
        if (i > 0) {
            const uint64_t a0 = off[i - 1], a1 = off[i], b1 = off[i + 1];
            if (a1 - a0 == b1 - a1) {
                same = true;
                for (uint64_t k = lane; k < a1 - a0 && same; k += 64)
                    if (nodes[a0 + k] != nodes[a1 + k]) same = false;
                same = __ballot(!same) == 0;
            }
        }
        if (lane == 0) keep[i] = same ? 0 : 1;
    }
}

// exclusive scan of keep[] and of the kept paths' node counts; one workgroup
__global__ __launch_bounds__(1024) void scan_keep_kernel(uint64_t n_paths, const uint8_t *__restrict__ keep,
                                                         const uint64_t *__restrict__ off, uint64_t *__restrict__ p_index,
                                                         uint64_t *__restrict__ n_off, uint64_t *__restrict__ totals) {
    __shared__ uint64_t s_cnt[1024];
    __shared__ uint64_t s_sum[1024];
    uint64_t carry_cnt = 0, carry_sum = 0;
    const int t = threadIdx.x;
    for (uint64_t base = 0; base < n_paths; base += 1024) {
        const uint64_t i = base + t;
        const bool k = i < n_paths && keep[i];
        const uint64_t c = k ? 1 : 0, v = k ? off[i + 1] - off[i] : 0;
        s_cnt[t] = c;
        s_sum[t] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            uint64_t a = 0, b = 0;
            if (t >= d) { a = s_cnt[t - d]; b = s_sum[t - d]; }
            __syncthreads();
            s_cnt[t] += a;
            s_sum[t] += b;
            __syncthreads();
        }
        if (i < n_paths) {
            p_index[i] = carry_cnt + s_cnt[t] - c;
            n_off[i] = carry_sum + s_sum[t] - v;
        }
        carry_cnt += s_cnt[1023];
        carry_sum += s_sum[1023];
        __syncthreads();
    }
    if (t == 0) { totals[0] = carry_cnt; totals[1] = carry_sum; }
}

__global__ __launch_bounds__(256) void copy_kept_kernel(uint64_t n_paths, const uint8_t *__restrict__ keep,
                                                        const uint64_t *__restrict__ p_index, const uint64_t *__restrict__ n_off,
                                                        const uint64_t *__restrict__ unit, const int64_t *__restrict__ length,
                                                        const uint64_t *__restrict__ off, const uint32_t *__restrict__ nodes,
                                                        const uint32_t *__restrict__ slots, uint64_t *__restrict__ o_unit,
                                                        int64_t *__restrict__ o_len, uint64_t *__restrict__ o_off,
                                                        uint32_t *__restrict__ o_nodes, uint32_t *__restrict__ o_slots) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    for (uint64_t i = wave; i < n_paths; i += n_waves) {
        if (!keep[i]) continue;
        const uint64_t pi = p_index[i], src = off[i], n = off[i + 1] - off[i], dst = n_off[i];
        if (lane == 0) {
            o_unit[pi] = unit[i];
            o_len[pi] = length[i];
            o_off[pi] = dst;
        }
        for (uint64_t k = lane; k < n; k += 64) {
            o_nodes[dst + k] = nodes[src + k];
            o_slots[dst + k] = slots[src + k];
        }
    }
}

__global__ void pair_key_kernel(uint64_t n_paths, const uint64_t *__restrict__ off, const uint32_t *__restrict__ nodes,
                                unsigned long long *__restrict__ key, uint32_t *__restrict__ idx) {
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_paths) {
        key[i] = ((unsigned long long) nodes[off[i]] << 32) | (unsigned long long) nodes[off[i + 1] - 1];
        idx[i] = (uint32_t) i;
    }
}

// one wave per path: front, back, node count and a hash of the node sequence.  The hash is position-dependent (a sum of
// per-position mixes, so lanes combine in any order): equal sequences hash equally, which is all the host relies on.
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

__global__ __launch_bounds__(256) void summary_kernel(uint64_t n_paths, const uint64_t *__restrict__ off,
                                                      const uint32_t *__restrict__ nodes, uint32_t *__restrict__ front,
                                                      uint32_t *__restrict__ back, uint32_t *__restrict__ n_nodes,
                                                      uint64_t *__restrict__ hash) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    for (uint64_t i = wave; i < n_paths; i += n_waves) {
        const uint64_t a = off[i], n = off[i + 1] - a;
        uint64_t h = 0;
        for (uint64_t k = lane; k < n; k += 64) h += mix64(((uint64_t) nodes[a + k] << 32 | (k & 0xFFFFFFFFull)) + 0x9E3779B97F4A7C15ull);
        for (int d = 32; d > 0; d >>= 1) h += __shfl_xor(h, d);
        if (lane == 0) {
            front[i] = n ? nodes[a] : 0xFFFFFFFFu;
            back[i] = n ? nodes[a + n - 1] : 0xFFFFFFFFu;
            n_nodes[i] = (uint32_t) n;
            hash[i] = mix64(h ^ n);
        }
    }
}

// path index[k] -> staging[dst_off[k] ...), one wave per requested path
__global__ __launch_bounds__(256) void fetch_kernel(uint64_t n, const uint64_t *__restrict__ index,
                                                    const uint64_t *__restrict__ dst_off, const uint64_t *__restrict__ off,
                                                    const uint32_t *__restrict__ nodes, const uint32_t *__restrict__ slots,
                                                    uint32_t *__restrict__ o_nodes, uint32_t *__restrict__ o_slots) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t) gridDim.x * blockDim.x) >> 6;
    for (uint64_t k = wave; k < n; k += n_waves) {
        const uint64_t a = off[index[k]], len = off[index[k] + 1] - a, d = dst_off[k];
        for (uint64_t j = lane; j < len; j += 64) {
            o_nodes[d + j] = nodes[a + j];
            o_slots[d + j] = slots[a + j];
        }
    }
}

__global__ void count_dropped_kernel(uint64_t n, const uint8_t *__restrict__ keep, uint32_t *__restrict__ dropped) {
    uint32_t c = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) c += keep[i] == 0;
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(dropped, c);
}

__global__ void invert_flags_kernel(uint64_t n, uint8_t *__restrict__ f) {       // keep (1 = differs) -> dup (1 = equal)
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) f[i] = f[i] ? 0 : 1;
}

unsigned blocks_for(uint64_t n, unsigned per) {
    uint64_t b = (n + per - 1) / per;
    return (unsigned) std::max<uint64_t>(1, std::min<uint64_t>(b, 256u * 16u));
}

}  // namespace

extern "C" int tlm_pathset_concat(tlm_ctx *ctx, int n, tlm_pathset *const *sets, tlm_pathset **out) {
    TLM_CHECK(ctx && sets && out && n >= 1 && n <= 255, TLM_EINVAL, "tlm_pathset_concat: bad argument");
    TLM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    uint64_t np = 0, nn = 0;
    for (int k = 0; k < n; k++) {
        TLM_CHECK(sets[k] && sets[k]->ctx == ctx, TLM_EINVAL, "tlm_pathset_concat: set %d is null or of another context", k);
        np += sets[k]->n_paths;
        nn += sets[k]->n_nodes_total;
    }
    tlm_pathset *ps = nullptr;
    if (int rc = alloc_pathset(ctx, np, nn, &ps)) return rc;
    uint64_t pb = 0, nb = 0;
    for (int k = 0; k < n; k++) {
        const tlm_pathset *q = sets[k];
        if (q->n_paths) {
            hipLaunchKernelGGL(rebase_kernel, dim3((unsigned) ((q->n_paths + 255) / 256)), dim3(256), 0, s, q->n_paths, q->unit,
                               q->off, (uint64_t) k << 56, nb, ps->unit + pb, ps->off + pb);
            TLM_HIP(hipMemcpyAsync(ps->length + pb, q->length, q->n_paths * 8, hipMemcpyDeviceToDevice, s));
        }
        if (q->n_nodes_total) {
            TLM_HIP(hipMemcpyAsync(ps->nodes + nb, q->nodes, q->n_nodes_total * 4, hipMemcpyDeviceToDevice, s));
            TLM_HIP(hipMemcpyAsync(ps->slots + nb, q->slots, q->n_nodes_total * 4, hipMemcpyDeviceToDevice, s));
        }
        pb += q->n_paths;
        nb += q->n_nodes_total;
    }
    TLM_HIP(hipMemcpyAsync(ps->off + np, &nn, 8, hipMemcpyHostToDevice, s));
    TLM_HIP(hipStreamSynchronize(s));
    TLM_HIP(hipGetLastError());
    *out = ps;
    return TLM_OK;
}

namespace {

__global__ void count_descents_kernel(uint64_t n, const uint64_t *__restrict__ unit, uint32_t *__restrict__ descents) {
    uint32_t c = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x + 1; i < n; i += (uint64_t) gridDim.x * blockDim.x)
        c += unit[i] < unit[i - 1];
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(descents, c);
}

__global__ void iota_kernel(uint64_t n, uint64_t *__restrict__ idx) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) idx[i] = i;
}

__global__ void sorted_counts_kernel(uint64_t n, const uint64_t *__restrict__ order, const uint64_t *__restrict__ off,
                                     const int64_t *__restrict__ length, uint64_t *__restrict__ cnt, int64_t *__restrict__ o_length) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint64_t j = order[i];
        cnt[i] = off[j + 1] - off[j];
        o_length[i] = length[j];
    }
}

}   // namespace

/* paths_ order is ascending (builder tag | unit): a set assembled from blocks that interleave in that order (the full
 * exchange of a sliced partition) is brought into it here -- radix sort of the units, scan of the node counts, one gather.
 * Consumes `in`; hands it back untouched when it is ascending already. */
int tlm_pathset_sort_by_unit(tlm_ctx *ctx, tlm_pathset *in, tlm_pathset **out) {
    TLM_CHECK(ctx && in && out && in->ctx == ctx, TLM_EINVAL, "tlm_pathset_sort_by_unit: bad argument");
    TLM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint64_t n = in->n_paths;
    *out = in;
    if (n < 2) return TLM_OK;
    Tmp cnt, key2, idx, idx2, tmp, ncnt;
    if (int rc = tlm_pool_alloc(&cnt.p, 4, s)) return rc;
    TLM_HIP(hipMemsetAsync(cnt.p, 0, 4, s));
    hipLaunchKernelGGL(count_descents_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, n, in->unit, (uint32_t *) cnt.p);
    uint32_t descents = 0;
    TLM_HIP(hipMemcpyAsync(&descents, cnt.p, 4, hipMemcpyDeviceToHost, s));
    TLM_HIP(hipStreamSynchronize(s));
    if (!descents) return TLM_OK;
    tlm_pathset *ps = nullptr;
    if (int rc = alloc_pathset(ctx, n, in->n_nodes_total, &ps)) return rc;
    struct Drop {
        tlm_pathset *p;
        ~Drop() { tlm_pathset_destroy(p); }
    } drop{ps};
    if (int rc = tlm_pool_alloc(&idx.p, n * 8, s)) return rc;
    if (int rc = tlm_pool_alloc(&idx2.p, n * 8, s)) return rc;
    if (int rc = tlm_pool_alloc(&ncnt.p, n * 8, s)) return rc;
    hipLaunchKernelGGL(iota_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, n, (uint64_t *) idx.p);
    size_t tb = 0, tb2 = 0;
    TLM_HIP(rocprim::radix_sort_pairs(nullptr, tb, (unsigned long long *) in->unit, (unsigned long long *) ps->unit, (unsigned long long *) idx.p,
                                      (unsigned long long *) idx2.p, (size_t) n, 0, 64, s));
    TLM_HIP(rocprim::exclusive_scan(nullptr, tb2, (uint64_t *) ncnt.p, ps->off, (uint64_t) 0, (size_t) n, rocprim::plus<uint64_t>(), s));
    if (int rc = tlm_pool_alloc(&tmp.p, std::max<size_t>(std::max(tb, tb2), 16), s)) return rc;
    TLM_HIP(rocprim::radix_sort_pairs(tmp.p, tb, (unsigned long long *) in->unit, (unsigned long long *) ps->unit, (unsigned long long *) idx.p,
                                      (unsigned long long *) idx2.p, (size_t) n, 0, 64, s));
    hipLaunchKernelGGL(sorted_counts_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, n, (const uint64_t *) idx2.p, in->off, in->length,
                       (uint64_t *) ncnt.p, ps->length);
    TLM_HIP(rocprim::exclusive_scan(tmp.p, tb2, (uint64_t *) ncnt.p, ps->off, (uint64_t) 0, (size_t) n, rocprim::plus<uint64_t>(), s));
    hipLaunchKernelGGL(fetch_kernel, dim3(blocks_for(n, 4)), dim3(256), 0, s, n, (const uint64_t *) idx2.p, (const uint64_t *) ps->off,
                       in->off, in->nodes, in->slots, ps->nodes, ps->slots);
    TLM_HIP(hipMemcpyAsync(ps->off + n, &in->n_nodes_total, 8, hipMemcpyHostToDevice, s));
    TLM_HIP(hipStreamSynchronize(s));
    TLM_HIP(hipGetLastError());
    drop.p = in;            // the unsorted set goes, the sorted one is the result
    *out = ps;
    return TLM_OK;
}

extern "C" int tlm_pathset_filter_unique(tlm_ctx *ctx, const tlm_pathset *in, tlm_pathset **out) {
    TLM_CHECK(ctx && in && out && in->ctx == ctx, TLM_EINVAL, "tlm_pathset_filter_unique: bad argument");
    TLM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint64_t n = in->n_paths;
    Tmp keep, pidx, noff, tot;
    if (int rc = tlm_pool_alloc(&keep.p, n ? n : 1, s)) return rc;
    if (int rc = tlm_pool_alloc(&pidx.p, (n ? n : 1) * 8, s)) return rc;
    if (int rc = tlm_pool_alloc(&noff.p, (n ? n : 1) * 8, s)) return rc;
    if (int rc = tlm_pool_alloc(&tot.p, 16, s)) return rc;
    uint64_t h_tot[2] = {0, 0};
    if (n) {
        hipLaunchKernelGGL(dup_flags_kernel, dim3(blocks_for(n, 4)), dim3(256), 0, s, n, in->off, in->nodes, (uint8_t *) keep.p);
        hipLaunchKernelGGL(scan_keep_kernel, dim3(1), dim3(1024), 0, s, n, (const uint8_t *) keep.p, in->off, (uint64_t *) pidx.p,
                           (uint64_t *) noff.p, (uint64_t *) tot.p);
        TLM_HIP(hipMemcpyAsync(h_tot, tot.p, 16, hipMemcpyDeviceToHost, s));
        TLM_HIP(hipStreamSynchronize(s));
    }
    tlm_pathset *ps = nullptr;
    if (int rc = alloc_pathset(ctx, h_tot[0], h_tot[1], &ps)) return rc;
    if (n) {
        hipLaunchKernelGGL(copy_kept_kernel, dim3(blocks_for(n, 4)), dim3(256), 0, s, n, (const uint8_t *) keep.p,
                           (const uint64_t *) pidx.p, (const uint64_t *) noff.p, in->unit, in->length, in->off, in->nodes,
                           in->slots, ps->unit, ps->length, ps->off, ps->nodes, ps->slots);
    }
    TLM_HIP(hipMemcpyAsync(ps->off + ps->n_paths, &ps->n_nodes_total, 8, hipMemcpyHostToDevice, s));
    TLM_HIP(hipStreamSynchronize(s));
    TLM_HIP(hipGetLastError());
    *out = ps;
    return TLM_OK;
}

extern "C" int tlm_pathset_filter_unique_take(tlm_ctx *ctx, tlm_pathset *in, tlm_pathset **out) {
    TLM_CHECK(ctx && in && out && in->ctx == ctx, TLM_EINVAL, "tlm_pathset_filter_unique_take: bad argument");
    TLM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint64_t n = in->n_paths;
    if (n > 1) {        // is there an adjacent duplicate at all?  (usually not: then the set is handed back as it is)
        Tmp keep, cnt;
        if (int rc = tlm_pool_alloc(&keep.p, n, s)) return rc;
        if (int rc = tlm_pool_alloc(&cnt.p, 4, s)) return rc;
        TLM_HIP(hipMemsetAsync(cnt.p, 0, 4, s));
        hipLaunchKernelGGL(dup_flags_kernel, dim3(blocks_for(n, 4)), dim3(256), 0, s, n, in->off, in->nodes, (uint8_t *) keep.p);
        hipLaunchKernelGGL(count_dropped_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, n, (const uint8_t *) keep.p, (uint32_t *) cnt.p);
        uint32_t dropped = 0;
        TLM_HIP(hipMemcpyAsync(&dropped, cnt.p, 4, hipMemcpyDeviceToHost, s));
        TLM_HIP(hipStreamSynchronize(s));
        if (dropped) {
            if (int rc = tlm_pathset_filter_unique(ctx, in, out)) return rc;
            tlm_pathset_destroy(in);
            return TLM_OK;
        }
    }
    in->n_units = 0;
    in->stats.clear();
    *out = in;
    return TLM_OK;
}

extern "C" int tlm_pathset_anchor_pairs(tlm_ctx *ctx, const tlm_pathset *ps, uint32_t *order_host, uint64_t *key_host) {
    TLM_CHECK(ctx && ps && ps->ctx == ctx, TLM_EINVAL, "tlm_pathset_anchor_pairs: bad argument");
    TLM_CHECK(ps->n_paths < (1ull << 32), TLM_ELIMIT, "tlm_pathset_anchor_pairs: too many paths");
    TLM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint64_t n = ps->n_paths;
    if (!n) return TLM_OK;
    Tmp key, idx, key2, idx2, tmp;
    if (int rc = tlm_pool_alloc(&key.p, n * 8, s)) return rc;
    if (int rc = tlm_pool_alloc(&idx.p, n * 4, s)) return rc;
    if (int rc = tlm_pool_alloc(&key2.p, n * 8, s)) return rc;
    if (int rc = tlm_pool_alloc(&idx2.p, n * 4, s)) return rc;
    hipLaunchKernelGGL(pair_key_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, s, n, ps->off, ps->nodes,
                       (unsigned long long *) key.p, (uint32_t *) idx.p);
    size_t tb = 0;
    TLM_HIP(rocprim::radix_sort_pairs(nullptr, tb, (unsigned long long *) key.p, (unsigned long long *) key2.p, (uint32_t *) idx.p,
                                      (uint32_t *) idx2.p, (size_t) n, 0, 64, s));
    if (int rc = tlm_pool_alloc(&tmp.p, tb ? tb : 16, s)) return rc;
    TLM_HIP(rocprim::radix_sort_pairs(tmp.p, tb, (unsigned long long *) key.p, (unsigned long long *) key2.p, (uint32_t *) idx.p,
                                      (uint32_t *) idx2.p, (size_t) n, 0, 64, s));      // stable: paths_ order inside a bucket
    if (order_host) TLM_HIP(hipMemcpyAsync(order_host, idx2.p, n * 4, hipMemcpyDeviceToHost, s));
    if (key_host) TLM_HIP(hipMemcpyAsync(key_host, key2.p, n * 8, hipMemcpyDeviceToHost, s));
    TLM_HIP(hipStreamSynchronize(s));
    TLM_HIP(hipGetLastError());
    return TLM_OK;
}

extern "C" int tlm_pathset_summary(tlm_ctx *ctx, const tlm_pathset *ps, uint32_t *front, uint32_t *back, uint32_t *n_nodes,
                                   uint64_t *hash) {
    TLM_CHECK(ctx && ps && ps->ctx == ctx, TLM_EINVAL, "tlm_pathset_summary: bad argument");
    TLM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint64_t n = ps->n_paths;
    if (!n) return TLM_OK;
    Tmp f, b, c, h;
    if (int rc = tlm_pool_alloc(&f.p, n * 4, s)) return rc;
    if (int rc = tlm_pool_alloc(&b.p, n * 4, s)) return rc;
    if (int rc = tlm_pool_alloc(&c.p, n * 4, s)) return rc;
    if (int rc = tlm_pool_alloc(&h.p, n * 8, s)) return rc;
    hipLaunchKernelGGL(summary_kernel, dim3(blocks_for(n, 4)), dim3(256), 0, s, n, ps->off, ps->nodes, (uint32_t *) f.p,
                       (uint32_t *) b.p, (uint32_t *) c.p, (uint64_t *) h.p);
    if (front) TLM_HIP(hipMemcpyAsync(front, f.p, n * 4, hipMemcpyDeviceToHost, s));
    if (back) TLM_HIP(hipMemcpyAsync(back, b.p, n * 4, hipMemcpyDeviceToHost, s));
    if (n_nodes) TLM_HIP(hipMemcpyAsync(n_nodes, c.p, n * 4, hipMemcpyDeviceToHost, s));
    if (hash) TLM_HIP(hipMemcpyAsync(hash, h.p, n * 8, hipMemcpyDeviceToHost, s));
    TLM_HIP(hipStreamSynchronize(s));
    TLM_HIP(hipGetLastError());
    return TLM_OK;
}

extern "C" int tlm_pathset_fetch(tlm_ctx *ctx, const tlm_pathset *ps, uint64_t n, const uint64_t *index, uint32_t *nodes,
                                 uint32_t *slots) {
    TLM_CHECK(ctx && ps && ps->ctx == ctx && (n == 0 || index), TLM_EINVAL, "tlm_pathset_fetch: bad argument");
    if (!n) return TLM_OK;
    for (uint64_t k = 0; k < n; k++)
        TLM_CHECK(index[k] < ps->n_paths, TLM_EINVAL, "tlm_pathset_fetch: path %llu of %llu", (unsigned long long) index[k],
                  (unsigned long long) ps->n_paths);
    TLM_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    // sizes of the requested paths: their off[] pairs, gathered on the host from two small copies per path would be
    // slow for many paths; the whole off[] array is 8 bytes per path
    std::vector<uint64_t> h_off(ps->n_paths + 1);
    TLM_HIP(hipMemcpyAsync(h_off.data(), ps->off, (ps->n_paths + 1) * 8, hipMemcpyDeviceToHost, s));
    TLM_HIP(hipStreamSynchronize(s));
    std::vector<uint64_t> dst(n + 1, 0);
    for (uint64_t k = 0; k < n; k++) dst[k + 1] = dst[k] + (h_off[index[k] + 1] - h_off[index[k]]);
    const uint64_t total = dst[n];
    Tmp d_idx, d_dst, o_n, o_s;
    if (int rc = tlm_pool_alloc(&d_idx.p, n * 8, s)) return rc;
    if (int rc = tlm_pool_alloc(&d_dst.p, (n + 1) * 8, s)) return rc;
    if (int rc = tlm_pool_alloc(&o_n.p, (total ? total : 1) * 4, s)) return rc;
    if (int rc = tlm_pool_alloc(&o_s.p, (total ? total : 1) * 4, s)) return rc;
    TLM_HIP(hipMemcpyAsync(d_idx.p, index, n * 8, hipMemcpyHostToDevice, s));
    TLM_HIP(hipMemcpyAsync(d_dst.p, dst.data(), (n + 1) * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(fetch_kernel, dim3(blocks_for(n, 4)), dim3(256), 0, s, n, (const uint64_t *) d_idx.p,
                       (const uint64_t *) d_dst.p, ps->off, ps->nodes, ps->slots, (uint32_t *) o_n.p, (uint32_t *) o_s.p);
    if (nodes && total) TLM_HIP(hipMemcpyAsync(nodes, o_n.p, total * 4, hipMemcpyDeviceToHost, s));
    if (slots && total) TLM_HIP(hipMemcpyAsync(slots, o_s.p, total * 4, hipMemcpyDeviceToHost, s));
    TLM_HIP(hipStreamSynchronize(s));
    TLM_HIP(hipGetLastError());
    return TLM_OK;
}

// ---- internal (exchange.hip): the same on device buffers ------------------------------------------------------
// Summary columns of `ps` into DEVICE arrays of n_paths entries; dup[i] = 1 iff path i has the node sequence of path
// i - 1 (exact, dup_flags_kernel).
int tlm_pathset_summary_dev(const tlm_pathset *ps, hipStream_t s, uint32_t *front, uint32_t *back, uint32_t *n_nodes,
                            uint64_t *hash, uint8_t *dup) {
    const uint64_t n = ps->n_paths;
    if (!n) return TLM_OK;
    hipLaunchKernelGGL(summary_kernel, dim3(blocks_for(n, 4)), dim3(256), 0, s, n, ps->off, ps->nodes, front, back, n_nodes, hash);
    hipLaunchKernelGGL(dup_flags_kernel, dim3(blocks_for(n, 4)), dim3(256), 0, s, n, ps->off, ps->nodes, dup);
    hipLaunchKernelGGL(invert_flags_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, n, dup);
    TLM_HIP(hipGetLastError());
    return TLM_OK;
}

// Paths index[0..n) (DEVICE) of `ps` -> DEVICE buffers at dst_off[k] (DEVICE, n + 1 entries).
int tlm_pathset_gather_dev(const tlm_pathset *ps, hipStream_t s, uint64_t n, const uint64_t *index_dev, const uint64_t *dst_off_dev,
                           uint32_t *nodes_dev, uint32_t *slots_dev) {
    if (!n) return TLM_OK;
    hipLaunchKernelGGL(fetch_kernel, dim3(blocks_for(n, 4)), dim3(256), 0, s, n, index_dev, dst_off_dev, ps->off, ps->nodes,
                       ps->slots, nodes_dev, slots_dev);
    TLM_HIP(hipGetLastError());
    return TLM_OK;
}

// K5 — lastz `--chain` (reference call site src/mimeo/wrappers.py:1031 `--chain`).
//
// Work unit = one group = one (target scaffold, query scaffold, strand); one workgroup owns it:
//   * the HSPs of all groups are sorted by (group, tstart, qstart, length) with two stable
//     device-wide radix sorts (microsatellite-rich units reach 10^5 HSPs: no O(n^2) sort);
//   * chain DP  best[j] = score[j] + max(0, max{best[i] : i ends at or before the start of j in both
//     sequences}), evaluated forward in tiles of 64: one wavefront finalises a tile (lane = HSP,
//     64 shuffle steps), then the whole workgroup relaxes every later HSP against the tile's 64
//     final values from LDS — by one binary search in the tile's members ordered by query end (with
//     their running best) when the HSP starts behind the whole tile in the target, member by member
//     otherwise.  Tiles and the members of a tile are visited in ascending order and
//     only a strict improvement replaces a predecessor, so ties go to the earliest predecessor
//     and the earliest chain end;
//   * flag the chain and order the chained HSPs by (score desc, tstart, qstart, length) — the
//     order in which K6 turns them into anchors.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "device_util.h"

namespace mimeo {

constexpr int CH_THREADS = 1024;
constexpr int CH_TILE = 64;  // HSPs finalised per step of the chain DP (one wavefront)

__device__ __forceinline__ bool hsp_less(const mimeo_hsp &a, const mimeo_hsp &b) {
    if (a.tstart != b.tstart) return a.tstart < b.tstart;
    if (a.qstart != b.qstart) return a.qstart < b.qstart;
    return a.length < b.length;
}
// anchor order: score descending, then (tstart, qstart, length)
__device__ __forceinline__ bool anchor_less(const mimeo_hsp &a, const mimeo_hsp &b) {
    if (a.score != b.score) return a.score > b.score;
    return hsp_less(a, b);
}

// sort plumbing: key1 = (qstart, length), key2 = (group = unit of the batch, tstart); the HSPs arrive in no order,
// tagged with their unit (K4 works on the whole batch at once)
__global__ void k5_key1(const mimeo_hsp *__restrict__ in, uint64_t n, uint64_t *__restrict__ key, uint32_t *__restrict__ val) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    key[i] = ((uint64_t)in[i].qstart << 32) | in[i].length;
    val[i] = (uint32_t)i;
}
__global__ void k5_key2(const mimeo_hsp *__restrict__ in, const uint32_t *__restrict__ unit, const uint32_t *__restrict__ perm,
                        uint64_t n, uint64_t *__restrict__ key) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t src = perm[i];
    key[i] = ((uint64_t)unit[src] << 32) | in[src].tstart;
}
// the sorted keys carry the group in their high word: a group's HSP range is where that word changes (groups
// without HSPs keep the empty range the host gave them)
__global__ void k5_group_ranges(const uint64_t *__restrict__ key, uint64_t n, Group *__restrict__ groups) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = (uint32_t)(key[i] >> 32);
    if (i == 0 || (uint32_t)(key[i - 1] >> 32) != g) groups[g].hsp_begin = i;
    if (i + 1 == n || (uint32_t)(key[i + 1] >> 32) != g) groups[g].hsp_end = i + 1;
}
__global__ void k5_gather(const mimeo_hsp *__restrict__ in, const uint32_t *__restrict__ perm, uint64_t n,
                          mimeo_hsp *__restrict__ hs) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    mimeo_hsp h = in[perm[i]];
    h.flags = 0;
    hs[i] = h;
}

__global__ __launch_bounds__(CH_THREADS) void k5_chain(Group *__restrict__ groups,
                                                       mimeo_hsp *__restrict__ hs, long long *__restrict__ best,
                                                       long long *__restrict__ cand, int *__restrict__ pred,
                                                       uint32_t *__restrict__ order, int do_chain) {
    Group &G = groups[blockIdx.x];
    const uint64_t b0 = G.hsp_begin;
    const uint32_t n = (uint32_t)(G.hsp_end - G.hsp_begin);
    const uint32_t tid = threadIdx.x;
    __shared__ long long s_best[CH_THREADS / 64];
    __shared__ uint32_t s_idx[CH_THREADS / 64];
    __shared__ uint32_t s_m;
    __shared__ uint32_t s_te[CH_TILE], s_qe[CH_TILE];
    __shared__ long long s_b[CH_TILE];
    // the same tile ordered by query end, with the running best (and its tile member) over that order: an HSP that
    // starts behind every member of the tile in the target finds its best predecessor by one binary search
    __shared__ uint32_t s_sqe[CH_TILE], s_pmi[CH_TILE], s_maxte;
    __shared__ long long s_pmb[CH_TILE];
    if (n == 0) { if (tid == 0) G.nchain = 0; return; }
    // 1. hs[b0 .. b0+n) arrives sorted by (tstart, qstart, length) (device-wide radix sorts, chain_device)
    if (do_chain) {
        for (uint32_t k = tid; k < n; k += CH_THREADS) { cand[b0 + k] = 0; pred[b0 + k] = -1; }
        __syncthreads();
        for (uint32_t t0 = 0; t0 < n; t0 += CH_TILE) {
            // a. wavefront 0 finalises HSPs t0 .. t0+63: every earlier tile has already relaxed them
            if (tid < CH_TILE) {
                const uint32_t j = t0 + tid;
                const bool live = j < n;
                mimeo_hsp hj;
                hj.tstart = hj.qstart = 0xFFFFFFFFu; hj.length = 0; hj.score = 0;
                long long cj = 0;
                int pj = -1;
                if (live) { hj = hs[b0 + j]; cj = cand[b0 + j]; pj = pred[b0 + j]; }
                const uint32_t te = hj.tstart + hj.length, qe = hj.qstart + hj.length;
                const uint32_t cnt = min((uint32_t)CH_TILE, n - t0);
                for (uint32_t jj = 0; jj + 1 < cnt; jj++) {
                    const long long bj = __shfl(cj + hj.score, (int)jj);  // final: members before jj are done
                    const uint32_t tej = (uint32_t)__shfl((int)te, (int)jj), qej = (uint32_t)__shfl((int)qe, (int)jj);
                    if (live && tid > jj && tej <= hj.tstart && qej <= hj.qstart && bj > cj) { cj = bj; pj = (int)(t0 + jj); }
                }
                const long long bfin = cj + hj.score;
                if (live) {
                    best[b0 + j] = bfin;
                    pred[b0 + j] = pj;
                    s_te[tid] = te; s_qe[tid] = qe; s_b[tid] = bfin;
                }
                // bitonic sort of the 64 members by (query end, member) with shuffles; dead lanes sort to the end
                uint32_t kq = live ? qe : 0xFFFFFFFFu, ki = tid;
                long long kb = live ? bfin : INT64_MIN;
                for (uint32_t k = 2; k <= 64; k <<= 1)
                    for (uint32_t jx = k >> 1; jx > 0; jx >>= 1) {
                        const uint32_t oq = (uint32_t)__shfl_xor((int)kq, (int)jx), oi = (uint32_t)__shfl_xor((int)ki, (int)jx);
                        const long long ob = __shfl_xor(kb, (int)jx);
                        const bool up = (tid & k) == 0, lower = (tid & jx) == 0;
                        const bool mine_less = kq < oq || (kq == oq && ki < oi);
                        const bool keep = (lower == up) ? mine_less : !mine_less;   // keep the smaller in the lower lane of an ascending pair
                        if (!keep) { kq = oq; ki = oi; kb = ob; }
                    }
                // inclusive running maximum of the final values in that order; ties to the smaller member
                long long pb = kb;
                uint32_t pi = ki;
                for (int o = 1; o < 64; o <<= 1) {
                    const long long ub = __shfl_up(pb, o);
                    const uint32_t ui = (uint32_t)__shfl_up((int)pi, o);
                    if (tid >= (uint32_t)o && (ub > pb || (ub == pb && ui < pi))) { pb = ub; pi = ui; }
                }
                s_sqe[tid] = kq; s_pmb[tid] = pb; s_pmi[tid] = pi;
                uint32_t mte = live ? te : 0u;
                for (int o = 32; o > 0; o >>= 1) mte = max(mte, (uint32_t)__shfl_xor((int)mte, o));
                if (tid == 0) s_maxte = mte;
            }
            __syncthreads();
            // b. everybody relaxes the HSPs behind the tile against its final values
            const uint32_t cnt = min((uint32_t)CH_TILE, n - t0);
            for (uint32_t k = t0 + CH_TILE + tid; k < n; k += CH_THREADS) {
                const mimeo_hsp &hk = hs[b0 + k];
                const uint32_t ts = hk.tstart, qs = hk.qstart;
                long long c = cand[b0 + k];
                int pk = -2;
                if (s_maxte <= ts) {
                    // every member ends in front of this HSP in the target: the best one among those that also end in
                    // front of it in the query = running maximum at the last sorted member with query end <= qs
                    uint32_t lo = 0, hi = CH_TILE;  // number of sorted members with query end <= qs
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (s_sqe[mid] <= qs) lo = mid + 1; else hi = mid;
                    }
                    if (lo && s_pmb[lo - 1] > c) { c = s_pmb[lo - 1]; pk = (int)(t0 + s_pmi[lo - 1]); }
                } else {
                    for (uint32_t ii = 0; ii < cnt; ii++)
                        if (s_te[ii] <= ts && s_qe[ii] <= qs && s_b[ii] > c) { c = s_b[ii]; pk = (int)(t0 + ii); }
                }
                if (pk != -2) { cand[b0 + k] = c; pred[b0 + k] = pk; }
            }
            __syncthreads();
        }
        // argmax of best, earliest on ties
        long long mb = INT64_MIN;
        uint32_t mi = 0xFFFFFFFFu;
        for (uint32_t k = tid; k < n; k += CH_THREADS) {
            long long v = best[b0 + k];
            if (v > mb) { mb = v; mi = k; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            long long ob = __shfl_xor(mb, o);
            uint32_t oi = __shfl_xor(mi, o);
            if (ob > mb || (ob == mb && oi < mi)) { mb = ob; mi = oi; }
        }
        if ((tid & 63) == 0) { s_best[tid >> 6] = mb; s_idx[tid >> 6] = mi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < CH_THREADS / 64; w++)
                if (s_best[w] > mb || (s_best[w] == mb && s_idx[w] < mi)) { mb = s_best[w]; mi = s_idx[w]; }
            uint32_t m = 0;
            for (int k = (int)mi; k >= 0; k = pred[b0 + k]) { hs[b0 + k].flags = 1; m++; }
            s_m = m;
        }
        __syncthreads();
    } else {
        for (uint32_t k = tid; k < n; k += CH_THREADS) hs[b0 + k].flags = 1;
        if (tid == 0) s_m = n;
        __syncthreads();
    }
    // 2. the anchor order of the flagged HSPs is made by ONE stable device-wide sort behind this kernel (k5_rank_keys)
    if (tid == 0) G.nchain = s_m;
}

// Anchor order = (score descending, tstart, qstart, length) among the chained HSPs of a group.  The HSPs already lie in
// (group, tstart, qstart, length) order, so one STABLE radix sort by (group, not chained, -score) leaves every group's
// chained HSPs in anchor order at the front of its range (round 1 ranked each chained HSP against all HSPs of its
// group: O(n m) per group, the larger part of the minute a 150 Mbp self unit spent in K5).
__global__ void k5_rank_keys(const mimeo_hsp *__restrict__ hs, const uint64_t *__restrict__ gkey, uint64_t n, uint64_t *__restrict__ key,
                             uint32_t *__restrict__ val) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t MAXS = (1ull << 40) - 1ull;
    const int64_t sc = hs[i].score;
    const uint64_t inv = MAXS - (uint64_t)(sc < 0 ? 0 : (sc > (int64_t)MAXS ? (int64_t)MAXS : sc));
    key[i] = ((gkey[i] >> 32) << 41) | ((hs[i].flags & 1u) ? 0ull : (1ull << 40)) | inv;
    val[i] = (uint32_t)i;
}
__global__ void k5_write_order(const uint32_t *__restrict__ val, const uint64_t *__restrict__ key, const Group *__restrict__ groups,
                               uint64_t n, uint32_t *__restrict__ order) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    order[r] = val[r] - (uint32_t)groups[key[r] >> 41].hsp_begin;   // index inside the group
}

int chain_device(Group *d_groups, uint32_t ngroups, const mimeo_hsp *d_hsps, const uint32_t *d_hsp_unit, uint64_t nhsps,
                 int do_chain, mimeo_hsp *d_sorted, long long *d_best, long long *d_cand, int *d_pred, uint32_t *d_order) {
    if (!ngroups || !nhsps) return 0;
    if (nhsps >= (1ull << 32)) { set_error("more than 2^32 HSPs in one batch"); return MIMEO_ERR_LIMIT; }
    hipStream_t st = stream();
    static DeviceBuf kA, kB, vA, vB, tmp;  // K5 runs on the calling thread only
    int rc;
    if ((rc = kA.reserve(nhsps * 8)) || (rc = kB.reserve(nhsps * 8)) || (rc = vA.reserve(nhsps * 4)) ||
        (rc = vB.reserve(nhsps * 4)))
        return rc;
    const dim3 blk(256), grd((uint32_t)((nhsps + 255) / 256));
    hipLaunchKernelGGL(k5_key1, grd, blk, 0, st, d_hsps, nhsps, (uint64_t *)kA.p, (uint32_t *)vA.p);
    size_t tb = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, tb, (uint64_t *)kA.p, (uint64_t *)kB.p, (uint32_t *)vA.p, (uint32_t *)vB.p,
                                      (size_t)nhsps, 0, 64, st));
    if ((rc = tmp.reserve(tb + 16))) return rc;
    HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, (uint64_t *)kA.p, (uint64_t *)kB.p, (uint32_t *)vA.p, (uint32_t *)vB.p,
                                      (size_t)nhsps, 0, 64, st));
    hipLaunchKernelGGL(k5_key2, grd, blk, 0, st, d_hsps, d_hsp_unit, (const uint32_t *)vB.p, nhsps, (uint64_t *)kA.p);
    HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, (uint64_t *)kA.p, (uint64_t *)kB.p, (uint32_t *)vB.p, (uint32_t *)vA.p,
                                      (size_t)nhsps, 0, 64, st));
    hipLaunchKernelGGL(k5_group_ranges, grd, blk, 0, st, (const uint64_t *)kB.p, nhsps, d_groups);
    hipLaunchKernelGGL(k5_gather, grd, blk, 0, st, d_hsps, (const uint32_t *)vA.p, nhsps, d_sorted);
    hipLaunchKernelGGL(k5_chain, dim3(ngroups), dim3(CH_THREADS), 0, st, d_groups, d_sorted, d_best,
                       d_cand, d_pred, d_order, do_chain);
    // kB still holds the sorted (group, tstart) keys: anchor order by one stable sort over 54 key bits
    hipLaunchKernelGGL(k5_rank_keys, grd, blk, 0, st, (const mimeo_hsp *)d_sorted, (const uint64_t *)kB.p, nhsps, (uint64_t *)kA.p,
                       (uint32_t *)vB.p);
    HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, (uint64_t *)kA.p, (uint64_t *)kB.p, (uint32_t *)vB.p, (uint32_t *)vA.p,
                                      (size_t)nhsps, 0, 64, st));
    hipLaunchKernelGGL(k5_write_order, grd, blk, 0, st, (const uint32_t *)vA.p, (const uint64_t *)kB.p, (const Group *)d_groups, nhsps,
                       d_order);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace mimeo

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

// Summaries of a finalised tile of 64 HSPs (a slice of larger arrays in the two-level kernel)
struct TileLds {
    uint32_t *te, *qe;       // members in index order: target end, query end
    long long *b;            // ... final chain score
    uint32_t *sqe, *pmi;     // the same tile ordered by query end, with the running best (and its member) over that order: an HSP
    long long *pmb;          //   that starts behind every member in the target finds its best predecessor by one binary search
    uint32_t *maxte;
};

// a. wavefront 0 finalises HSPs t0 .. t0+63 of the group: every earlier tile has already relaxed them
__device__ __forceinline__ void chain_tile_final(const mimeo_hsp *__restrict__ hs, long long *__restrict__ best, const long long *__restrict__ cand,
                                                 int *__restrict__ pred, uint64_t b0, uint32_t n, uint32_t t0, const TileLds &L) {
    const uint32_t tid = threadIdx.x;
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
        L.te[tid] = te; L.qe[tid] = qe; L.b[tid] = bfin;
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
    L.sqe[tid] = kq; L.pmb[tid] = pb; L.pmi[tid] = pi;
    uint32_t mte = live ? te : 0u;
    for (int o = 32; o > 0; o >>= 1) mte = max(mte, (uint32_t)__shfl_xor((int)mte, o));
    if (tid == 0) *L.maxte = mte;
}

// the best predecessor of an HSP (ts, qs) among the cnt members of a finalised tile whose first member is HSP t0:
// strict improvements of (c, pk) only, members in ascending order
__device__ __forceinline__ void chain_tile_relax(const TileLds &L, uint32_t cnt, uint32_t t0, uint32_t ts, uint32_t qs, long long &c, int &pk) {
    if (*L.maxte <= ts) {
        // every member ends in front of this HSP in the target: the best one among those that also end in
        // front of it in the query = running maximum at the last sorted member with query end <= qs
        uint32_t lo = 0, hi = CH_TILE;  // number of sorted members with query end <= qs
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (L.sqe[mid] <= qs) lo = mid + 1; else hi = mid;
        }
        if (lo && L.pmb[lo - 1] > c) { c = L.pmb[lo - 1]; pk = (int)(t0 + L.pmi[lo - 1]); }
    } else {
        for (uint32_t ii = 0; ii < cnt; ii++)
            if (L.te[ii] <= ts && L.qe[ii] <= qs && L.b[ii] > c) { c = L.b[ii]; pk = (int)(t0 + ii); }
    }
}

// chain end = argmax of best (earliest on ties), flags along its predecessors; G.nchain
__device__ __forceinline__ void chain_tail(Group &G, mimeo_hsp *__restrict__ hs, const long long *__restrict__ best, const int *__restrict__ pred,
                                           uint64_t b0, uint32_t n, long long *s_best, uint32_t *s_idx, uint32_t *s_m) {
    const uint32_t tid = threadIdx.x;
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
        *s_m = m;
    }
    __syncthreads();
    // the anchor order of the flagged HSPs is made by ONE stable device-wide sort behind this kernel (k5_rank_keys)
    if (tid == 0) G.nchain = *s_m;
}

// groups beyond CH_BIG HSPs are left to k5_chain_big when skip_big is set
constexpr uint32_t CH_BIG = 32768;
__global__ __launch_bounds__(CH_THREADS) void k5_chain(Group *__restrict__ groups,
                                                       mimeo_hsp *__restrict__ hs, long long *__restrict__ best,
                                                       long long *__restrict__ cand, int *__restrict__ pred,
                                                       uint32_t *__restrict__ order, int do_chain, int skip_big) {
    Group &G = groups[blockIdx.x];
    const uint64_t b0 = G.hsp_begin;
    const uint32_t n = (uint32_t)(G.hsp_end - G.hsp_begin);
    const uint32_t tid = threadIdx.x;
    __shared__ long long s_best[CH_THREADS / 64];
    __shared__ uint32_t s_idx[CH_THREADS / 64];
    __shared__ uint32_t s_m;
    __shared__ uint32_t s_te[CH_TILE], s_qe[CH_TILE];
    __shared__ long long s_b[CH_TILE];
    __shared__ uint32_t s_sqe[CH_TILE], s_pmi[CH_TILE], s_maxte;
    __shared__ long long s_pmb[CH_TILE];
    if (n == 0) { if (tid == 0) G.nchain = 0; return; }
    if (do_chain && skip_big && n > CH_BIG) return;
    // 1. hs[b0 .. b0+n) arrives sorted by (tstart, qstart, length) (device-wide radix sorts, chain_device)
    if (do_chain) {
        const TileLds L{s_te, s_qe, s_b, s_sqe, s_pmi, s_pmb, &s_maxte};
        for (uint32_t k = tid; k < n; k += CH_THREADS) { cand[b0 + k] = 0; pred[b0 + k] = -1; }
        __syncthreads();
        for (uint32_t t0 = 0; t0 < n; t0 += CH_TILE) {
            if (tid < CH_TILE) chain_tile_final(hs, best, cand, pred, b0, n, t0, L);
            __syncthreads();
            // b. everybody relaxes the HSPs behind the tile against its final values
            const uint32_t cnt = min((uint32_t)CH_TILE, n - t0);
            for (uint32_t k = t0 + CH_TILE + tid; k < n; k += CH_THREADS) {
                const mimeo_hsp &hk = hs[b0 + k];
                long long c = cand[b0 + k];
                int pk = -2;
                chain_tile_relax(L, cnt, t0, hk.tstart, hk.qstart, c, pk);
                if (pk != -2) { cand[b0 + k] = c; pred[b0 + k] = pk; }
            }
            __syncthreads();
        }
        chain_tail(G, hs, best, pred, b0, n, s_best, s_idx, &s_m);
    } else {
        for (uint32_t k = tid; k < n; k += CH_THREADS) hs[b0 + k].flags = 1;
        if (tid == 0) G.nchain = n;
    }
    (void)order;
}

// ---- two-level chain DP for large groups ------------------------------------------------------------------------------
// k5_chain makes one pass over ALL later HSPs of the group per tile of 64: n^2 / 64 visits, each with its 40 bytes from
// global memory — 13 s for the 1.35 * 10^6 HSPs of one strand of a 150 Mbp self pair, on one CU.  Here the HSPs are
// taken in BLOCKS of 2048: inside a block the tiles relax only the rest of the block (L2-resident); the finished block is
// then summarised once — its members ordered by query end with the running best over that order, beside the per-tile
// summaries it already has — and ONE pass over the later HSPs relaxes them against the whole block: a binary search
// over 2048 members when the HSP starts behind all of them in the target, tile by tile (exactly k5_chain's step)
// otherwise.  32 times fewer passes; predecessors and ties as in k5_chain: the maximum over the eligible members of a
// block with the smallest member on ties is what its 32 tiles, visited in order with strict improvements, arrive at.
constexpr uint32_t CH_BLOCK = 2048;   // 32 tiles
constexpr size_t CH_BIG_LDS = (size_t)CH_BLOCK * 48 + 1024;
__global__ __launch_bounds__(CH_THREADS) void k5_chain_big(Group *__restrict__ groups, const uint32_t *__restrict__ big_list,
                                                           const unsigned int *__restrict__ nbig, mimeo_hsp *__restrict__ hs,
                                                           long long *__restrict__ best, long long *__restrict__ cand, int *__restrict__ pred) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    long long *B_b = reinterpret_cast<long long *>(lds), *T_pmb = B_b + CH_BLOCK, *S_pmb = T_pmb + CH_BLOCK;
    uint32_t *B_te = reinterpret_cast<uint32_t *>(S_pmb + CH_BLOCK), *B_qe = B_te + CH_BLOCK, *T_sqe = B_qe + CH_BLOCK, *T_pmi = T_sqe + CH_BLOCK,
             *S_sqe = T_pmi + CH_BLOCK, *S_pmi = S_sqe + CH_BLOCK, *T_maxte = S_pmi + CH_BLOCK;
    unsigned long long *skey = reinterpret_cast<unsigned long long *>(S_pmb);   // sort scratch: the block summary's slot until it is written
    __shared__ long long s_best[CH_THREADS / 64];
    __shared__ uint32_t s_idx[CH_THREADS / 64];
    __shared__ uint32_t s_m, s_blkmaxte;
    const uint32_t tid = threadIdx.x;
    for (uint32_t li = blockIdx.x; li < *nbig; li += gridDim.x) {
        Group &G = groups[big_list[li]];
        const uint64_t b0 = G.hsp_begin;
        const uint32_t n = (uint32_t)(G.hsp_end - G.hsp_begin);
        for (uint32_t k = tid; k < n; k += CH_THREADS) { cand[b0 + k] = 0; pred[b0 + k] = -1; }
        __syncthreads();
        for (uint32_t blk0 = 0; blk0 < n; blk0 += CH_BLOCK) {
            const uint32_t blk1 = min(n, blk0 + CH_BLOCK), cntB = blk1 - blk0;
            // 1. the block's tiles, relaxing inside the block only
            for (uint32_t t0 = blk0; t0 < blk1; t0 += CH_TILE) {
                const uint32_t off = t0 - blk0;
                const TileLds L{B_te + off, B_qe + off, B_b + off, T_sqe + off, T_pmi + off, T_pmb + off, T_maxte + off / CH_TILE};
                if (tid < CH_TILE) chain_tile_final(hs, best, cand, pred, b0, n, t0, L);
                __syncthreads();
                const uint32_t cnt = min((uint32_t)CH_TILE, n - t0);
                for (uint32_t k = t0 + CH_TILE + tid; k < blk1; k += CH_THREADS) {
                    const mimeo_hsp &hk = hs[b0 + k];
                    long long c = cand[b0 + k];
                    int pk = -2;
                    chain_tile_relax(L, cnt, t0, hk.tstart, hk.qstart, c, pk);
                    if (pk != -2) { cand[b0 + k] = c; pred[b0 + k] = pk; }
                }
                __syncthreads();
            }
            if (blk1 >= n) break;   // nothing behind the last block
            // 2. the block's summary: members by (query end, member) — bitonic sort of 2048 keys — and the running best
            for (uint32_t e = tid; e < CH_BLOCK; e += CH_THREADS)
                skey[e] = e < cntB ? (((unsigned long long)B_qe[e] << 32) | e) : ~0ull;
            __syncthreads();
            for (uint32_t k = 2; k <= CH_BLOCK; k <<= 1)
                for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                    const uint32_t i = ((tid & ~(j - 1u)) << 1) | (tid & (j - 1u)), l = i | j;
                    const unsigned long long a = skey[i], b = skey[l];
                    if ((a > b) == ((i & k) == 0)) { skey[i] = b; skey[l] = a; }
                    __syncthreads();
                }
            unsigned long long key2[2];
            for (int h = 0; h < 2; h++) key2[h] = skey[tid + h * CH_THREADS];
            uint32_t mte = 0;
            for (uint32_t e = tid; e < cntB; e += CH_THREADS) mte = max(mte, B_te[e]);
            for (int o = 32; o > 0; o >>= 1) mte = max(mte, (uint32_t)__shfl_xor((int)mte, o));
            if ((tid & 63) == 0) s_idx[tid >> 6] = mte;
            __syncthreads();   // every key is read before its slot becomes a running best
            if (tid == 0) { uint32_t m = 0; for (int w = 0; w < CH_THREADS / 64; w++) m = max(m, s_idx[w]); s_blkmaxte = m; }
            for (int h = 0; h < 2; h++) {
                const uint32_t e = tid + h * CH_THREADS, idx = (uint32_t)key2[h];
                const bool live = key2[h] != ~0ull;
                S_sqe[e] = live ? (uint32_t)(key2[h] >> 32) : 0xFFFFFFFFu;
                S_pmi[e] = live ? idx : 0xFFFFFFFFu;
                S_pmb[e] = live ? B_b[idx] : INT64_MIN;
            }
            __syncthreads();
            for (uint32_t o = 1; o < CH_BLOCK; o <<= 1) {   // inclusive running maximum, ties to the smaller member
                long long ub[2]; uint32_t ui[2];
                for (int h = 0; h < 2; h++) {
                    const uint32_t e = tid + h * CH_THREADS;
                    ub[h] = e >= o ? S_pmb[e - o] : INT64_MIN; ui[h] = e >= o ? S_pmi[e - o] : 0xFFFFFFFFu;
                }
                __syncthreads();
                for (int h = 0; h < 2; h++) {
                    const uint32_t e = tid + h * CH_THREADS;
                    if (e >= o && (ub[h] > S_pmb[e] || (ub[h] == S_pmb[e] && ui[h] < S_pmi[e]))) { S_pmb[e] = ub[h]; S_pmi[e] = ui[h]; }
                }
                __syncthreads();
            }
            // 3. one pass over the HSPs behind the block
            const uint32_t blkmaxte = s_blkmaxte, nsub = (cntB + CH_TILE - 1) / CH_TILE;
            for (uint32_t k = blk1 + tid; k < n; k += CH_THREADS) {
                const mimeo_hsp &hk = hs[b0 + k];
                const uint32_t ts = hk.tstart, qs = hk.qstart;
                long long c = cand[b0 + k];
                int pk = -2;
                if (blkmaxte <= ts) {
                    uint32_t lo = 0, hi = cntB;  // number of sorted members with query end <= qs
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (S_sqe[mid] <= qs) lo = mid + 1; else hi = mid;
                    }
                    if (lo && S_pmb[lo - 1] > c) { c = S_pmb[lo - 1]; pk = (int)(blk0 + S_pmi[lo - 1]); }
                } else {
                    for (uint32_t s = 0; s < nsub; s++) {
                        const uint32_t off = s * CH_TILE;
                        const TileLds L{B_te + off, B_qe + off, B_b + off, T_sqe + off, T_pmi + off, T_pmb + off, T_maxte + s};
                        chain_tile_relax(L, min((uint32_t)CH_TILE, cntB - off), blk0 + off, ts, qs, c, pk);
                    }
                }
                if (pk != -2) { cand[b0 + k] = c; pred[b0 + k] = pk; }
            }
            __syncthreads();
        }
        chain_tail(G, hs, best, pred, b0, n, s_best, s_idx, &s_m);
        __syncthreads();
    }
}

// the groups k5_chain_big takes
__global__ void k5_big_list(const Group *__restrict__ groups, uint32_t ngroups, uint32_t *__restrict__ list, unsigned int *__restrict__ nbig) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < ngroups && groups[g].hsp_end - groups[g].hsp_begin > CH_BIG) list[atomicAdd(nbig, 1u)] = g;
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
    // large groups: the two-level kernel (a fixed grid that loops over their list; nothing to do = a few microseconds)
    static DeviceBuf big;
    const int use_big = do_chain && !getenv("MIMEO_K5_NO_BIG");
    if (use_big) {
        if ((rc = big.reserve(((size_t)ngroups + 4) * 4))) return rc;
        unsigned int *nbig = (unsigned int *)big.p;
        uint32_t *list = (uint32_t *)big.p + 4;
        HIP_TRY(hipMemsetAsync(nbig, 0, 4, st));
        hipLaunchKernelGGL(k5_big_list, dim3((ngroups + 255) / 256), dim3(256), 0, st, (const Group *)d_groups, ngroups, list, nbig);
        static bool attr_done = false;
        if (!attr_done) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k5_chain_big), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CH_BIG_LDS));
            attr_done = true;
        }
        hipLaunchKernelGGL(k5_chain_big, dim3(64), dim3(CH_THREADS), CH_BIG_LDS, st, d_groups, (const uint32_t *)list, (const unsigned int *)nbig,
                           d_sorted, d_best, d_cand, d_pred);
    }
    hipLaunchKernelGGL(k5_chain, dim3(ngroups), dim3(CH_THREADS), 0, st, d_groups, d_sorted, d_best,
                       d_cand, d_pred, d_order, do_chain, use_big);
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

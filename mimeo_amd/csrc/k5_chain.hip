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
    for (uint32_t k = tid; k < n; k += blockDim.x) {
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
        for (int w = 1; w < (int)(blockDim.x / 64); w++)
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
    if (do_chain && skip_big && n > (uint32_t)skip_big) return;   // skip_big = the size beyond which the large-group kernels take a group
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

// the large groups: list 0 = the wave kernel's (fewer than 2^24 HSPs), list 1 = k5_chain_big's
constexpr uint32_t CW_MAX = 1u << 24;
__global__ void k5_big_list(const Group *__restrict__ groups, uint32_t ngroups, uint32_t big_min, int all_old, uint32_t *__restrict__ list_wave,
                            uint32_t *__restrict__ list_old, unsigned int *__restrict__ cnt) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    const uint64_t n = groups[g].hsp_end - groups[g].hsp_begin;
    if (n <= big_min) return;
    if (n < CW_MAX && !all_old) list_wave[atomicAdd(cnt, 1u)] = g;
    else list_old[atomicAdd(cnt + 1, 1u)] = g;
}

// ---- wave chain DP for large groups -----------------------------------------------------------------------------------
// k5_chain_big still walks a group tile by tile (64 HSPs per step, each step two workgroup barriers and a pass over the rest
// of its block) and block by block (a sort of 2048 keys and a pass over every later HSP): 147 ms for the 2.5 * 10^5 HSPs
// of one 10 Mbp x 10 Mbp unit with 1 % microsatellites, on one CU.  Here the predecessor query is what it is — a
// dominance maximum, "best chain among the HSPs that end at or before (ts, qs) in both sequences" — answered from a
// Fenwick tree (maximum) over the group's HSPs ranked by query end, into which an HSP is put once the sweep over the
// target has passed its end:
//   * the HSPs are in (tstart, qstart, length) order; E = the same HSPs ordered by target end; tcnt[k] = how many end
//     at or before tstart[k]; qcnt[k] = how many query ends are <= qstart[k]; qpos[i] = rank of i by query end;
//   * a step takes the HSPs [a, b).  Before it, E[.. tcnt[a]) are in the tree (all of them are final: they end before
//     a starts).  A WAVE is the longest range in which no HSP can precede another and no HSP outside ends inside:
//     every tstart below the smallest target end above tstart[a].  Its members are finalised in parallel, each by
//     one tree query (independent loads: the node addresses follow from qcnt alone).  A wave shorter than 64 is taken
//     as a TILE of 64 by one wavefront: tree query, then the HSPs that end inside the tile's target range (E[tcnt[a]
//     .. tcnt[k]), earlier members only), then the members of the tile in front of it, lane by lane as k5_chain does;
//   * tree nodes and candidates are (chain score << 24 | 2^24-1 - member): the maximum is the best chain and, on
//     ties, the earliest member — the predecessor k5_chain's ascending visits with strict improvements arrive at.
//     Insertion is atomicMax, order-free.  A chain scores less than 100 * 2^32 < 2^39 (its HSPs do not overlap).
// Steps = about n / 64 (a microsatellite rectangle alternates starts and ends every motif length), 4 us each.
struct WaveStep { uint32_t tcnt, end; };   // end: bit 31 = a wave (no member precedes another)
constexpr uint32_t CW_THREADS = 512, CW_PURE = 0x80000000u;

__global__ void k5w_end_keys(const mimeo_hsp *__restrict__ hs, const uint64_t *__restrict__ gkey, uint64_t n, int query, uint64_t *__restrict__ key,
                             uint32_t *__restrict__ val) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const mimeo_hsp &h = hs[i];
    key[i] = (gkey[i] & 0xFFFFFFFF00000000ull) | (uint64_t)((query ? h.qstart : h.tstart) + h.length);
    val[i] = (uint32_t)i;
}
__device__ __forceinline__ uint64_t upper_bound64(const uint64_t *__restrict__ a, uint64_t lo, uint64_t hi, uint64_t key) {
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (a[mid] <= key) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ __forceinline__ uint64_t lower_bound64(const uint64_t *__restrict__ a, uint64_t lo, uint64_t hi, uint64_t key) {
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}
// per HSP of a large group: its rank by query end (from the sorted order), the two counts and the step that begins at it
__global__ void k5w_prepare(const Group *__restrict__ groups, const mimeo_hsp *__restrict__ hs, const uint64_t *__restrict__ gkey,
                            const uint64_t *__restrict__ kte, const uint64_t *__restrict__ kqe, const uint32_t *__restrict__ vq, uint64_t n,
                            uint32_t big_min, uint32_t *__restrict__ qpos, uint32_t *__restrict__ qcnt, WaveStep *__restrict__ step) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    {   // i as a position of the query-end order
        const Group &G = groups[kqe[i] >> 32];
        if (G.hsp_end - G.hsp_begin > big_min) qpos[vq[i]] = (uint32_t)(i - G.hsp_begin) + 1u;
    }
    const uint64_t gk = gkey[i] & 0xFFFFFFFF00000000ull;
    const Group &G = groups[gk >> 32];
    const uint64_t b0 = G.hsp_begin, e0 = G.hsp_end;
    const uint32_t ng = (uint32_t)(e0 - b0), a = (uint32_t)(i - b0);
    if (ng <= big_min) return;
    const mimeo_hsp &h = hs[i];
    const uint32_t tc = (uint32_t)(upper_bound64(kte, b0, e0, gk | h.tstart) - b0);   // < ng: the HSP itself ends behind its start
    qcnt[i] = (uint32_t)(upper_bound64(kqe, b0, e0, gk | h.qstart) - b0);
    const uint64_t next_end = gk | (uint32_t)kte[b0 + tc];
    const uint32_t wave_end = (uint32_t)(lower_bound64(gkey, i, e0, next_end) - b0);   // first HSP that starts at or behind that end
    const uint32_t tile_end = min(a + (uint32_t)CH_TILE, ng);
    step[i] = WaveStep{tc, wave_end >= tile_end ? (wave_end | CW_PURE) : tile_end};
}

// development statistics (MIMEO_K5_STATS): one thread per large group walks its steps
__global__ void k5w_stats(const Group *__restrict__ groups, const uint32_t *__restrict__ list, uint32_t nlist, const WaveStep *__restrict__ step,
                          const mimeo_hsp *__restrict__ hs, unsigned long long *__restrict__ out) {
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= nlist) return;
    const Group &G = groups[list[li]];
    const uint64_t b0 = G.hsp_begin;
    const uint32_t n = (uint32_t)(G.hsp_end - G.hsp_begin);
    unsigned long long tiles = 0, waves = 0, wave_members = 0, window = 0, chunks = 0, maxwin = 0;
    uint32_t a = 0, c = 0;
    while (a < n) {
        const WaveStep s = step[b0 + a];
        const uint32_t b = s.end & ~CW_PURE;
        if (s.end & CW_PURE) { waves++; wave_members += b - a; }
        else {
            tiles++;
            const uint32_t w = step[b0 + b - 1].tcnt - c;
            window += w; chunks += (w + 63) / 64; maxwin = w > maxwin ? w : maxwin;
        }
        c = s.tcnt; a = b;
    }
    // what if a step took every HSP up to the first one that a member can precede (512 at most)?
    unsigned long long runs = 0, runwin = 0;
    for (uint32_t a2 = 0, c2 = 0; a2 < n;) {
        uint32_t mint = 0xFFFFFFFFu, k = a2;
        while (k < n && k - a2 < 512u && hs[b0 + k].tstart < mint) { mint = min(mint, hs[b0 + k].tstart + hs[b0 + k].length); k++; }
        runs++;
        runwin += step[b0 + k - 1].tcnt - c2;
        c2 = step[b0 + a2].tcnt; a2 = k;
    }
    unsigned long long *o = out + (size_t)li * 8;
    o[0] = n; o[1] = tiles; o[2] = waves; o[3] = wave_members; o[4] = window; o[5] = chunks; o[6] = maxwin; o[7] = (runs << 32) | runwin;
}

__device__ __forceinline__ unsigned long long cw_pack(long long best, uint32_t member) {   // 0 = no predecessor (only a positive chain is one)
    return best > 0 ? (((unsigned long long)best << 24) | (unsigned long long)(CW_MAX - 1u - member)) : 0ull;
}
// The tree: ranks in blocks of 2^bshift (1024 at least, 4096 blocks at most).  Inside a block a Fenwick tree (maximum) in
// global memory, changed by L2 atomics; over the blocks a Fenwick tree in LDS.  (One tree over all ranks sends every
// insertion through the same few top nodes: 3 * 10^5 atomics on one address, one after the other.)
// Query = prefix maximum over the first cnt ranks.  The node addresses follow from cnt alone, so all loads are issued before
// the first is waited for (a loop over the set bits pays one memory round trip per node; so do atomic loads, which the
// compiler waits for one by one): plain loads, CW_BITS of them whatever cnt is, and the caller invalidates the vector L1 (an
// agent-scope acquire fence) behind the barrier that ends a step.
constexpr int CW_BITS = 12;
constexpr uint32_t CW_BLOCKS = 4096;
struct WaveTree {
    unsigned long long *fen;    // the group's nodes: block j at fen[j << bshift ..]
    unsigned long long *blk;    // LDS: node j (1-based) at blk[j - 1]
    uint32_t n, bshift, nblk;
};
struct WaveQuery { unsigned long long v[CW_BITS], w[CW_BITS]; };
__device__ __forceinline__ void cw_query_issue(const WaveTree &T, uint32_t cnt, WaveQuery &Q) {
    const uint32_t q = cnt >> T.bshift, base = q << T.bshift;
    uint32_t x = cnt - base, j = q;
#pragma unroll
    for (int it = 0; it < CW_BITS; it++) {   // straight-line code: the loads in flight together
        const unsigned long long ld = T.fen[x ? base + x - 1u : 0u];
        Q.v[it] = x ? ld : 0ull;
        x &= x - 1u;
    }
#pragma unroll
    for (int it = 0; it < CW_BITS; it++) {
        const unsigned long long ld = T.blk[j ? j - 1u : 0u];
        Q.w[it] = j ? ld : 0ull;
        j &= j - 1u;
    }
}
__device__ __forceinline__ unsigned long long cw_query_reduce(WaveQuery &Q) {
#pragma unroll
    for (int it = 0; it < CW_BITS; it++) Q.v[it] = Q.w[it] > Q.v[it] ? Q.w[it] : Q.v[it];
#pragma unroll
    for (int st = 1; st < CW_BITS; st <<= 1)
#pragma unroll
        for (int it = 0; it + st < CW_BITS; it += 2 * st) Q.v[it] = Q.v[it + st] > Q.v[it] ? Q.v[it + st] : Q.v[it];
    return Q.v[0];
}
__device__ __forceinline__ unsigned long long cw_query(const WaveTree &T, uint32_t cnt) {
    WaveQuery Q;
    cw_query_issue(T, cnt, Q);
    return cw_query_reduce(Q);
}
__device__ __forceinline__ void cw_insert(const WaveTree &T, uint32_t pos, unsigned long long v) {   // pos: rank, 1-based
    if (!v) return;
    const uint32_t r0 = pos - 1u, b = r0 >> T.bshift, base = b << T.bshift, bsz = 1u << T.bshift;
    for (uint32_t x = (r0 - base) + 1u; x <= bsz && base + x <= T.n; x += x & (0u - x))
        __hip_atomic_fetch_max(T.fen + (base + x - 1u), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t j = b + 1u; j <= T.nblk; j += j & (0u - j))
        __hip_atomic_fetch_max(T.blk + (j - 1u), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {   // lane: wave-uniform
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}

// the HSPs in target-end order, as the sweep needs them: (target end, query end, rank by query end); their chain values
// (packed with the member, 0 = not final yet) lie beside them in the same order, written when a member is finalised, so
// that the window and the insertions read two sequential streams
__global__ void k5w_end_records(const Group *__restrict__ groups, const mimeo_hsp *__restrict__ hs, const uint64_t *__restrict__ kte, const uint32_t *__restrict__ E,
                                const uint32_t *__restrict__ qpos, uint64_t n, uint32_t big_min, uint4 *__restrict__ erec, uint32_t *__restrict__ epos) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    const Group &G = groups[kte[x] >> 32];
    if (G.hsp_end - G.hsp_begin <= big_min) return;
    const uint32_t i = E[x];
    const mimeo_hsp &h = hs[i];
    erec[x] = make_uint4((uint32_t)kte[x], h.qstart + h.length, qpos[i], 0u);
    epos[i] = (uint32_t)x;
}

__global__ __launch_bounds__(CW_THREADS) void k5_chain_wave(Group *__restrict__ groups, const uint32_t *__restrict__ list, mimeo_hsp *__restrict__ hs,
                                                            long long *__restrict__ best, unsigned long long *__restrict__ fen_all,
                                                            int *__restrict__ pred, const uint4 *__restrict__ erec_all, unsigned long long *__restrict__ beste_all,
                                                            const uint32_t *__restrict__ epos, const uint32_t *__restrict__ qcnt,
                                                            const WaveStep *__restrict__ step, unsigned long long *__restrict__ dbg) {
    constexpr uint32_t NW = CW_THREADS / 64;
    unsigned long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;   // development timing (MIMEO_K5_STATS): wavefront 0, 10 ns ticks
#define CW_TICK(slot) do { if (dbg) { const unsigned long long t1 = wall_clock64(); tm[slot] += t1 - t0; t0 = t1; } } while (0)
    __shared__ long long s_best[NW];
    __shared__ uint32_t s_idx[NW];
    __shared__ uint32_t s_m;
    __shared__ unsigned long long s_part[NW][CH_TILE];
    __shared__ unsigned long long s_blk[CW_BLOCKS];
    Group &G = groups[list[blockIdx.x]];
    const uint64_t b0 = G.hsp_begin;
    const uint32_t n = (uint32_t)(G.hsp_end - G.hsp_begin), tid = threadIdx.x, lane = tid & 63u, wv_no = tid >> 6;
    unsigned long long *fen = fen_all + b0, *beste = beste_all + b0;
    const uint4 *erec = erec_all + b0;
    uint32_t bshift = 10;
    while (((n - 1u) >> bshift) >= CW_BLOCKS) bshift++;   // n < 2^24: 12 at most
    const WaveTree T{fen, s_blk, n, bshift, ((n - 1u) >> bshift) + 1u};
    for (uint32_t k = tid; k < n; k += CW_THREADS) { fen[k] = 0ull; beste[k] = 0ull; }
    for (uint32_t k = tid; k < CW_BLOCKS; k += CW_THREADS) s_blk[k] = 0ull;
    __threadfence();
    __syncthreads();
    uint32_t a = 0, c = 0;   // E[.. c) are in the tree, visibly
    WaveStep s = step[b0];
    // every wavefront keeps the next tile's members in its registers (lane = member): fetched a step ahead
    mimeo_hsp hn;
    uint32_t qn = 0, cn = 0, en = 0;
    hn.tstart = hn.qstart = 0xFFFFFFFFu; hn.length = 0; hn.score = 0;
    if (lane < n) { hn = hs[b0 + lane]; qn = qcnt[b0 + lane]; cn = step[b0 + lane].tcnt; en = epos[b0 + lane]; }
    if (dbg) t0 = wall_clock64();
    while (a < n) {
        const uint32_t ca = s.tcnt, b = s.end & ~CW_PURE;
        const bool pure = (s.end & CW_PURE) != 0;
        CW_TICK(0);   // waiting for the header
        if (b < n) s = step[b0 + b];   // the next step's header: in flight during this one
        const mimeo_hsp hk = hn;
        const uint32_t qk = qn, ck = cn, ek = en;
        // the members of the tile behind this step: fetched now, but behind the loads this step waits for (loads return in order)
#define CW_PREFETCH() do { hn.tstart = hn.qstart = 0xFFFFFFFFu; hn.length = 0; hn.score = 0; \
        if (b + lane < n) { hn = hs[b0 + b + lane]; qn = qcnt[b0 + b + lane]; cn = step[b0 + b + lane].tcnt; en = epos[b0 + b + lane]; } } while (0)
        if (pure) {
            CW_PREFETCH();
            // a wave: every HSP that ended at or before its first start goes into the tree, then one query per member
            for (uint32_t x = c + tid; x < ca; x += CW_THREADS) cw_insert(T, erec[x].z, beste[x]);
            __threadfence();
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (uint32_t k = a + tid; k < b; k += CW_THREADS) {
                const unsigned long long m = cw_query(T, qcnt[b0 + k]);
                const long long bk = (long long)(m >> 24) + hs[b0 + k].score;
                best[b0 + k] = bk;
                pred[b0 + k] = m ? (int)(CW_MAX - 1u - (uint32_t)(m & (CW_MAX - 1u))) : -1;
                beste_all[epos[b0 + k]] = cw_pack(bk, k);
            }
            CW_TICK(1);   // a wave, up to its last barrier
        } else {
            // a tile of up to 64 members.  Window = E[c .. tcnt of the last member): the HSPs that are not in the tree for
            // sure.  It is dealt to the wavefronts 1 ..; a share is two coalesced loads, every lane tests the entries
            // that can matter against its own member.  The entries in front of ca — ends at or before the
            // tile's first start, final since the step before — go into the tree on the way: legal predecessors of every
            // member in the target, tested here as well, so it does not matter when a query sees them.
            const uint32_t k = a + lane, cnt = b - a;
            const bool live = k < b;
            const uint32_t ts_last = (uint32_t)__builtin_amdgcn_readlane((int)hk.tstart, (int)(cnt - 1u));
            const uint32_t cend = (uint32_t)__builtin_amdgcn_readlane((int)ck, (int)(cnt - 1u));
            unsigned long long m = 0;
            if (wv_no == 0) {
                WaveQuery Q;
                cw_query_issue(T, live ? qk : 0u, Q);   // wavefront 0 asks the tree
                CW_PREFETCH();
                m = cw_query_reduce(Q);
                CW_TICK(2);
            } else {
                bool fetched = false;
                if (cend > c) {
                    // the window in equal shares (8 entries at least) for the other wavefronts
                    constexpr uint32_t H = NW - 1u;
                    const uint32_t per = max((cend - c + H - 1u) / H, 8u), w_begin = c + (wv_no - 1u) * per, w_end = min(cend, w_begin + per);
                    for (uint32_t w0 = w_begin; w0 < w_end; w0 += (uint32_t)CH_TILE) {
                        uint4 r = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
                        unsigned long long v = 0;
                        if (w0 + lane < w_end) { r = erec[w0 + lane]; v = beste[w0 + lane]; }
                        if (!fetched) { CW_PREFETCH(); fetched = true; }
                        if (w0 + lane < min(w_end, ca)) cw_insert(T, r.z, v);
                        const bool use = v != 0ull && r.x <= ts_last;   // 0: a member of this tile (not final)
                        for (unsigned long long mask = __ballot(use); mask; mask &= mask - 1ull) {
                            const int jj = __ffsll((long long)mask) - 1;
                            const unsigned long long vj = readlane64(v, jj);
                            const uint32_t tej = (uint32_t)__builtin_amdgcn_readlane((int)r.x, jj), qej = (uint32_t)__builtin_amdgcn_readlane((int)r.y, jj);
                            if (tej <= hk.tstart && qej <= hk.qstart) m = vj > m ? vj : m;
                        }
                    }
                }
                if (!fetched) CW_PREFETCH();
            }
            CW_TICK(3);   // the tests of the window's entries
            if (wv_no) s_part[wv_no][lane] = m;
            // LDS only: the insertions may still be on their way (they are waited for at the end of the step)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            CW_TICK(4);   // waiting for the other wavefronts' windows
            if (wv_no == 0) {
                for (uint32_t w = 1; w < NW; w++) { const unsigned long long o = s_part[w][lane]; m = o > m ? o : m; }
                // members of the tile in front of a member, in ascending order (a member is final once those before it are done)
                const uint32_t te = hk.tstart + hk.length, qe = hk.qstart + hk.length;
                for (unsigned long long mask = __ballot(live && lane + 1u < cnt && te <= ts_last); mask; mask &= mask - 1ull) {
                    const int jj = __ffsll((long long)mask) - 1;
                    const long long bj = (long long)readlane64((unsigned long long)((long long)(m >> 24) + hk.score), jj);
                    const uint32_t tej = (uint32_t)__builtin_amdgcn_readlane((int)te, jj), qej = (uint32_t)__builtin_amdgcn_readlane((int)qe, jj);
                    if (live && lane > (uint32_t)jj && tej <= hk.tstart && qej <= hk.qstart) {
                        const unsigned long long v = cw_pack(bj, a + (uint32_t)jj);
                        m = v > m ? v : m;
                    }
                }
                if (live) {
                    const long long bk = (long long)(m >> 24) + hk.score;
                    best[b0 + k] = bk;
                    pred[b0 + k] = m ? (int)(CW_MAX - 1u - (uint32_t)(m & (CW_MAX - 1u))) : -1;
                    beste_all[ek] = cw_pack(bk, k);
                }
            }
            CW_TICK(5);   // members, stores issued
            __threadfence();
            CW_TICK(6);   // stores and insertions done
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the vector L1 may hold tree nodes from before this step's insertions
        CW_TICK(7);   // the barrier that ends the step
        c = ca;
        a = b;
    }
    if (dbg && tid == 0) for (int q = 0; q < 8; q++) dbg[(size_t)blockIdx.x * 8 + q] = tm[q];
#undef CW_TICK
#undef CW_PREFETCH
    chain_tail(G, hs, best, pred, b0, n, s_best, s_idx, &s_m);
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
    // large groups: the wave kernel (k5_chain_big beyond 2^24 HSPs, or all of them under MIMEO_K5_BIG=old)
    static DeviceBuf big, wK1, wK2, wE, wVq, wQpos, wQcnt, wStep, wErec, wEpos;
    const int use_big = do_chain && !getenv("MIMEO_K5_NO_BIG");
    if (use_big) {
        const char *mode = getenv("MIMEO_K5_BIG"), *bm = getenv("MIMEO_K5_BIG_MIN");
        const uint32_t big_min = bm ? (uint32_t)atoi(bm) : CH_BIG;
        if ((rc = big.reserve(((size_t)2 * ngroups + 4) * 4))) return rc;
        unsigned int *cnt = (unsigned int *)big.p;
        uint32_t *list_wave = (uint32_t *)big.p + 4, *list_old = list_wave + ngroups;
        HIP_TRY(hipMemsetAsync(cnt, 0, 8, st));
        hipLaunchKernelGGL(k5_big_list, dim3((ngroups + 255) / 256), dim3(256), 0, st, (const Group *)d_groups, ngroups, big_min,
                           (mode && !strcmp(mode, "old")) ? 1 : 0, list_wave, list_old, cnt);
        unsigned int h[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(h, cnt, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (h[1]) {
            static bool attr_done = false;
            if (!attr_done) {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k5_chain_big), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CH_BIG_LDS));
                attr_done = true;
            }
            hipLaunchKernelGGL(k5_chain_big, dim3(std::min(h[1], 256u)), dim3(CH_THREADS), CH_BIG_LDS, st, d_groups, (const uint32_t *)list_old,
                               (const unsigned int *)(cnt + 1), d_sorted, d_best, d_cand, d_pred);
        }
        if (h[0]) {
            if ((rc = wK1.reserve(nhsps * 8)) || (rc = wK2.reserve(nhsps * 8)) || (rc = wE.reserve(nhsps * 4)) || (rc = wVq.reserve(nhsps * 4)) ||
                (rc = wQpos.reserve(nhsps * 4)) || (rc = wQcnt.reserve(nhsps * 4)) || (rc = wStep.reserve(nhsps * sizeof(WaveStep))) ||
                (rc = wErec.reserve(nhsps * sizeof(uint4))) || (rc = wEpos.reserve(nhsps * 4)))
                return rc;
            int gbits = 1;
            while (gbits < 32 && (1ull << gbits) < (uint64_t)ngroups) gbits++;
            // kA / vB are free between the third sort and k5_rank_keys; kB holds the (group, tstart) keys of the sorted HSPs
            hipLaunchKernelGGL(k5w_end_keys, grd, blk, 0, st, (const mimeo_hsp *)d_sorted, (const uint64_t *)kB.p, nhsps, 0, (uint64_t *)kA.p, (uint32_t *)vB.p);
            HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, (uint64_t *)kA.p, (uint64_t *)wK1.p, (uint32_t *)vB.p, (uint32_t *)wE.p, (size_t)nhsps, 0,
                                              32 + gbits, st));
            hipLaunchKernelGGL(k5w_end_keys, grd, blk, 0, st, (const mimeo_hsp *)d_sorted, (const uint64_t *)kB.p, nhsps, 1, (uint64_t *)kA.p, (uint32_t *)vB.p);
            HIP_TRY(rocprim::radix_sort_pairs(tmp.p, tb, (uint64_t *)kA.p, (uint64_t *)wK2.p, (uint32_t *)vB.p, (uint32_t *)wVq.p, (size_t)nhsps, 0,
                                              32 + gbits, st));
            hipLaunchKernelGGL(k5w_prepare, grd, blk, 0, st, (const Group *)d_groups, (const mimeo_hsp *)d_sorted, (const uint64_t *)kB.p,
                               (const uint64_t *)wK1.p, (const uint64_t *)wK2.p, (const uint32_t *)wVq.p, nhsps, big_min, (uint32_t *)wQpos.p,
                               (uint32_t *)wQcnt.p, (WaveStep *)wStep.p);
            if (getenv("MIMEO_K5_STATS")) {
                DeviceBuf sb;
                if ((rc = sb.reserve((size_t)h[0] * 64))) return rc;
                hipLaunchKernelGGL(k5w_stats, dim3((h[0] + 63) / 64), dim3(64), 0, st, (const Group *)d_groups, (const uint32_t *)list_wave, h[0],
                                   (const WaveStep *)wStep.p, (const mimeo_hsp *)d_sorted, (unsigned long long *)sb.p);
                std::vector<unsigned long long> hv((size_t)h[0] * 8);
                HIP_TRY(hipStreamSynchronize(st));
                HIP_TRY(hipMemcpy(hv.data(), sb.p, hv.size() * 8, hipMemcpyDeviceToHost));
                for (unsigned int g = 0; g < h[0]; g++)
                    fprintf(stderr, "[k5w] group %u: n %llu tiles %llu waves %llu (members %llu) window entries %llu chunks %llu widest %llu; as runs of <= 512: %llu steps, window entries %llu\n", g, hv[g * 8], hv[g * 8 + 1],
                            hv[g * 8 + 2], hv[g * 8 + 3], hv[g * 8 + 4], hv[g * 8 + 5], hv[g * 8 + 6], hv[g * 8 + 7] >> 32, hv[g * 8 + 7] & 0xFFFFFFFFull);
                sb.release();
            }
            static DeviceBuf wDbg;
            const bool tstats = getenv("MIMEO_K5_STATS") != nullptr;
            if (tstats && (rc = wDbg.reserve((size_t)h[0] * 64))) return rc;
            hipLaunchKernelGGL(k5w_end_records, grd, blk, 0, st, (const Group *)d_groups, (const mimeo_hsp *)d_sorted, (const uint64_t *)wK1.p,
                               (const uint32_t *)wE.p, (const uint32_t *)wQpos.p, nhsps, big_min, (uint4 *)wErec.p, (uint32_t *)wEpos.p);
            // the sorted target-end keys are done with: their buffer holds the chain values in target-end order
            hipLaunchKernelGGL(k5_chain_wave, dim3(h[0]), dim3(CW_THREADS), 0, st, d_groups, (const uint32_t *)list_wave, d_sorted, d_best,
                               (unsigned long long *)d_cand, d_pred, (const uint4 *)wErec.p, (unsigned long long *)wK1.p, (const uint32_t *)wEpos.p,
                               (const uint32_t *)wQcnt.p, (const WaveStep *)wStep.p, tstats ? (unsigned long long *)wDbg.p : nullptr);
            if (tstats) {
                std::vector<unsigned long long> hv((size_t)h[0] * 8);
                HIP_TRY(hipStreamSynchronize(st));
                HIP_TRY(hipMemcpy(hv.data(), wDbg.p, hv.size() * 8, hipMemcpyDeviceToHost));
                for (unsigned int g = 0; g < h[0]; g++)
                    fprintf(stderr, "[k5w] group %u us: header %.0f waves %.0f piece+query loads %.0f window tests %.0f other windows %.0f members %.0f fence %.0f end barrier %.0f\n", g,
                            hv[g * 8] * 0.01, hv[g * 8 + 1] * 0.01, hv[g * 8 + 2] * 0.01, hv[g * 8 + 3] * 0.01, hv[g * 8 + 4] * 0.01, hv[g * 8 + 5] * 0.01,
                            hv[g * 8 + 6] * 0.01, hv[g * 8 + 7] * 0.01);
            }
        }
    }
    hipLaunchKernelGGL(k5_chain, dim3(ngroups), dim3(CH_THREADS), 0, st, d_groups, d_sorted, d_best,
                       d_cand, d_pred, d_order, do_chain, use_big ? (int)(getenv("MIMEO_K5_BIG_MIN") ? atoi(getenv("MIMEO_K5_BIG_MIN")) : CH_BIG) : 0);
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

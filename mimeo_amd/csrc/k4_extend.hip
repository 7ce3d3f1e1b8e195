// K4 — gap-free x-drop extension of seed hits into HSPs with lastz's per-diagonal "already
// extended" suppression, --entropy and --hspthresh (reference call site
// src/mimeo/wrappers.py:1031-1032 `--gfextend --entropy --hspthresh=H`).
//
// lastz resolves the suppression sequentially (one diagonal-extent array updated while the
// query is scanned).  The same result is obtained here without ordering the hit flood
// (DESIGN.md §4, K4):
//   * heavy kernel, one launch per unit — K34 (k34_fused.hip: seed scan, pre-filter and exact walks in one
//     kernel, no hit array) or, for A/B checks, k4_extend_hits below on the hit array of the stand-alone K3 join.
//     A hit is walked left from its seed end; the walk also looks, bit-parallel, for an earlier seed hit on its
//     own diagonal whose seed end lies at a position it reached.  If there is none the hit is a HEAD: no earlier
//     extension can cover it, so it is certainly extended by the sequential process; the lane finishes the
//     extension and emits a candidate HSP if it scores >= hspthresh.  Otherwise the hit is a FOLLOWER and only
//     (unit, diagonal, seed end, nearest earlier seed end) is kept.
//   * ONCE PER BATCH of units (the reference's per-pair loop, wrappers.py:1015-1059, issues these per pair):
//     walks that outlive the frame (k4_extend_generic, then k4_extend_long: one wavefront per hit, wave-level
//     prefix sums); the followers of ALL units are radix-sorted by (unit, diagonal, seed end) in one sort; a run in
//     which each names its predecessor is a segment owned by the head in front of it; k4_resolve_small /
//     k4_resolve_segments replay lastz's rule inside each segment (extend, skip everything whose seed end <=
//     reach, extend the next one, ...); k4_entropy applies the entropy adjustment and the threshold.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>

#include <rocprim/rocprim.hpp>

#include "k4_device.h"

namespace mimeo {

// ---- walks that outlive the frame of the heavy kernel: one lane per hit, windows loaded on demand -------------
__global__ __launch_bounds__(EXT_THREADS) void k4_extend_generic(const UnitDesc *__restrict__ units, ExtQueues q, int xdrop,
                                                              int hspthresh, int transitions,
                                                              const uint32_t *__restrict__ group_tab) {
    __shared__ uint32_t tab[GROUP_TAB];
    for (int i = threadIdx.x; i < GROUP_TAB; i += EXT_THREADS) tab[i] = group_tab[i];
    __syncthreads();
    // the hits come in eight shards (counts produced by the heavy kernels): sh_end[r] = hits in shards 0 .. r
    uint64_t sh_end[8];
    {
        uint64_t acc = 0;
#pragma unroll
        for (int r = 0; r < 8; r++) { acc += min((uint64_t)q.ctr->nmed8[r], q.med_cap); sh_end[r] = acc; }
    }
    const uint64_t nhits = sh_end[7];
    // What a hit leaves — a follower record, a place in the long queue or a candidate — waits in LDS until the wavefront has 64
    // of a kind: appended lane by lane where the walks end, these were three same-address atomics per wavefront and step
    // (~13 ns each: the duration of this kernel on a batch with 3 * 10^7 such hits)
    constexpr uint32_t GCAP = 64;
    __shared__ uint64_t s_fk[EXT_THREADS / 64][GCAP];
    __shared__ uint32_t s_fp[EXT_THREADS / 64][GCAP];
    __shared__ uint2 s_lh[EXT_THREADS / 64][GCAP];
    __shared__ uint32_t s_lu[EXT_THREADS / 64][GCAP];
    __shared__ Cand s_cd[EXT_THREADS / 64][GCAP];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t n_fol = 0, n_long = 0, n_cd = 0;   // wave-uniform
    auto flush_fol = [&]() {
        __builtin_amdgcn_wave_barrier();
        unsigned long long b = 0;
        if (lane == 0) b = atomicAdd(&q.ctr->nfollow8[queue_shard()], (unsigned long long)n_fol);
        b = __shfl(b, 0) + lane;
        if (lane < n_fol && b < q.follow_cap) { const size_t at = (size_t)queue_shard() * q.follow_cap + b; q.fkey[at] = s_fk[wv][lane]; q.fprev[at] = s_fp[wv][lane]; }
        n_fol = 0;
        __builtin_amdgcn_wave_barrier();
    };
    auto flush_long = [&]() {
        __builtin_amdgcn_wave_barrier();
        unsigned long long b = 0;
        if (lane == 0) b = atomicAdd(&q.ctr->nlong, (unsigned long long)n_long);
        b = __shfl(b, 0) + lane;
        if (lane < n_long && b < q.long_cap) { q.longq[b] = s_lh[wv][lane]; q.longu[b] = s_lu[wv][lane]; }
        n_long = 0;
        __builtin_amdgcn_wave_barrier();
    };
    auto flush_cd = [&]() {
        __builtin_amdgcn_wave_barrier();
        unsigned long long b = 0;
        if (lane == 0) b = atomicAdd(&q.ctr->ncand, (unsigned long long)n_cd);
        b = __shfl(b, 0) + lane;
        if (lane < n_cd && b < q.cand_cap) q.cand[b] = s_cd[wv][lane];
        n_cd = 0;
        __builtin_amdgcn_wave_barrier();
    };
    const uint64_t stride = (uint64_t)gridDim.x * EXT_THREADS;
    for (uint64_t g0 = (uint64_t)blockIdx.x * EXT_THREADS + wv * 64u; g0 < nhits; g0 += stride) {   // wave-uniform bound
        const uint64_t gid = g0 + lane;
        bool o_fol = false, o_long = false, o_cd = false;
        uint64_t r_fk = 0;
        uint32_t r_fp = 0, unit = 0;
        uint2 h = make_uint2(0, 0);
        Cand r_cd{0, 0, 0, 0, 0};
        if (gid < nhits) {
            uint32_t r = 0;
#pragma unroll
            for (int k = 0; k < 7; k++) r += gid >= sh_end[k] ? 1u : 0u;
            const uint64_t at = (uint64_t)r * q.med_cap + (gid - (r ? sh_end[r - 1] : 0));
            h = q.medq[at];
            unit = q.medu[at];
            const StrandView T = units[unit].T, Q = units[unit].Q;
            const int32_t et = (int32_t)h.x + SEED_LEN, eq = (int32_t)h.y + SEED_LEN;
            const int32_t d = (int32_t)h.x - (int32_t)h.y;
            // ---- left walk, with detection of an earlier seed hit at every reached boundary
            WalkState L{0, 0, 0, 0, false, false, 0};
            const uint32_t maxl = (uint32_t)min(et, eq);
            bool is_long = false;
            for (int win = 0; !L.done; win++) {
                if (win == LONG_WINDOWS) { is_long = true; break; }
                const int32_t P = et - 32 * (win + 1) - SEED_LEN, Pq = P - d;
                const Win64 tw = win64(T, P), qw = win64(Q, Pq);
                const uint64_t dl64 = tw.lo ^ qw.lo, dh64 = tw.hi ^ qw.hi, nm64 = dl64 | dh64;
                const uint32_t nlo = (uint32_t)nm64, nhi = (uint32_t)(nm64 >> 32);
                const uint32_t tlo = (uint32_t)dl64, thi = (uint32_t)(dl64 >> 32);
                // seed hits among the 32 starts P .. P+31: care positions carry at most one non-match,
                // and it must be a transition (dl = 0)
                uint32_t ones = 0, twos = 0, tv = 0;
#pragma unroll
                for (int c = 0; c < SEED_LEN; c++) {
                    if (!((CARE19 >> c) & 1u)) continue;
                    const uint32_t v = c ? __builtin_amdgcn_alignbit(nhi, nlo, c) : nlo;
                    twos |= ones & v;
                    ones |= v;
                    tv |= c ? __builtin_amdgcn_alignbit(thi, tlo, c) : tlo;
                }
                const uint32_t bad = transitions ? (twos | tv) : ones;
                const uint32_t H = ~bad & seedvalid32(T, P, (uint32_t)tw.sv) & (uint32_t)qw.sv;
                // walk bits in step order: step s <-> position P + 19 + 31 - s
                walk_window(tab, L, __brev((uint32_t)(dl64 >> SEED_LEN)), __brev((uint32_t)(dh64 >> SEED_LEN)),
                            __brev((uint32_t)((tw.lo ^ tw.hi) >> SEED_LEN)), __brev((uint32_t)((tw.nm | qw.nm) >> SEED_LEN)),
                            __brev(H), maxl, xdrop);
            }
            if (!is_long && L.found) {
                o_fol = true;
                r_fk = follow_key(q, unit, d, Q.len, (uint32_t)et);
                r_fp = (uint32_t)et - L.found_step;  // position of the base just summed = that seed's end
            } else {
                // ---- right walk
                WalkState R{0, 0, 0, 0, false, false, 0};
                if (!is_long) {
                    const uint32_t maxr = min(T.len - (uint32_t)et, Q.len - (uint32_t)eq);
                    for (int win = 0; !R.done; win++) {
                        if (win == LONG_WINDOWS) { is_long = true; break; }
                        const int32_t P = et + 32 * win, Pq = P - d;
                        const Win32 tw = win32(T, P), qw = win32(Q, Pq);
                        walk_window(tab, R, tw.lo ^ qw.lo, tw.hi ^ qw.hi, tw.lo ^ tw.hi, tw.nm | qw.nm, 0u, maxr, xdrop);
                    }
                }
                if (is_long) o_long = true;
                else {
                    const int32_t score = L.best + R.best;
                    if (score >= hspthresh) { o_cd = true; r_cd = Cand{(uint32_t)et - L.bk, (uint32_t)eq - L.bk, L.bk + R.bk, score, unit}; }
                }
            }
        }
        uint64_t m = __ballot(o_fol);
        if (m) {
            const uint32_t add = (uint32_t)__popcll(m);
            if (n_fol + add > GCAP) flush_fol();
            if (o_fol) { const uint32_t i = n_fol + (uint32_t)__popcll(m & lt_mask); s_fk[wv][i] = r_fk; s_fp[wv][i] = r_fp; }
            n_fol += add;
        }
        m = __ballot(o_long);
        if (m) {
            const uint32_t add = (uint32_t)__popcll(m);
            if (n_long + add > GCAP) flush_long();
            if (o_long) { const uint32_t i = n_long + (uint32_t)__popcll(m & lt_mask); s_lh[wv][i] = h; s_lu[wv][i] = unit; }
            n_long += add;
        }
        m = __ballot(o_cd);
        if (m) {
            const uint32_t add = (uint32_t)__popcll(m);
            if (n_cd + add > GCAP) flush_cd();
            if (o_cd) s_cd[wv][n_cd + (uint32_t)__popcll(m & lt_mask)] = r_cd;
            n_cd += add;
        }
    }
    if (n_fol) flush_fol();
    if (n_long) flush_long();
    if (n_cd) flush_cd();
}

// ---- heavy kernel of the A/B path: the hit array of the stand-alone K3 join, one lane per hit ----------------
// VARIANT 9 / 5: pre-filter with checkpoints every 16 steps, reading the two-plane copy (9: neither strand holds
// an N) or the full planes (5); 1 = no pre-filter, every hit is walked.  The production heavy kernel is K34
// (k34_fused.hip); this one is kept so that two independent decompositions of the stage can be compared byte for
// byte at full size (MIMEO_HEAVY=v1, tests/test_gpu_hsp.py).
template <int VARIANT>
__device__ __forceinline__ void extend_hits_body(const StrandView &T, const StrandView &Q,
                                                 const uint2 *__restrict__ hits, uint64_t nhits,
                                                 int xdrop, int hspthresh, int transitions,
                                                 const uint32_t *__restrict__ group_tab, const ExtQueues &q, uint32_t unit,
                                                 int skip_diag0, const unsigned long long *__restrict__ nhits_dev) {
    constexpr bool FILTER = VARIANT == 5 || VARIANT == 9;
    constexpr int SCAP = 256;   // staged generic-walk hits / followers per wavefront: one same-address atomic per 256 records
    // nhits_dev: the hits are the walk queue of this unit, filled by K34 just before on the same stream: eight shards of
    // capacity `nhits` each, the counts on the device.  sh_end[r] = hits in shards 0 .. r.
    uint64_t sh_end[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t shard_cap = nhits;
    if (nhits_dev) {
        uint64_t acc = 0;
#pragma unroll
        for (int r = 0; r < 8; r++) { acc += min((uint64_t)nhits_dev[r], shard_cap); sh_end[r] = acc; }
        nhits = acc;
    }
    __shared__ uint32_t tab[GROUP_TAB];
    __shared__ uint2 s_med[FAST_THREADS / 64][SCAP];
    __shared__ uint64_t s_fk[FAST_THREADS / 64][SCAP];
    __shared__ uint32_t s_fp[FAST_THREADS / 64][SCAP];
    __shared__ Cand s_cd[FAST_THREADS / 64][QCAP];
    __shared__ uint2 s_walk[FILTER ? FAST_THREADS / 64 : 1][QCAP];  // hits that passed the pre-filter, per wavefront
    for (int i = threadIdx.x; i < GROUP_TAB; i += FAST_THREADS) tab[i] = group_tab[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    WaveFill fill{0, 0, 0};
    uint32_t n_walk = 0, n_walked = 0;  // wave-uniform fill levels

    auto walk_batch = [&](bool active, uint2 h, bool final) {
        bool q_med = false, q_fol = false, q_cd = false;
        uint64_t r_fk = 0;
        uint32_t r_fp = 0;
        Cand r_cd{0, 0, 0, 0, 0};
        if (active) walk_hit<1>(tab, T, Q, h, xdrop, hspthresh, transitions, q_med, q_fol, q_cd, r_fk, r_fp, r_cd);
        stage_records<true, SCAP>(q, unit, s_med[wv], s_fk[wv], s_fp[wv], s_cd[wv], fill, q_med, h, q_fol, r_fk, r_fp, q_cd, r_cd, final);
    };

    const uint64_t stride = (uint64_t)gridDim.x * FAST_THREADS;
    for (uint64_t g0 = (uint64_t)blockIdx.x * FAST_THREADS + wv * 64u; g0 < nhits; g0 += stride) {
        const uint64_t gid = g0 + lane;
        uint2 h = make_uint2(0, 0);
        if (gid < nhits) {
            uint64_t at = gid;
            if (nhits_dev) {   // flat index -> (shard, index in the shard)
                uint32_t r = 0;
#pragma unroll
                for (int k = 0; k < 7; k++) r += gid >= sh_end[k] ? 1u : 0u;
                at = (uint64_t)r * shard_cap + (gid - (r ? sh_end[r - 1] : 0));
            }
            h = hits[at];
        }
        const bool valid = gid < nhits && !(skip_diag0 && h.x == h.y);  // the main diagonal of a self unit belongs to k4_diag0
        if (!FILTER) {
            walk_batch(valid, h, false);
        } else {
            bool need = false;
            if (valid)
                need = VARIANT == 9 ? hit_needs_walk<16, true>(T, Q, h, xdrop, hspthresh, transitions)
                                    : hit_needs_walk<16, false>(T, Q, h, xdrop, hspthresh, transitions);
            const uint64_t m = __ballot(need);
            if (m) {
                const uint32_t add = (uint32_t)__popcll(m);
                if (n_walk + add > (uint32_t)QCAP) {  // walk what is queued (nearly a full wavefront), then queue
                    __builtin_amdgcn_wave_barrier();
                    const uint2 hq = s_walk[wv][lane < n_walk ? lane : 0];
                    walk_batch(lane < n_walk, hq, false);
                    n_walked += n_walk;
                    n_walk = 0;
                    __builtin_amdgcn_wave_barrier();
                }
                if (need) s_walk[wv][n_walk + __popcll(m & lt_mask)] = h;
                n_walk += add;
            }
        }
    }
    if (FILTER && n_walk) {
        __builtin_amdgcn_wave_barrier();
        const uint2 hq = s_walk[wv][lane < n_walk ? lane : 0];
        walk_batch(lane < n_walk, hq, false);
        n_walked += n_walk;
    }
    // the statistic: one atomic per WORKGROUP (a same-address atomic serialises at ~13 ns: one per wavefront was 13 us of a
    // 190 us launch)
    __shared__ unsigned int s_nwalked;
    if (FILTER) {
        if (threadIdx.x == 0) s_nwalked = 0;
        __syncthreads();
        if (n_walked && lane == 0) atomicAdd(&s_nwalked, n_walked);
        __syncthreads();
        if (threadIdx.x == 0 && s_nwalked) atomicAdd(&q.ctr->nwalked, (unsigned long long)s_nwalked);
    }
    walk_batch(false, make_uint2(0, 0), true);  // flush the staged records
    if (!nhits_dev && blockIdx.x == 0 && threadIdx.x == 0) q.unit_hits[unit] = nhits;
}
// the hit array of ONE unit (A/B path: K3's join)
template <int VARIANT>
__global__ __launch_bounds__(FAST_THREADS) void k4_extend_hits(StrandView T, StrandView Q, const uint2 *__restrict__ hits, uint64_t nhits,
                                                              int xdrop, int hspthresh, int transitions,
                                                              const uint32_t *__restrict__ group_tab, ExtQueues q, uint32_t unit, int skip_diag0) {
    extend_hits_body<VARIANT>(T, Q, hits, nhits, xdrop, hspthresh, transitions, group_tab, q, unit, skip_diag0, nullptr);
}
// the walk queues of ALL units of the batch in one launch (grid.y = unit): a unit's region holds eight shards of walk_cap
// entries, filled by K34 just before on the same stream, the counts on the device
template <int VARIANT>
__global__ __launch_bounds__(FAST_THREADS) void k4_walk_batch(const FusedUnit *__restrict__ funits, int xdrop, int hspthresh, int transitions,
                                                             const uint32_t *__restrict__ group_tab, ExtQueues q) {
    const FusedUnit &U = funits[blockIdx.y];
    extend_hits_body<VARIANT>(U.Tv, U.Qv, q.walkq + U.walk_base, U.walk_cap, xdrop, hspthresh, transitions, group_tab, q, U.unit, 0,
                              q.nwalk_u + (size_t)blockIdx.y * 8);
}

// the walk queues of the batch: total entries, and the fullest shard in 1/1024 of its capacity (> 1024: an overflow, the
// batch is repeated with larger regions)
__global__ __launch_bounds__(256) void k4_walk_summary(const FusedUnit *__restrict__ funits, uint32_t nactive, const unsigned long long *__restrict__ nwalk_u,
                                                       ExtCounters *ctr) {
    unsigned long long tot = 0, worst = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nactive * 8u; i += gridDim.x * 256) {
        const unsigned long long n = nwalk_u[i], cap = funits[i >> 3].walk_cap;
        tot += n;
        worst = max(worst, cap ? (n * 1024ull + cap - 1) / cap : (n ? 1ull << 40 : 0ull));
    }
    for (int o = 32; o > 0; o >>= 1) { tot += __shfl_xor(tot, o); worst = max(worst, (unsigned long long)__shfl_xor(worst, o)); }
    if ((threadIdx.x & 63) == 0) { if (tot) atomicAdd(&ctr->nwalk_total, tot); if (worst) atomicMax(&ctr->nwalk_over, worst); }
}

// ---- long hits, one wavefront each ------------------------------------------------------------------------
// (three wavefronts per SIMD fit its registers; the hits differ in length by orders of magnitude — a microsatellite diagonal
// against a few hundred bases — so a wavefront takes the next 16 hits from a counter instead of a fixed stride; records: WaveStage)
constexpr uint32_t CLAIM = 16;
__global__ __launch_bounds__(EXT_THREADS) void k4_extend_long(const UnitDesc *__restrict__ units, ExtQueues q, int xdrop,
                                                              int hspthresh, int transitions) {
    __shared__ WaveStage s_stage[EXT_THREADS / 64];
    WaveStage &S = s_stage[threadIdx.x >> 6];
    uint32_t nf = 0, nc = 0;
    const uint64_t nlong = min((uint64_t)q.ctr->nlong, q.long_cap);
    for (;;) {
        unsigned long long w0 = 0;
        if ((threadIdx.x & 63u) == 0) w0 = atomicAdd(&q.ctr->long_next, (unsigned long long)CLAIM);
        w0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w0);
        if (w0 >= nlong) break;
        for (uint64_t wid = w0; wid < min(nlong, (uint64_t)w0 + CLAIM); wid++) {
            const uint32_t unit = q.longu[wid];
            stage_push(S, nf, nc, wave_extend_record(units[unit].T, units[unit].Q, q.longq[wid], xdrop, hspthresh, transitions, true, q, unit, nullptr), q);
        }
    }
    stage_flush(S, nf, nc, q);
}

// ---- the follower shards gathered into one array (input of the sort) ---------------------------------------------
__global__ __launch_bounds__(256) void k4_compact_followers(ExtQueues q, uint64_t *__restrict__ key, uint32_t *__restrict__ prev) {
    uint64_t sh_end[8];
    uint64_t acc = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) { acc += min((uint64_t)q.ctr->nfollow8[r], q.follow_cap); sh_end[r] = acc; }
    if (blockIdx.x == 0 && threadIdx.x == 0) q.ctr->nfollow = acc;
    for (uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x; gid < acc; gid += (uint64_t)gridDim.x * 256) {
        uint32_t r = 0;
#pragma unroll
        for (int k = 0; k < 7; k++) r += gid >= sh_end[k] ? 1u : 0u;
        const uint64_t at = (uint64_t)r * q.follow_cap + (gid - (r ? sh_end[r - 1] : 0));
        key[gid] = q.fkey[at];
        prev[gid] = q.fprev[at];
    }
}

// ---- follower segments --------------------------------------------------------------------------------------
// flag[i] = 1 iff sorted follower i starts a segment (its predecessor is a head, not follower i-1)
__global__ void k4_segment_flags(const uint64_t *__restrict__ key, const uint32_t *__restrict__ prev, uint64_t n,
                                 ExtQueues q, uint8_t *__restrict__ flag) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool start = true;
    if (i > 0) {
        uint64_t a = key[i - 1], b = key[i];
        // a RUN_END record closes the run its predecessor in the list opened (or continued): never a segment start
        start = prev[i] != RUN_END && (key_unit_diag(q, a) != key_unit_diag(q, b) || key_end(q, a) != prev[i]);
    }
    flag[i] = start ? 1 : 0;
}

// One LANE per segment: almost every segment is a pair of neighbouring random hits (one head, one
// follower, walks of a few dozen bases).  Segments with more than SMALL_SEG members or with a walk
// longer than LONG_WINDOWS windows are queued for k4_resolve_segments (one wavefront each).
constexpr uint32_t SMALL_SEG = 8;
__global__ __launch_bounds__(EXT_THREADS) void k4_resolve_small(const UnitDesc *__restrict__ units, ExtQueues q,
                                                                const uint64_t *__restrict__ key,
                                                                const uint32_t *__restrict__ prev, uint64_t nfollow,
                                                                const uint64_t *__restrict__ seg_start,
                                                                const uint64_t *__restrict__ nseg_dev,
                                                                int xdrop, int hspthresh,
                                                                const uint32_t *__restrict__ group_tab,
                                                                uint64_t *__restrict__ bigseg) {
    __shared__ uint32_t tab[GROUP_TAB];
    const uint64_t nseg = *nseg_dev;
    if ((uint64_t)blockIdx.x * EXT_THREADS >= nseg) return;  // the grid is sized for the followers, segments are fewer
    for (int i = threadIdx.x; i < GROUP_TAB; i += EXT_THREADS) tab[i] = group_tab[i];
    __syncthreads();
    const uint64_t sid = (uint64_t)blockIdx.x * EXT_THREADS + threadIdx.x;
    if (sid >= nseg) return;
    const uint64_t beg = seg_start[sid], end = sid + 1 < nseg ? seg_start[sid + 1] : nfollow;
    bool big = end - beg > SMALL_SEG;
    Cand out[SMALL_SEG];
    uint32_t nout = 0;
    const uint32_t unit = key_unit(q, key[beg]);
    if (!big) {
        const StrandView T = units[unit].T, Q = units[unit].Q;
        const int32_t d = key_diag(q, key[beg], Q.len);
        const int32_t het = (int32_t)prev[beg];
        WalkState H{0, 0, 0, 0, false, false, 0};
        big = !lane_walk(tab, T, Q, het, d, +1, min(T.len - (uint32_t)het, Q.len - (uint32_t)(het - d)), xdrop, H);
        uint32_t reach = (uint32_t)het + H.bk;
        for (uint64_t i = beg; i < end && !big; i++) {
            // the positions this record stands for: its own seed end, or — a RUN_END record — every one behind the record before it
            const uint32_t hi = key_end(q, key[i]);
            const uint32_t lo = prev[i] == RUN_END ? key_end(q, key[i - 1]) + 1u : hi;
            for (uint32_t et = max(lo, reach + 1u); et <= hi && !big; et = reach + 1u) {   // the next one beyond the region the previous extension reached
                const int32_t eq = (int32_t)et - d;
                WalkState L{0, 0, 0, 0, false, false, 0}, R{0, 0, 0, 0, false, false, 0};
                if (nout == SMALL_SEG ||
                    !lane_walk(tab, T, Q, (int32_t)et, d, -1, (uint32_t)min((int32_t)et, eq), xdrop, L) ||
                    !lane_walk(tab, T, Q, (int32_t)et, d, +1, min(T.len - et, Q.len - (uint32_t)eq), xdrop, R)) {
                    big = true;
                    break;
                }
                if (L.best + R.best >= hspthresh) out[nout++] = Cand{et - L.bk, (uint32_t)eq - L.bk, L.bk + R.bk, L.best + R.best, unit};
                reach = et + R.bk;
            }
        }
    }
    if (big) {  // nothing has been emitted for this segment yet: the wavefront kernel redoes it
        bigseg[wave_slot(&q.ctr->nbig)] = sid;
        return;
    }
    for (uint32_t k = 0; k < nout; k++) {
        const unsigned long long i = wave_slot(&q.ctr->ncand);
        if (i < q.cand_cap) q.cand[i] = out[k];
    }
}

// one wavefront per segment: replay "skip while seed end <= reach, else extend" (lastz diagEnd rule)
__global__ __launch_bounds__(EXT_THREADS) void k4_resolve_segments(const UnitDesc *__restrict__ units, ExtQueues q,
                                                                   const uint64_t *__restrict__ key,
                                                                   const uint32_t *__restrict__ prev, uint64_t nfollow,
                                                                   const uint64_t *__restrict__ seg_start,
                                                                   const uint64_t *__restrict__ nseg_dev,
                                                                   const uint64_t *__restrict__ list, int xdrop,
                                                                   int hspthresh, int transitions) {
    const uint64_t nseg = *nseg_dev, nlist = q.ctr->nbig;
    __shared__ WaveStage s_stage[EXT_THREADS / 64];
    WaveStage &S = s_stage[threadIdx.x >> 6];
    uint32_t nf = 0, nc = 0;
    for (;;) {   // segments differ in size by orders of magnitude: the next 16 from a counter
    unsigned long long w0 = 0;
    if ((threadIdx.x & 63) == 0) w0 = atomicAdd(&q.ctr->seg_next, (unsigned long long)CLAIM);
    w0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w0 >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w0);
    if (w0 >= nlist) break;
    // the heads of the 16 segments, one per lane: four dependent loads for all of them instead of four for each
    uint64_t l_beg = 0, l_end = 0, l_k0 = 0;
    uint32_t l_het = 0;
    if ((threadIdx.x & 63u) < CLAIM && w0 + (threadIdx.x & 63u) < nlist) {
        const uint64_t sid = list[w0 + (threadIdx.x & 63u)];
        l_beg = seg_start[sid];
        l_end = sid + 1 < nseg ? seg_start[sid + 1] : nfollow;
        l_k0 = key[l_beg];
        l_het = prev[l_beg];
    }
    for (uint64_t wid = w0; wid < min(nlist, (uint64_t)w0 + CLAIM); wid++) {
    const int src = (int)(wid - w0);
    uint64_t beg = __shfl(l_beg, src), end = __shfl(l_end, src);
    uint64_t k0 = __shfl(l_k0, src);
    const uint32_t unit = key_unit(q, k0);
    const StrandView T = units[unit].T, Q = units[unit].Q;
    const int32_t d = key_diag(q, k0, Q.len);
    // head: seed end = prev[beg]; only its right extent matters (it was emitted by the heavy kernel or the walks)
    int32_t het = (int32_t)__shfl(l_het, src);
    WalkResult R = wave_walk_fast(T, Q, het, d, +1, min(T.len - (uint32_t)het, Q.len - (uint32_t)(het - d)), xdrop);
    uint32_t reach = (uint32_t)het + R.bsteps;
    uint64_t i = beg;
    while (i < end) {
        // first record at or after i that reaches beyond `reach` (records are sorted by seed end; a RUN_END record stands for
        // every position behind the record before it up to its own)
        // (the next 64 records in one load, lane by lane — most segments end there; a binary search, one memory round trip per
        // step, only behind them)
        uint64_t nxt;
        {
            const uint64_t x = i + (threadIdx.x & 63u);
            const uint64_t m = __ballot(x < end && key_end(q, key[x]) > reach);
            if (m) nxt = i + (uint64_t)__builtin_ctzll(m);
            else {
                uint64_t lo = min(end, i + 64u), hi = end;
                while (lo < hi) {
                    uint64_t mid = (lo + hi) >> 1;
                    if (key_end(q, key[mid]) > reach) hi = mid; else lo = mid + 1;
                }
                nxt = lo;
            }
        }
        if (nxt >= end) break;
        uint32_t et = key_end(q, key[nxt]);
        if (prev[nxt] == RUN_END) et = max(key_end(q, key[nxt - 1]) + 1u, reach + 1u);   // the first member of the run beyond the reach
        uint2 h = make_uint2(et - SEED_LEN, (uint32_t)((int32_t)et - d) - SEED_LEN);
        uint32_t rext = 0;
        stage_push(S, nf, nc, wave_extend_record(T, Q, h, xdrop, hspthresh, transitions, false, q, unit, &rext), q);
        reach = et + rext;
        i = nxt;   // the same record again: a run may hold more members beyond the new reach (a plain record is passed by the search)
    }
    }
    }
    stage_flush(S, nf, nc, q);
}

// ---- the main diagonal of a strand aligned to itself ---------------------------------------------------------
// When target and query are the same strand every valid seed position is a hit on diagonal 0
// (millions of followers of one head).  The heavy kernel leaves those hits alone and this kernel replays the
// sequential rule directly on the seed-validity planes with one wavefront per self unit: extend the first seed,
// skip every seed whose end lies inside the reach, extend the next one, ...
__global__ __launch_bounds__(64) void k4_diag0(const UnitDesc *__restrict__ units, const uint32_t *__restrict__ selfs, ExtQueues q,
                                               int xdrop, int hspthresh, int transitions) {
    const uint32_t unit = selfs[blockIdx.x];
    const StrandView T = units[unit].T, Q = units[unit].Q;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nwords = (T.len + 31u) >> 5;
    uint32_t p = 0;  // first seed start still to be considered
    for (;;) {
        uint32_t found = 0xFFFFFFFFu;
        for (uint32_t w0 = p >> 5; w0 < nwords; w0 += 64u) {
            const uint32_t w = w0 + lane;
            uint32_t m = 0;
            if (w < nwords) {
                m = (T.svt ? T.svt[w] : T.pw[w].w) & Q.pw[w].w;
                if (w == (p >> 5)) m &= 0xFFFFFFFFu << (p & 31u);
            }
            const uint64_t b = __ballot(m != 0);
            if (b) {
                const int l = __builtin_ctzll(b);
                const uint32_t mm = (uint32_t)__shfl((int)m, l);
                found = ((w0 + (uint32_t)l) << 5) + (uint32_t)__builtin_ctz(mm);
                break;
            }
        }
        if (found == 0xFFFFFFFFu) break;
        uint32_t rext = 0;
        wave_extend_emit(T, Q, make_uint2(found, found), xdrop, hspthresh, transitions, false, q, unit, &rext);
        const uint32_t reach = found + SEED_LEN + rext;  // a later seed is extended iff its end lies beyond
        p = max(found + 1u, reach - (uint32_t)(SEED_LEN - 1));
    }
}

// ---- entropy adjustment + threshold: eight lanes per candidate ------------------------------------------------
// A C4 row has 1.5 M candidates, nearly all of them 30-300 columns long: a group of eight lanes takes one (256
// columns per step, reductions over three xor-shuffles), so a wavefront finishes eight candidates at a time; the
// few long ones (the main diagonal of a self unit: millions of columns) just take more steps of their group.
constexpr uint32_t ENT_LANES = 8;
constexpr uint32_t ENT_LONG = 1u << 16;     // columns: longer candidates (the main diagonal of a self unit: the whole scaffold) ...
constexpr uint32_t ENT_BIGCAP = 4096;       // ... are counted by the whole grid (k4_entropy_big), up to this many per batch
constexpr uint32_t ENT_STAGE = 64;   // HSPs a wavefront collects before it appends them (8 per step at most)
__device__ __forceinline__ void entropy_flush(const mimeo_hsp *sh, const uint32_t *su, uint32_t &n, const ExtQueues &q, mimeo_hsp *__restrict__ out,
                                              uint32_t *__restrict__ out_unit) {
    __builtin_amdgcn_wave_barrier();
    if (n) {
        const uint32_t lane = threadIdx.x & 63u;
        unsigned long long b = 0;
        if (lane == 0) b = atomicAdd(&q.ctr->nhsp, (unsigned long long)n);
        b = __shfl(b, 0);
        if (lane < n) { out[b + lane] = sh[lane]; out_unit[b + lane] = su[lane]; }
    }
    __builtin_amdgcn_wave_barrier();
    n = 0;
}
__global__ __launch_bounds__(EXT_THREADS) void k4_entropy(const UnitDesc *__restrict__ units, ExtQueues q, int hspthresh,
                                                          int entropy, mimeo_hsp *__restrict__ out,
                                                          uint32_t *__restrict__ out_unit) {
    __shared__ mimeo_hsp s_h[EXT_THREADS / 64][ENT_STAGE];
    __shared__ uint32_t s_u[EXT_THREADS / 64][ENT_STAGE];
    const uint32_t wv = threadIdx.x >> 6;
    uint32_t nstage = 0;
    const uint32_t sub = threadIdx.x & (ENT_LANES - 1u);
    const uint64_t ncand = min((uint64_t)q.ctr->ncand, q.cand_cap);
    const uint64_t ngroups = ((uint64_t)gridDim.x * EXT_THREADS) / ENT_LANES;
    // the loop bound is wave-uniform (rounded up to the wavefront's eight groups): the shuffles below need every lane
    const uint64_t first = (((uint64_t)blockIdx.x * EXT_THREADS + threadIdx.x) / 64) * (64 / ENT_LANES);
    for (uint64_t base = first; base < ncand; base += ngroups) {
    const uint64_t cid = base + (threadIdx.x & 63u) / ENT_LANES;
    const bool live = cid < ncand;
    Cand c = q.cand[live ? cid : ncand - 1];
    if (!live) c.len = 0;
    if (c.len > ENT_LONG) {   // one group of eight lanes would crawl through millions of columns with the wavefront waiting
        unsigned long long slot = ENT_BIGCAP;
        if (sub == 0) slot = wave_slot(&q.ctr->nbigcand);
        slot = __shfl(slot, (int)((threadIdx.x & 63u) & ~(ENT_LANES - 1u)));
        if (slot < ENT_BIGCAP) {
            if (sub == 0) q.bigcand[slot] = cid;
            c.len = 0;   // nothing to do here; k4_entropy_big emits it
        }
    }
    const bool deferred = live && c.len == 0;
    const StrandView T = units[c.unit].T, Q = units[c.unit].Q;
    int64_t raw = c.raw;
    const int32_t d = (int32_t)c.tstart - (int32_t)c.qstart;
    // the longest candidate of the wavefront sets the number of steps (wave-uniform loops: shuffles inside)
    uint32_t maxlen = c.len;
    for (int o = 32; o >= (int)ENT_LANES; o >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, o));
    if (__ballot(c.raw == RAW_SATURATED)) {  // recount the column scores of the segment in 64 bits (rare)
        int64_t sum = 0;
        for (uint32_t o0 = 0; o0 < maxlen; o0 += ENT_LANES * 32u) {
            const uint32_t o = o0 + sub * 32u;
            if (o < c.len && c.raw == RAW_SATURATED) {
                const int32_t pt = (int32_t)(c.tstart + o);
                const Win32 tw = win32(T, pt), qw = win32(Q, pt - d);
                const uint32_t rem = c.len - o, valid = rem < 32 ? (1u << rem) - 1u : 0xFFFFFFFFu;
                const uint32_t nn = (tw.nm | qw.nm) & valid, dl = (tw.lo ^ qw.lo) & ~nn & valid, dh = (tw.hi ^ qw.hi) & ~nn & valid;
                const uint32_t cg = tw.lo ^ tw.hi, ok = valid & ~nn;
                sum += 91ll * __popc(ok) + 9ll * __popc(ok & ~(dl | dh) & cg) - 122ll * __popc(~dl & dh) - 205ll * __popc(dl) -
                       9ll * __popc(dl & dh) - 2ll * __popc(dl & dh & cg) - 100ll * __popc(nn);
            }
        }
        for (int o = ENT_LANES / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (c.raw == RAW_SATURATED) raw = sum;
    }
    int64_t adj = raw;
    if (entropy) {
        uint32_t cnt[4] = {0, 0, 0, 0};
        for (uint32_t o0 = 0; o0 < maxlen; o0 += ENT_LANES * 32u) {
            const uint32_t o = o0 + sub * 32u;
            if (o < c.len) {
                const int32_t pt = (int32_t)(c.tstart + o);
                const Win32 tw = win32(T, pt), qw = win32(Q, pt - d);
                const uint32_t tlo = tw.lo, thi = tw.hi;
                uint32_t m = ~((tlo ^ qw.lo) | (thi ^ qw.hi)) & ~(tw.nm | qw.nm);
                const uint32_t rem = c.len - o;
                if (rem < 32) m &= (1u << rem) - 1u;
                cnt[0] += __popc(m & ~tlo & ~thi);
                cnt[1] += __popc(m & tlo & ~thi);
                cnt[2] += __popc(m & ~tlo & thi);
                cnt[3] += __popc(m & tlo & thi);
            }
        }
        for (int b = 0; b < 4; b++)
            for (int o = ENT_LANES / 2; o > 0; o >>= 1) cnt[b] += __shfl_xor(cnt[b], o);
        uint64_t n = (uint64_t)cnt[0] + cnt[1] + cnt[2] + cnt[3];
        double hh = 0.0;
        if (n) {
            for (int b = 0; b < 4; b++)
                if (cnt[b]) { double p = (double)cnt[b] / (double)n; hh -= p * log(p); }
            hh /= log(4.0);
        }
        int64_t q16 = (int64_t)floor(hh * 65536.0 + 0.5);
        q16 = q16 > 65536 ? 65536 : (q16 < 0 ? 0 : q16);
        adj = (raw * q16) >> 16;
    }
    // The HSPs wait in LDS until the wavefront has 64 of them: one atomic per wavefront and STEP was 2.5 * 10^6 same-address
    // atomics for the 2.7 * 10^7 candidates of 32 C5 units — 13 ns each, the whole duration of this kernel
    const bool emit = live && !deferred && adj >= hspthresh && sub == 0;
    const uint64_t em = __ballot(emit);
    if (em) {
        const uint32_t add = (uint32_t)__popcll(em);
        if (nstage + add > ENT_STAGE) entropy_flush(s_h[wv], s_u[wv], nstage, q, out, out_unit);
        if (emit) {
            const uint32_t at = nstage + (uint32_t)__popcll(em & ((1ull << (threadIdx.x & 63u)) - 1ull));
            mimeo_hsp h;
            h.tstart = c.tstart; h.qstart = c.qstart; h.length = c.len; h.flags = 0; h.score = adj; h.raw_score = raw;
            s_h[wv][at] = h;   // at most one HSP per candidate: the buffers hold cand_cap records
            s_u[wv][at] = c.unit;
        }
        nstage += add;
    }
    }
    entropy_flush(s_h[wv], s_u[wv], nstage, q, out, out_unit);
}

// the long candidates: every workgroup counts a slice (matched columns per base, and the raw score again in 64 bits
// when it had saturated), then one lane per candidate applies the same formula as k4_entropy
__global__ __launch_bounds__(256) void k4_entropy_big(const UnitDesc *__restrict__ units, ExtQueues q) {
    const uint64_t nbig = min((uint64_t)q.ctr->nbigcand, (uint64_t)ENT_BIGCAP);
    __shared__ unsigned long long red[5];
    for (uint64_t bi = blockIdx.y; bi < nbig; bi += gridDim.y) {
        const Cand c = q.cand[q.bigcand[bi]];
        const StrandView T = units[c.unit].T, Q = units[c.unit].Q;
        const int32_t d = (int32_t)c.tstart - (int32_t)c.qstart;
        unsigned long long cnt[4] = {0, 0, 0, 0};
        long long sum = 0;
        for (uint64_t o = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 32u; o < c.len; o += (uint64_t)gridDim.x * 256 * 32u) {
            const int32_t pt = (int32_t)(c.tstart + o);
            const Win32 tw = win32(T, pt), qw = win32(Q, pt - d);
            const uint32_t rem = (uint32_t)(c.len - o), valid = rem < 32 ? (1u << rem) - 1u : 0xFFFFFFFFu;
            const uint32_t nn = (tw.nm | qw.nm) & valid, dl = (tw.lo ^ qw.lo) & ~nn & valid, dh = (tw.hi ^ qw.hi) & ~nn & valid;
            const uint32_t cg = tw.lo ^ tw.hi, ok = valid & ~nn, m = ok & ~(dl | dh);
            cnt[0] += __popc(m & ~tw.lo & ~tw.hi);
            cnt[1] += __popc(m & tw.lo & ~tw.hi);
            cnt[2] += __popc(m & ~tw.lo & tw.hi);
            cnt[3] += __popc(m & tw.lo & tw.hi);
            sum += 91ll * __popc(ok) + 9ll * __popc(m & cg) - 122ll * __popc(~dl & dh & ok) - 205ll * __popc(dl) -
                   9ll * __popc(dl & dh) - 2ll * __popc(dl & dh & cg) - 100ll * __popc(nn);
        }
        if (threadIdx.x < 5) red[threadIdx.x] = 0;
        __syncthreads();
        for (int k = 0; k < 4; k++) {
            unsigned long long v = cnt[k];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if ((threadIdx.x & 63) == 0) atomicAdd(&red[k], v);
        }
        {
            long long v = sum;
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if ((threadIdx.x & 63) == 0) atomicAdd(&red[4], (unsigned long long)v);
        }
        __syncthreads();
        if (threadIdx.x < 5) atomicAdd(&q.bigacc[bi * 5 + threadIdx.x], red[threadIdx.x]);
        __syncthreads();
    }
}
__global__ void k4_entropy_big_finish(ExtQueues q, int hspthresh, int entropy, mimeo_hsp *__restrict__ out, uint32_t *__restrict__ out_unit) {
    const uint64_t nbig = min((uint64_t)q.ctr->nbigcand, (uint64_t)ENT_BIGCAP);
    for (uint64_t bi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; bi < nbig; bi += (uint64_t)gridDim.x * blockDim.x) {
        const Cand c = q.cand[q.bigcand[bi]];
        const unsigned long long *a = q.bigacc + bi * 5;
        int64_t raw = c.raw == RAW_SATURATED ? (int64_t)a[4] : (int64_t)c.raw, adj = raw;
        if (entropy) {
            const uint64_t n = a[0] + a[1] + a[2] + a[3];
            double hh = 0.0;
            if (n) {
                for (int b = 0; b < 4; b++)
                    if (a[b]) { double p = (double)a[b] / (double)n; hh -= p * log(p); }
                hh /= log(4.0);
            }
            int64_t q16 = (int64_t)floor(hh * 65536.0 + 0.5);
            q16 = q16 > 65536 ? 65536 : (q16 < 0 ? 0 : q16);
            adj = (raw * q16) >> 16;
        }
        if (adj >= hspthresh) {
            const unsigned long long i = atomicAdd(&q.ctr->nhsp, 1ull);
            mimeo_hsp h;
            h.tstart = c.tstart; h.qstart = c.qstart; h.length = c.len; h.flags = 0; h.score = adj; h.raw_score = raw;
            out[i] = h;
            out_unit[i] = c.unit;
        }
    }
}

// ---- shared plus strand: the HSPs of unit u once more, transposed, for its mirror unit ------------------------------------
// (A, B, +) and (B, A, +) of a self job: every rule of the gap-free stage is symmetric in target and query (seed words
// and the one-transition rule, N, the x-drop walks, the per-diagonal "already extended" rule — a diagonal keeps its
// order under the swap —, entropy of the identical columns, HOXD70), so the HSP set of (B, A, +) is the set of (A, B, +)
// with tstart and qstart exchanged; the pipeline only pairs units when neither scaffold has soft-masked bases (lastz
// excludes them from TARGET seeding only).  ctr[0] counts the copies.
__global__ __launch_bounds__(256) void k4_mirror_hsps(mimeo_hsp *__restrict__ hsps, uint32_t *__restrict__ hsp_unit, uint64_t nh,
                                                      const uint32_t *__restrict__ mirror_dst, unsigned long long *__restrict__ ctr) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nh; i += (uint64_t)gridDim.x * 256) {
        const uint32_t m = mirror_dst[hsp_unit[i]];
        if (m == NO_MIRROR) continue;
        const unsigned long long at = nh + wave_slot(ctr);
        mimeo_hsp h = hsps[i];
        const uint32_t t = h.tstart;
        h.tstart = h.qstart; h.qstart = t;
        hsps[at] = h;
        hsp_unit[at] = m;
    }
}

// ---- host orchestration ---------------------------------------------------------------------
static uint32_t *g_group_tab = nullptr;  // device copy of the 4-base group table (read-only)
static std::once_flag g_group_once;

const uint32_t *group_table_device() {
    std::call_once(g_group_once, [&] {
        std::vector<uint32_t> tab(GROUP_TAB);
        build_group_table(tab.data());
        if (hipMalloc((void **)&g_group_tab, GROUP_TAB * 4) != hipSuccess ||
            hipMemcpy(g_group_tab, tab.data(), GROUP_TAB * 4, hipMemcpyHostToDevice) != hipSuccess)
            g_group_tab = nullptr;
    });
    return g_group_tab;
}

// fused heavy kernel (k34_fused.hip)
int launch_fused_batch(const FusedUnit *d_units, uint32_t nactive, const ExtQueues &q, const HeavyPlan &plan, const mimeo_params *p, hipStream_t st,
                       uint32_t dbg);
void launch_sum_hits(const ExtQueues &q, uint32_t nunits, hipStream_t st);

void ExtBatch::release() {
    for (DeviceBuf *b : {&units, &ctr, &cand, &fkey, &fkey2, &fprev, &fprev2, &medq, &medu, &longq, &longu, &walkq, &flags, &segs, &tmp,
                         &nsel, &bigseg, &hsps, &hsp_unit, &unit_hits, &tile_hits, &selfs, &hits, &bigcand, &bigacc, &heavy, &mirror, &funits, &nwalk_u, &plan_buf, &arena_q, &arena_s})
        b->release();
    jc.release();
    for (auto &e : ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    for (auto &e : kev) (void)hipEventDestroy(e);
    kev.clear();
    if (side_done) { (void)hipEventDestroy(side_done); side_done = nullptr; }
    if (side) { (void)hipStreamDestroy(side); side = nullptr; }
    delete (ExtQueues *)q_;
    q_ = nullptr;
    started_ = false;
}

static uint32_t bits_for(uint64_t v) {  // bits needed to hold values 0 .. v
    uint32_t b = 1;
    while (b < 64 && (v >> b)) b++;
    return b;
}

uint32_t ext_batch_max_units(uint64_t max_tlen, uint64_t max_qlen) {
    const uint32_t ebits = bits_for(max_tlen + SEED_LEN), dbits = bits_for(max_tlen + max_qlen + SEED_LEN);
    const uint32_t ubits = 64 - std::min(63u, ebits + dbits);
    return ubits >= 20 ? (1u << 20) : (1u << ubits);
}

int ExtBatch::run(const std::vector<UnitWork> &work, const mimeo_params *p, uint64_t *nhsp_out, ExtStats *stats,
                  const std::vector<uint32_t> *mirror_dst) {
    *nhsp_out = 0;
    if (work.empty()) return 0;
    int rc = start(work, p, mirror_dst);
    if (rc) return rc;
    return finish(nhsp_out, stats);
}

// Device bytes of the batch's queues at the current capacities: followers and walks beyond the frame in eight shards (key +
// predecessor / hit + unit), long walks, candidates + HSPs (+ their mirror copies), the walk queue of one unit.  The
// follower sort's second buffer, flags, segment lists and rocPRIM's scratch are sized by the followers actually seen
// (finish()): up to 41 bytes each on top.
uint64_t ExtBatch::arena_bytes() const {   // the queues proper: what the first arena holds
    const uint64_t mf = mirror_dst_.empty() ? 1 : 2;
    return cap_f_ * 8 * 12 + cap_m_ * 8 * 12 + cap_l_ * 12 + cap_c_ * (sizeof(Cand) + mf * (sizeof(mimeo_hsp) + 4)) + (v1_ ? 8 : walk_entries_ * 8) + 4096;
}
// ... plus what the rest of the batch may ask for while they are alive: the follower sort's second arena at its worst (every
// shard full: 41 bytes per follower) and the chain / gapped stage's scratch and alignment slots (~200 bytes per HSP, 60 more
// where the chain stage meets large groups: k5_chain_wave's orders and tree)
uint64_t ExtBatch::queue_bytes() const {
    const uint64_t mf = mirror_dst_.empty() ? 1 : 2;
    return arena_bytes() + cap_f_ * 8 * 41 + cap_c_ * mf * 260;
}
// carve `bytes` (rounded up to 256) off an arena at *off
static void *carve(const DeviceBuf &arena, size_t *off, size_t bytes) {
    void *p = (char *)arena.p + *off;
    *off += (bytes + 255) & ~(size_t)255;
    return p;
}
// the arenas back to the device pool (mimeo_shutdown, or a call that ends in an allocation failure)
void ExtBatch::release_queues() {
    for (DeviceBuf *d : {&cand, &fkey, &fkey2, &fprev, &fprev2, &medq, &medu, &longq, &longu, &walkq, &flags, &segs, &tmp, &bigseg, &hsps, &hsp_unit})
        d->release();   // slices: nothing is freed here
    arena_q.release();
    arena_s.release();
}
uint64_t ExtBatch::held_bytes() const { return arena_q.cap + arena_s.cap; }
// free device memory plus what the batch's own buffers would give back, less a reserve for chain / gapped scratch
static int queue_budget(const ExtBatch &b, uint64_t *budget) {
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const uint64_t have = (uint64_t)free_b + b.held_bytes(), reserve = std::min<uint64_t>(have / 8, 8ull << 30);
    *budget = have - reserve;
    if (getenv("MIMEO_QUEUE_BUDGET_MB")) *budget = (uint64_t)atol(getenv("MIMEO_QUEUE_BUDGET_MB")) << 20;   // tests: force the split
    return 0;
}

int ExtBatch::start(const std::vector<UnitWork> &work, const mimeo_params *p, const std::vector<uint32_t> *mirror_dst) {
    hipStream_t st = stream();
    started_ = false;
    const uint32_t nunits = (uint32_t)work.size();
    if (!nunits) return 0;
    w_ = work;
    p_ = *p;
    uint64_t max_t = 0, max_q = 0;
    double expect_hits = 0, max_unit_hits = 0;
    std::vector<UnitDesc> &h_units = h_units_;   // a member: the copy below is asynchronous
    h_units.resize(nunits);
    h_selfs_.clear();
    for (uint32_t u = 0; u < nunits; u++) {
        const UnitWork &w = work[u];
        if (w.d.T.len >= 0x7FFFFF00u || w.d.Q.len >= 0x7FFFFF00u) { set_error("scaffold longer than 2^31 bases"); return MIMEO_ERR_LIMIT; }
        h_units[u] = w.d;
        max_t = std::max<uint64_t>(max_t, w.d.T.len);
        max_q = std::max<uint64_t>(max_q, w.d.Q.len);
        expect_hits += 13.0 * (double)w.ti.n * (double)w.qi.n / 16777216.0;
        max_unit_hits = std::max(max_unit_hits, 13.0 * (double)w.ti.n * (double)w.qi.n / 16777216.0);
        if (w.d.same) h_selfs_.push_back(u);
    }
    expect_hits_ = expect_hits;
    {   // a batch can be cut in two when at least two of its units launch the heavy kernels (a mirror unit rides with its source)
        uint32_t launching = 0;
        for (uint32_t u = 0; u < nunits; u++) launching += (work[u].ti.n && work[u].qi.n) ? 1u : 0u;
        splittable_ = launching > 1;
    }
    mirror_dst_.clear();
    if (mirror_dst && std::any_of(mirror_dst->begin(), mirror_dst->end(), [](uint32_t m) { return m != NO_MIRROR; })) {
        if (mirror_dst->size() != nunits) { set_error("internal: mirror table of the wrong size"); return MIMEO_ERR_ARG; }
        mirror_dst_ = *mirror_dst;
    }
    ebits_ = bits_for(max_t + SEED_LEN);
    dbits_ = bits_for(max_t + max_q + SEED_LEN);
    if (ebits_ + dbits_ > 63 || (nunits > 1 && bits_for(nunits - 1) + ebits_ + dbits_ > 64)) {
        set_error("internal: batch too large for the follower key");
        return MIMEO_ERR_ARG;
    }
    key_bits_ = std::min(64u, ebits_ + dbits_ + (nunits > 1 ? bits_for(nunits - 1) : 0));
    if (!ev[0]) {
        for (auto &e : ev) HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipEventCreateWithFlags(&side_done, hipEventDisableTiming));
        HIP_TRY(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    }
    // development / test switches: read once per batch, never per launch
    v1_ = getenv("MIMEO_HEAVY") && !strcmp(getenv("MIMEO_HEAVY"), "v1");
    k34_dbg_ = getenv("MIMEO_K34_DEBUG") ? (uint32_t)atoi(getenv("MIMEO_K34_DEBUG")) : 0u;
    {   // the form of K34's first pass (k34_fused.hip; bits 5 and 6 of its switch word): two-segment tiles cut at the middle key
        // (0) or by entry count (32); descriptors written level by level (0) or lane-major behind a prefix sum (64).
        // MIMEO_K34_FORM = level (0) | cut (32) | half (64) | lane (96: the round-2/3 form) picks one (tests, scripts/gpu_k34_ab.py).
        const char *form = getenv("MIMEO_K34_FORM");
        if (form && strcmp(form, "level") && strcmp(form, "cut") && strcmp(form, "half") && strcmp(form, "lane")) {
            set_error(std::string("MIMEO_K34_FORM=") + form + ": not one of level, cut, half, lane");
            return MIMEO_ERR_ARG;
        }
        const uint32_t bits = !form ? K34_FORM_DEFAULT : !strcmp(form, "level") ? 0u : !strcmp(form, "cut") ? 32u : !strcmp(form, "half") ? 64u : 96u;
        k34_dbg_ = (k34_dbg_ & ~96u) | bits;
    }
    k4_variant_ = getenv("MIMEO_K4_VARIANT") ? atoi(getenv("MIMEO_K4_VARIANT")) : 0;
    qw_blocks_ = getenv("MIMEO_QW_BLOCKS") ? (uint32_t)std::max(1, atoi(getenv("MIMEO_QW_BLOCKS"))) : 256u;
    k4_stats_ = getenv("MIMEO_K4_STATS") != nullptr;
    int rc;
    if ((rc = units.reserve((size_t)nunits * sizeof(UnitDesc))) || (rc = ctr.reserve(sizeof(ExtCounters))) ||
        (rc = unit_hits.reserve((size_t)nunits * 8)) || (rc = nsel.reserve(16)) ||
        (rc = tile_hits.reserve(v1_ ? 8 : (size_t)nunits * NTILE * 8)) || (rc = bigcand.reserve((size_t)ENT_BIGCAP * 8)) ||
        (rc = bigacc.reserve((size_t)ENT_BIGCAP * 5 * 8)) ||
        (rc = selfs.reserve((h_selfs_.size() + 1) * 4)))
        return rc;
    HIP_TRY(hipMemcpyAsync(units.p, h_units.data(), (size_t)nunits * sizeof(UnitDesc), hipMemcpyHostToDevice, st));
    if (!h_selfs_.empty()) HIP_TRY(hipMemcpyAsync(selfs.p, h_selfs_.data(), h_selfs_.size() * 4, hipMemcpyHostToDevice, st));
    // Queue capacities from the hit count expected on random sequence; a batch that does not fit (repeat-rich
    // units) is repeated with room for everything the counters saw.  On random sequence 0.65 % of the hits
    // are followers, ~1 % outlive the frame and 1e-4 become candidates.
    const double shrink = getenv("MIMEO_QUEUE_SHRINK") ? atof(getenv("MIMEO_QUEUE_SHRINK")) : 1.0;  // tests: force the rerun
    shrink_ = shrink > 0 ? shrink : 1.0;
    // `boost`: the largest excess over these shares that an earlier batch showed (a repeat-rich genome overflows the
    // first batch once, not every batch)
    // (followers and generic-walk hits: capacity PER SHARD, eight shards, with half as much again for their imbalance)
    cap_f_ = (uint64_t)(expect_hits * 0.02 / 8 * 1.5 * boost_f / shrink) + (uint64_t)(1048576 / shrink) + 64;
    cap_m_ = (uint64_t)(expect_hits * 0.03 / 8 * 1.5 * boost_m / shrink) + (uint64_t)(1048576 / shrink) + 64;
    cap_l_ = (uint64_t)(expect_hits * 0.002 * boost_l / shrink) + (uint64_t)(1048576 / shrink) + 64;
    cap_c_ = (uint64_t)(expect_hits * 0.002 * boost_c / shrink) + (uint64_t)(1048576 / shrink) + 64;
    // the walk queue: a region of eight shards per unit, sized from the unit's expected hits (K34 passes ~4 % of the hits of
    // random sequence on)
    (void)max_unit_hits;
    walk_entries_ = 0;
    walk_cap_u_.assign(nunits, 0);
    for (uint32_t u = 0; u < nunits; u++) {
        const UnitWork &w = work[u];
        if (!w.ti.n || !w.qi.n) continue;
        const double e = 13.0 * (double)w.ti.n * (double)w.qi.n / 16777216.0;
        walk_cap_u_[u] = (uint64_t)(e * 0.08 / 8 * 1.5 * boost_w / shrink) + (uint64_t)(16384 / shrink) + 64;
        walk_entries_ += 8 * walk_cap_u_[u];
    }
    {   // a batch whose queues cannot fit is cut in two by the caller before anything is allocated (a single unit is tried anyway)
        uint64_t budget = 0;
        if ((rc = queue_budget(*this, &budget))) return rc;
        if (getenv("MIMEO_TRACE"))
            fprintf(stderr, "[trace] batch of %u units, %.3g expected hits: queues %.2f GiB of %.2f GiB budget (boosts f %.1f m %.1f l %.1f c %.1f w %.1f; caps f %llu m %llu l %llu c %llu walk entries %llu)\n", nunits, expect_hits,
                    (double)queue_bytes() / (1 << 30), (double)budget / (1 << 30), boost_f, boost_m, boost_l, boost_c, boost_w,
                    (unsigned long long)cap_f_, (unsigned long long)cap_m_, (unsigned long long)cap_l_, (unsigned long long)cap_c_, (unsigned long long)walk_entries_);
        if (splittable_ && queue_bytes() > budget) return MIMEO_ERR_SPLIT;
    }
    if (!mirror_dst_.empty()) {
        if ((rc = mirror.reserve((size_t)nunits * 4 + 16))) return rc;
        HIP_TRY(hipMemcpyAsync(mirror.p, mirror_dst_.data(), (size_t)nunits * 4, hipMemcpyHostToDevice, st));
    }
    if (!q_) q_ = new ExtQueues();
    if ((rc = enqueue_heavy())) return rc;
    started_ = true;
    return 0;
}

// the heavy phase of the batch on the calling thread's stream, with the current capacities
int ExtBatch::enqueue_heavy() {
    hipStream_t st = stream();
    const uint32_t nunits = (uint32_t)w_.size();
    const mimeo_params *p = &p_;
    const uint32_t *tab = group_table_device();
    if (!tab) { set_error("group table upload failed"); return MIMEO_ERR_HIP; }
    ExtQueues &q = *(ExtQueues *)q_;
    memset(&q, 0, sizeof q);
    q.ebits = ebits_; q.dbits = dbits_;
    const uint64_t cap_f = cap_f_, cap_m = cap_m_, cap_l = cap_l_, cap_c = cap_c_;
    const bool v1 = v1_;
    const std::vector<UnitWork> &work = w_;
    const std::vector<uint32_t> &h_selfs = h_selfs_;
    int rc;
    {
        {   // the queues of the batch: slices of one allocation (it grows when a batch needs more, with room to spare, and stays)
            const size_t mf = mirror_dst_.empty() ? 1 : 2;
            const size_t sz[10] = {cap_f * 8 * 8, cap_f * 8 * 4, cap_m * 8 * 8, cap_m * 8 * 4, cap_l * 8, cap_l * 4, cap_c * sizeof(Cand),
                                   cap_c * sizeof(mimeo_hsp) * mf, cap_c * 4 * mf, v1 ? 8 : walk_entries_ * 8 + 8};
            size_t total = 0;
            for (size_t z : sz) total += (z + 255) & ~(size_t)255;
            if (total > arena_q.cap) {
                for (DeviceBuf *d : {&fkey, &fprev, &medq, &medu, &longq, &longu, &cand, &hsps, &hsp_unit, &walkq}) d->release();
                // grow with room to spare when the device has it (the next overflow should not allocate again)
                size_t free_b = 0, total_b = 0;
                HIP_TRY(hipMemGetInfo(&free_b, &total_b));
                const size_t have = free_b + arena_q.cap;
                // the first allocation is what the batch asks for; one that has to grow (an overflow, a larger batch) takes half as much
                // again when the device has it, so that the next one does not allocate (hipMalloc costs ~35 ms per GiB)
                size_t want = arena_q.cap ? total + total / 2 : total;
                if (want + (queue_bytes() - arena_bytes()) + (8ull << 30) > have) want = total;   // the sort arena and the chain scratch come after
                if ((rc = arena_q.reserve(want)) && want > total) rc = arena_q.reserve(total);
                if (rc) return rc == MIMEO_ERR_NOMEM && splittable_ ? MIMEO_ERR_SPLIT : rc;
            }
            size_t off = 0;
            DeviceBuf *bufs[10] = {&fkey, &fprev, &medq, &medu, &longq, &longu, &cand, &hsps, &hsp_unit, &walkq};
            for (int i = 0; i < 10; i++) bufs[i]->set_view(carve(arena_q, &off, sz[i]), sz[i]);
        }
        q.ctr = (ExtCounters *)ctr.p;
        q.cand = (Cand *)cand.p; q.fkey = (uint64_t *)fkey.p; q.fprev = (uint32_t *)fprev.p;
        q.medq = (uint2 *)medq.p; q.medu = (uint32_t *)medu.p; q.longq = (uint2 *)longq.p; q.longu = (uint32_t *)longu.p;
        q.bigcand = (unsigned long long *)bigcand.p; q.bigacc = (unsigned long long *)bigacc.p;
        q.unit_hits = (unsigned long long *)unit_hits.p;
        q.tile_hits = (unsigned long long *)tile_hits.p;
        q.walkq = (uint2 *)walkq.p;
        q.cand_cap = cap_c; q.follow_cap = cap_f; q.med_cap = cap_m; q.long_cap = cap_l; q.walk_cap = 0;
        const UnitDesc *d_units = (const UnitDesc *)units.p;
        HIP_TRY(hipMemsetAsync(ctr.p, 0, sizeof(ExtCounters), st));
        HIP_TRY(hipMemsetAsync(unit_hits.p, 0, (size_t)nunits * 8, st));
        if (!v1) HIP_TRY(hipMemsetAsync(tile_hits.p, 0, (size_t)nunits * NTILE * 8, st));   // K34 adds to it: a tile's pairs, wavefront by wavefront
        HIP_TRY(hipMemsetAsync(nsel.p, 0, 16, st));
        HIP_TRY(hipMemsetAsync(bigacc.p, 0, (size_t)ENT_BIGCAP * 5 * 8, st));
        HIP_TRY(hipEventRecord(ev[0], st));
        // the main diagonals of the self units: one wavefront each, beside the heavy kernels
        if (!h_selfs.empty()) {
            HIP_TRY(hipStreamWaitEvent(side, ev[0], 0));
            hipLaunchKernelGGL(k4_diag0, dim3((uint32_t)h_selfs.size()), dim3(64), 0, side, d_units, (const uint32_t *)selfs.p, q,
                               p->xdrop, p->hspthresh, p->transitions);
            HIP_TRY(hipEventRecord(side_done, side));
        }
        // ---- heavy phase, nothing read back
        if (!v1) {
            // K34 over every unit of the batch in ONE launch (plus its split pass), then the hits it could not dismiss (3-4 % on
            // random sequence, most of them false alarms of its cheap filter) through the sharp filter and the exact walk of
            // round 1's fast kernel, again one launch for all units (grid.y = unit), then the walk-queue statistics
            std::vector<FusedUnit> h_fu;
            bool slim = true;
            uint64_t base = 0;
            for (uint32_t u = 0; u < nunits; u++) {
                const UnitWork &w = work[u];
                if (!w.ti.n || !w.qi.n) continue;
                FusedUnit f;
                memset(&f, 0, sizeof f);
                f.T = w.ti; f.Q = w.qi; f.Tv = w.d.T; f.Qv = w.d.Q; f.unit = u; f.same = w.d.same;
                f.walk_base = base; f.walk_cap = walk_cap_u_[u];
                base += 8 * f.walk_cap;
                slim = slim && !w.d.T.has_n && !w.d.Q.has_n;
                h_fu.push_back(f);
            }
            const uint32_t nactive = (uint32_t)h_fu.size();
            nactive_ = nactive;
            if (nactive) {
                h_funits_ = std::move(h_fu);   // a member: the copy below is asynchronous
                // counters: 8 walk-queue shards per unit, then the split-pass tile counts
                // the split pass's tile lists (number + estimate per slot) and its plan (dense tiles, part bases, splits, counters)
                const size_t slots = (size_t)std::min(nactive, 32768u) * NTILE;
                if ((rc = funits.reserve((size_t)nactive * sizeof(FusedUnit))) || (rc = nwalk_u.reserve((size_t)nactive * 9 * 8)) ||
                    (rc = heavy.reserve((size_t)nactive * NTILE * 12)) || (rc = plan_buf.reserve(slots * 28 + 4096)))
                    return rc == MIMEO_ERR_NOMEM && splittable_ ? MIMEO_ERR_SPLIT : rc;
                q.heavy_est = (unsigned long long *)heavy.p;
                q.heavy = (uint32_t *)(q.heavy_est + (size_t)nactive * NTILE);
                HeavyPlan plan;
                plan.est = q.heavy_est;
                plan.split = (uint4 *)plan_buf.p;                       // slots * 16
                plan.tile = (uint2 *)(plan.split + slots);              // slots * 8
                plan.base = (uint32_t *)(plan.tile + slots);            // (slots + 1) * 4
                plan.ctr = (unsigned int *)(plan.base + slots + 4);     // 3 counters
                plan_ctr_ = plan.ctr;
                q.nwalk_u = (unsigned long long *)nwalk_u.p;
                q.nheavy_u = q.nwalk_u + (size_t)nactive * 8;
                HIP_TRY(hipMemcpyAsync(funits.p, h_funits_.data(), (size_t)nactive * sizeof(FusedUnit), hipMemcpyHostToDevice, st));
                HIP_TRY(hipMemsetAsync(nwalk_u.p, 0, (size_t)nactive * 9 * 8, st));
                while (kev.size() < 2) {   // an event pair around the seed-scan kernel (both passes)
                    hipEvent_t e;
                    HIP_TRY(hipEventCreate(&e));
                    kev.push_back(e);
                }
                HIP_TRY(hipEventRecord(kev[0], st));
                if ((rc = launch_fused_batch((const FusedUnit *)funits.p, nactive, q, plan, p, st, k34_dbg_))) return rc;
                HIP_TRY(hipEventRecord(kev[1], st));
                for (uint32_t u0 = 0; u0 < nactive; u0 += 32768u) {
                    const uint32_t nu = std::min(32768u, nactive - u0);
                    ExtQueues qq = q;
                    qq.nwalk_u = q.nwalk_u + (size_t)u0 * 8;
                    const FusedUnit *fu = (const FusedUnit *)funits.p + u0;
                    // a unit's queue is worked off by at most qw_blocks_ workgroups (their end-of-kernel flushes are same-address
                    // atomics: more workgroups cost more than they bring), fewer when many small units share the launch
                    const uint32_t qblocks = std::max(1u, std::min(qw_blocks_, (8192u + nu - 1) / nu));
                    if (slim) hipLaunchKernelGGL(k4_walk_batch<9>, dim3(qblocks, nu), dim3(FAST_THREADS), 0, st, fu, p->xdrop, p->hspthresh, p->transitions, tab, qq);
                    else hipLaunchKernelGGL(k4_walk_batch<5>, dim3(qblocks, nu), dim3(FAST_THREADS), 0, st, fu, p->xdrop, p->hspthresh, p->transitions, tab, qq);
                }
                hipLaunchKernelGGL(k4_walk_summary, dim3(std::min(256u, (nactive * 8u + 255u) / 256u)), dim3(256), 0, st, (const FusedUnit *)funits.p, nactive,
                                   (const unsigned long long *)q.nwalk_u, q.ctr);
            }
        } else {
            for (uint32_t u = 0; u < nunits; u++) {
                const UnitWork &w = work[u];
                if (!w.ti.n || !w.qi.n) continue;
                // A/B path: materialise the hits (exact count: one round trip per unit), then the round-1 fast kernel
                uint64_t nh = 0;
                if ((rc = join_hits(jc, w.ti, w.qi, p->transitions, hits, &nh, nullptr))) return rc;
                if (!nh) continue;
                uint64_t nb = std::min<uint64_t>((nh + FAST_THREADS - 1) / FAST_THREADS, 1024);
                const int variant = k4_variant_;
                const bool slim = !w.d.T.has_n && !w.d.Q.has_n;
#define K4_LAUNCH(V) hipLaunchKernelGGL(k4_extend_hits<V>, dim3((uint32_t)nb), dim3(FAST_THREADS), 0, st, w.d.T, w.d.Q, (const uint2 *)hits.p, nh, \
                           p->xdrop, p->hspthresh, p->transitions, tab, q, u, (int)w.d.same)
                if (variant == 1) K4_LAUNCH(1);
                else if (variant == 5 || !slim) K4_LAUNCH(5);
                else K4_LAUNCH(9);
#undef K4_LAUNCH
            }
        }
        if (!v1) launch_sum_hits(q, nunits, st);
        HIP_TRY(hipEventRecord(ev[1], st));
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int ExtBatch::finish(uint64_t *nhsp_out, ExtStats *stats) {
    *nhsp_out = 0;
    if (!started_) return 0;
    started_ = false;
    hipStream_t st = stream();
    const uint32_t nunits = (uint32_t)w_.size();
    const mimeo_params *p = &p_;
    const uint32_t *tab = group_table_device();
    ExtQueues &q = *(ExtQueues *)q_;
    const std::vector<UnitWork> &work = w_;
    const std::vector<uint32_t> &h_selfs = h_selfs_;
    const bool v1 = v1_;
    const uint32_t key_bits = key_bits_;
    const double expect_hits = expect_hits_;
    const UnitDesc *d_units = (const UnitDesc *)units.p;
    int rc;
    ExtCounters c;
    uint64_t nf_total = 0, nm_total = 0;
    float ms_heavy = 0, ms_tails = 0, ms_walk = 0, ms_k34 = 0;
    for (int attempt = 0;; attempt++) {
        uint64_t &cap_f = cap_f_, &cap_m = cap_m_, &cap_l = cap_l_, &cap_c = cap_c_;
        HIP_TRY(hipStreamWaitEvent(st, ev[1], 0));   // the heavy phase may have run on another stream
        // ---- tails, once per batch
        HIP_TRY(hipEventRecord(ev[3], st));
        hipLaunchKernelGGL(k4_extend_generic, dim3(1024), dim3(EXT_THREADS), 0, st, d_units, q, p->xdrop, p->hspthresh,
                           p->transitions, tab);
        if (!h_selfs.empty()) HIP_TRY(hipStreamWaitEvent(st, side_done, 0));
        hipLaunchKernelGGL(k4_extend_long, dim3(768), dim3(EXT_THREADS), 0, st, d_units, q, p->xdrop, p->hspthresh, p->transitions);
        HIP_TRY(hipMemcpyAsync(&c, ctr.p, sizeof c, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));   // round trip 1: follower count (sizes the sort), overflow check
        uint64_t nf = 0, nm = 0, maxf = 0, maxm = 0;
        for (int r = 0; r < 8; r++) {
            nf += std::min<uint64_t>(c.nfollow8[r], cap_f);
            nm += std::min<uint64_t>(c.nmed8[r], cap_m);
            maxf = std::max<uint64_t>(maxf, c.nfollow8[r]);
            maxm = std::max<uint64_t>(maxm, c.nmed8[r]);
        }
        nf_total = nf; nm_total = nm;
        bool over = maxf > cap_f || maxm > cap_m || c.nlong > cap_l || c.ncand > cap_c || c.nwalk_over > 1024;
        if (!over && nf) {
            // the sort's second buffers, flags, segment lists and rocPRIM's scratch: slices of the second arena (~41 bytes per follower)
            size_t t1 = 0, t2 = 0;
            rocprim::counting_iterator<uint64_t> iota(0);
            HIP_TRY(rocprim::radix_sort_pairs(nullptr, t1, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)nf, 0, key_bits, st));
            HIP_TRY(rocprim::select(nullptr, t2, iota, (uint8_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nsel.p, (size_t)nf, st));
            const size_t sz[6] = {nf * 8, nf * 4, nf, nf * 8, nf * 8, std::max(t1, t2) + 16};
            size_t total = 0;
            for (size_t z : sz) total += (z + 255) & ~(size_t)255;
            if (total > arena_s.cap) {
                for (DeviceBuf *d : {&fkey2, &fprev2, &flags, &segs, &bigseg, &tmp}) d->release();
                size_t free_b = 0, total_b = 0;
                HIP_TRY(hipMemGetInfo(&free_b, &total_b));
                size_t want = arena_s.cap ? total + total / 2 : total + total / 8;
                if (want + (4ull << 30) > free_b + arena_s.cap) want = total;
                if ((rc = arena_s.reserve(want)) && want > total) rc = arena_s.reserve(total);
                if (rc) return rc == MIMEO_ERR_NOMEM && splittable_ ? MIMEO_ERR_SPLIT : rc;
            }
            size_t off = 0;
            DeviceBuf *bufs[6] = {&fkey2, &fprev2, &flags, &segs, &bigseg, &tmp};
            for (int i = 0; i < 6; i++) bufs[i]->set_view(carve(arena_s, &off, sz[i]), sz[i]);
            // the eight shards gathered into fkey2 / fprev2; the sort writes back into the (now free) shard area
            hipLaunchKernelGGL(k4_compact_followers, dim3(1024), dim3(256), 0, st, q, (uint64_t *)fkey2.p, (uint32_t *)fprev2.p);
            HIP_TRY(rocprim::radix_sort_pairs(tmp.p, t1, (uint64_t *)fkey2.p, (uint64_t *)fkey.p, (uint32_t *)fprev2.p,
                                              (uint32_t *)fprev.p, (size_t)nf, 0, key_bits, st));
            hipLaunchKernelGGL(k4_segment_flags, dim3((uint32_t)((nf + 255) / 256)), dim3(256), 0, st, (const uint64_t *)fkey.p,
                               (const uint32_t *)fprev.p, nf, q, (uint8_t *)flags.p);
            HIP_TRY(rocprim::select(tmp.p, t2, iota, (uint8_t *)flags.p, (uint64_t *)segs.p, (uint64_t *)nsel.p, (size_t)nf, st));
            // segments: at most nf of them; the kernels read the real number from nsel
            hipLaunchKernelGGL(k4_resolve_small, dim3((uint32_t)((nf + EXT_THREADS - 1) / EXT_THREADS)), dim3(EXT_THREADS), 0, st,
                               d_units, q, (const uint64_t *)fkey.p, (const uint32_t *)fprev.p, nf, (const uint64_t *)segs.p,
                               (const uint64_t *)nsel.p, p->xdrop, p->hspthresh, tab, (uint64_t *)bigseg.p);
            hipLaunchKernelGGL(k4_resolve_segments, dim3(1024), dim3(EXT_THREADS), 0, st, d_units, q, (const uint64_t *)fkey.p,
                               (const uint32_t *)fprev.p, nf, (const uint64_t *)segs.p, (const uint64_t *)nsel.p,
                               (const uint64_t *)bigseg.p, p->xdrop, p->hspthresh, p->transitions);
        }
        if (!over) {
            hipLaunchKernelGGL(k4_entropy, dim3(1536), dim3(EXT_THREADS), 0, st, d_units, q, p->hspthresh, p->entropy,
                               (mimeo_hsp *)hsps.p, (uint32_t *)hsp_unit.p);
            hipLaunchKernelGGL(k4_entropy_big, dim3(128, 16), dim3(256), 0, st, d_units, q);
            hipLaunchKernelGGL(k4_entropy_big_finish, dim3(16), dim3(256), 0, st, q, p->hspthresh, p->entropy, (mimeo_hsp *)hsps.p,
                               (uint32_t *)hsp_unit.p);
            HIP_TRY(hipEventRecord(ev[2], st));
            HIP_TRY(hipMemcpyAsync(&c, ctr.p, sizeof c, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));   // round trip 2: HSP count; the resolution may have added candidates
            over = c.ncand > cap_c;
        }
        HIP_TRY(hipGetLastError());
        if (k34_dbg_ & 8)
            fprintf(stderr, "[k34] passed on because: left stop unproven %llu, right %llu, bound %llu, alarm %llu; ONLY alarm %llu, only left %llu, only right %llu, only bound %llu\n",
                    c.dbg[0], c.dbg[1], c.dbg[2], c.dbg[3], c.dbg[4], c.dbg[5], c.dbg[6], c.dbg[7]);
        if (k4_stats_ && plan_ctr_ && nactive_) {
            unsigned int pc[3] = {0, 0, 0};
            HIP_TRY(hipMemcpy(pc, plan_ctr_, 12, hipMemcpyDeviceToHost));
            fprintf(stderr, "[k34] split pass: %u tiles listed of %u, %u parts of ~131072 hits\n", pc[1], nactive_ * (unsigned)NTILE, pc[2]);
        }
        if (k4_stats_)
            fprintf(stderr, "[k4] units %u walk queue %llu walked %llu generic %llu long %llu followers %llu candidates %llu hsps %llu%s\n", nunits,
                    c.nwalk_total, c.nwalked, (unsigned long long)nm, c.nlong, (unsigned long long)nf, c.ncand, c.nhsp, over ? "  (queue overflow: batch repeated)" : "");
        if (!over) {
            float a = 0, b = 0, w = 0;
            HIP_TRY(hipEventElapsedTime(&a, ev[0], ev[1]));
            HIP_TRY(hipEventElapsedTime(&b, ev[1], ev[2]));
            HIP_TRY(hipEventElapsedTime(&w, ev[1], ev[3]));
            ms_heavy += a;
            ms_tails += b;
            ms_walk += w;
            if (!v1 && nactive_) {
                float t = 0;
                HIP_TRY(hipEventElapsedTime(&t, kev[0], kev[1]));
                ms_k34 += t;
            }
            break;
        }
        // (a walk queue that overflowed hides followers and candidates of the hits it dropped: the repeat may overflow those in turn)
        if (attempt >= 4) { set_error("extension queues overflowed four times in a row"); return MIMEO_ERR_LIMIT; }
        // room for what the counters saw, and half as much again: the repeated tails may add candidates of their own
        cap_f = std::max<uint64_t>(cap_f, maxf + maxf / 2 + 1024);
        cap_m = std::max<uint64_t>(cap_m, maxm + maxm / 2 + 1024);
        cap_l = std::max<uint64_t>(cap_l, c.nlong + c.nlong / 2 + 1024);
        cap_c = std::max<uint64_t>(cap_c, 2 * c.ncand + 65536);
        if (c.nwalk_over > 1024) {   // every unit's walk-queue region grows by what the fullest shard lacked, and a quarter
            const double grow = 1.25 * (double)c.nwalk_over / 1024.0;
            walk_entries_ = 0;
            for (auto &w : walk_cap_u_) { if (w) w = (uint64_t)((double)w * grow) + 1024; walk_entries_ += 8 * w; }
        }   // (boost_w learns from the batch's real share of walked hits at the end of finish(), whatever the capacities were)
        if (stats) stats->reruns++;
        // what this batch showed goes into the sizing of the next ones whatever happens to it now
        if (expect_hits > 1e6) {
            boost_f = std::min(4096.0, std::max(boost_f, 1.5 * (double)(8 * maxf) / (0.02 * expect_hits)));
            boost_m = std::min(4096.0, std::max(boost_m, 1.5 * (double)(8 * maxm) / (0.03 * expect_hits)));
            boost_l = std::min(4096.0, std::max(boost_l, 1.5 * (double)c.nlong / (0.002 * expect_hits)));
            boost_c = std::min(4096.0, std::max(boost_c, 1.5 * (double)c.ncand / (0.002 * expect_hits)));
            boost_w = std::min(4096.0, std::max(boost_w, 1.5 * (double)c.nwalk_total / (0.08 * expect_hits)));   // the counters count beyond the capacities
            // ... and the fullest shard of the worst unit, not the batch's average, is what has to fit (microsatellites are not
            // spread evenly over the pairs); shrink_ (tests) made the capacities smaller, not the need larger
            boost_w = std::min(4096.0, std::max(boost_w, 1.5 * boost_w * ((double)c.nwalk_over / 1024.0) / shrink_));
        }
        {
            uint64_t budget = 0;
            if ((rc = queue_budget(*this, &budget))) return rc;
            if (getenv("MIMEO_TRACE"))
                fprintf(stderr, "[trace] overflow of a batch of %u units: followers %llu (fullest shard), generic %llu, long %llu, cand %llu, walk %.2fx; queues would take %.2f GiB of %.2f GiB\n",
                        nunits, (unsigned long long)maxf, (unsigned long long)maxm, c.nlong, c.ncand, (double)c.nwalk_over / 1024.0, (double)queue_bytes() / (1 << 30), (double)budget / (1 << 30));
            if (splittable_ && queue_bytes() > budget) return MIMEO_ERR_SPLIT;   // the caller cuts the batch in two
        }
        if ((rc = enqueue_heavy())) return rc;   // the batch again, on this stream, with room
    }
    uint64_t nhsp = c.nhsp;
    if (!mirror_dst_.empty() && nhsp) {   // shared plus strand: the mirror units receive their (transposed) HSPs
        unsigned long long *mctr = (unsigned long long *)((char *)mirror.p + (size_t)nunits * 4 + (8 - ((size_t)nunits * 4) % 8) % 8);
        HIP_TRY(hipMemsetAsync(mctr, 0, 8, st));
        hipLaunchKernelGGL(k4_mirror_hsps, dim3((uint32_t)std::min<uint64_t>(1024, (nhsp + 255) / 256)), dim3(256), 0, st, (mimeo_hsp *)hsps.p,
                           (uint32_t *)hsp_unit.p, nhsp, (const uint32_t *)mirror.p, mctr);
        unsigned long long added = 0;
        HIP_TRY(hipMemcpyAsync(&added, mctr, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        nhsp += added;
    }
    h_unit_hits.resize(nunits);
    HIP_TRY(hipMemcpy(h_unit_hits.data(), unit_hits.p, (size_t)nunits * 8, hipMemcpyDeviceToHost));
    *nhsp_out = nhsp;
    if (expect_hits > 1e6) {
        const double e = expect_hits;
        boost_f = std::min(4096.0, std::max(boost_f, 1.5 * (double)nf_total / (0.02 * e)));
        boost_m = std::min(4096.0, std::max(boost_m, 1.5 * (double)nm_total / (0.03 * e)));
        boost_l = std::min(4096.0, std::max(boost_l, 1.5 * (double)c.nlong / (0.002 * e)));
        boost_c = std::min(4096.0, std::max(boost_c, 1.5 * (double)c.ncand / (0.002 * e)));
        boost_w = std::min(4096.0, std::max(boost_w, 1.5 * (double)c.nwalk_total / (0.08 * e)));
        boost_w = std::min(4096.0, std::max(boost_w, 1.5 * boost_w * ((double)c.nwalk_over / 1024.0) / shrink_));   // the fullest shard of the worst unit
    }
    if (stats) {
        for (uint32_t u = 0; u < nunits; u++) {
            if (!(work[u].ti.n && work[u].qi.n)) continue;   // a mirror unit's seed scan is never run: it counts no hits and no bytes
            stats->seed_hits += h_unit_hits[u];
            const uint64_t Lq = work[u].d.Q.len, H = h_unit_hits[u];
            // B_scan = ceil(Lq/4) + 8*W*(Lq-18) + 4*H + 8*H, W = 13
            stats->scan_bytes_algorithmic += (Lq + 3) / 4 + (Lq > 18 ? 8ull * 13ull * (Lq - 18) : 0) + 12ull * H;
            // compulsory traffic of the fused kernel: both offset arrays, positions and frames of both sides once
            stats->scan_bytes_kernel += 2ull * 4ull * ((uint64_t)NBUCKET + 1) + 52ull * ((uint64_t)work[u].ti.n + work[u].qi.n);
            stats->heavy_launches++;   // units: the seed-scan kernel takes a batch of them per launch
        }
        if (!v1 && nactive_) stats->heavy_kernel_launches++;
        stats->walked += c.nwalked;
        stats->walk_queue += c.nwalk_total;
        stats->ms_walk += ms_walk;
        stats->ms_k34 += v1 ? ms_heavy : ms_k34;
        stats->followers += nf_total;
        stats->candidates += c.ncand;
        stats->ms_heavy += ms_heavy;
        stats->ms_tails += ms_tails;
    }
    return 0;
}

}  // namespace mimeo

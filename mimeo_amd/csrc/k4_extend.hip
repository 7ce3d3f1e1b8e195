// K4 — gap-free x-drop extension of seed hits into HSPs with lastz's per-diagonal "already
// extended" suppression, --entropy and --hspthresh (reference call site
// src/mimeo/wrappers.py:1031-1032 `--gfextend --entropy --hspthresh=H`).
//
// lastz resolves the suppression sequentially (one diagonal-extent array updated while the
// query is scanned).  The same result is obtained here without ordering the hit flood
// (DESIGN.md §4, K4):
//   * k4_extend_hits — one lane per hit, stateless.  While walking left from the seed end the
//     lane also looks, bit-parallel, for an earlier seed hit on its own diagonal whose seed
//     end lies at a position the walk reached.  If there is none the hit is a HEAD: no earlier
//     extension can cover it, so it is certainly extended by the sequential process; the lane
//     finishes the extension and emits a candidate HSP if it scores >= hspthresh.  Otherwise
//     the hit is a FOLLOWER and only (diagonal, seed end, nearest earlier seed end) is kept.
//   * followers are sorted by (diagonal, seed end); a run whose members each name their
//     predecessor is a segment owned by the head in front of it; k4_resolve_segments replays
//     lastz's rule inside each segment with one wavefront (extend, skip everything whose seed
//     end <= reach, extend the next one, ...).
//   * walks longer than LONG_CAP bases are finished by k4_extend_long, one wavefront per hit,
//     64 bases per step with wave-level prefix sums.
//   * k4_entropy applies the entropy adjustment to candidates and the threshold.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>

#include <rocprim/rocprim.hpp>

#include "device_util.h"

namespace mimeo {

constexpr uint32_t CARE19 = 0x7A997u;  // care positions of 1110100110010101111 (bit i = offset i)
constexpr int EXT_THREADS = 256;
constexpr int LONG_WINDOWS = 8;  // per-lane walks give up after 8*32 bases per direction

struct ExtCounters {
    unsigned long long ncand, nfollow, nlong, nhsp, nmed, nbig, nwalked;  // nwalked: hits the pre-filter let through
};

struct Cand {
    uint32_t tstart, qstart, len;
    int32_t raw;  // RAW_SATURATED: the score does not fit (a gap-free segment beyond ~21 Mbp); k4_entropy recounts it in 64 bits
};
constexpr int32_t RAW_SATURATED = 0x7FFFFFFF;

// is there a seed hit whose 19-window starts at target position p (query p - d)?
__device__ __forceinline__ bool seed_hit_at(const StrandView &T, const StrandView &Q, int32_t p, int32_t d,
                                            int transitions) {
    int32_t pq = p - d;
    const Win32 tw = win32(T, p), qw = win32(Q, pq);
    if (!((seedvalid32(T, p, tw.sv) & qw.sv) & 1u)) return false;
    uint32_t dl = (tw.lo ^ qw.lo) & CARE19;
    uint32_t dh = (tw.hi ^ qw.hi) & CARE19;
    if (!transitions) return (dl | dh) == 0;
    return dl == 0 && __popc(dh) <= 1;
}

// ---- K4a: one lane per hit --------------------------------------------------------------
// The walks advance four bases at a time through a 4096-entry LDS table indexed by the
// (dl, dh, cg) bits of the four bases in walk order; an entry packs the group's score sum S,
// its best prefix M (and where), and its lowest prefix mn (10 bits each).  Because four
// HOXD70 columns can move the running score by at most 500 < xdrop, a group cannot both set a
// new best and trigger the x-drop: "run + mn < best - xdrop" is exactly "the walk stops in this
// group", and otherwise the group is applied in one step.  Groups holding an N, an earlier seed
// hit, or the sequence end fall back to single bases.
constexpr int GROUP_TAB = 4096;
constexpr int QCAP = 64;  // per-wave staging capacity of the K4a output queues (= the most one iteration adds)
constexpr int FAST_THREADS = 512;  // K4a fast kernel: 8 wavefronts share one copy of the group table

static inline int host_sub(int dl, int dh, int cg) {
    static const int lo[4] = {91, 100, -31, -31}, hi[4] = {-114, -114, -123, -125};
    return (dl ? hi : lo)[(dh << 1) | cg];
}
static void build_group_table(uint32_t *tab) {
    for (int idx = 0; idx < GROUP_TAB; idx++) {
        int p = 0, M = -100000, mn = 100000, posM = 0;
        for (int k = 0; k < 4; k++) {
            p += host_sub((idx >> k) & 1, (idx >> (4 + k)) & 1, (idx >> (8 + k)) & 1);
            if (p > M) { M = p; posM = k; }
            if (p < mn) mn = p;
        }
        tab[idx] = ((uint32_t)p & 0x3FFu) | (((uint32_t)M & 0x3FFu) << 10) | (((uint32_t)mn & 0x3FFu) << 20) |
                   ((uint32_t)posM << 30);
    }
}

struct WalkState {
    int32_t run, best;
    uint32_t bk, k;  // steps at the best prefix, steps done
    bool done, found;
    uint32_t found_step;  // step index (1-based) whose boundary carries an earlier seed hit
};

// up to 32 steps of a walk whose masks are in step order (bit s <-> step s of this window)
__device__ __forceinline__ void walk_window(const uint32_t *__restrict__ tab, WalkState &w, uint32_t mdl, uint32_t mdh,
                                            uint32_t mcg, uint32_t mnn, uint32_t mH, uint32_t limit, int xdrop,
                                            uint32_t start = 0) {
    for (uint32_t pos = start; pos < 32;) {
        const uint32_t rem = limit - w.k;
        if (rem == 0) { w.done = true; return; }
        if (rem >= 4 && pos <= 28 && !(((mnn | mH) >> pos) & 0xFu)) {
            const uint32_t idx = ((mdl >> pos) & 0xFu) | (((mdh >> pos) & 0xFu) << 4) | (((mcg >> pos) & 0xFu) << 8);
            const uint32_t e = tab[idx];
            const int32_t S = ((int32_t)(e << 22)) >> 22, M = ((int32_t)(e << 12)) >> 22, mn = ((int32_t)(e << 2)) >> 22;
            if (w.run + mn < w.best - xdrop) { w.done = true; return; }
            if (w.run + M > w.best) { w.best = w.run + M; w.bk = w.k + (e >> 30) + 1; }
            w.run += S;
            w.k += 4;
            pos += 4;
        } else {
            w.k++;
            w.run += sub_score((mdl >> pos) & 1u, (mdh >> pos) & 1u, (mcg >> pos) & 1u, (mnn >> pos) & 1u);
            if (w.run > w.best) { w.best = w.run; w.bk = w.k; }
            if (w.run < w.best - xdrop) { w.done = true; return; }
            if ((mH >> pos) & 1u) { w.found = true; w.found_step = w.k; w.done = true; return; }
            pos++;
        }
    }
}

__global__ __launch_bounds__(EXT_THREADS) void k4_extend_generic(StrandView T, StrandView Q,
                                                              const uint2 *__restrict__ hits, uint64_t nhits_arg,
                                                              const unsigned long long *__restrict__ nhits_dev,
                                                              int xdrop, int hspthresh, int transitions,
                                                              const uint32_t *__restrict__ group_tab,
                                                              ExtCounters *__restrict__ ctr, Cand *__restrict__ cand,
                                                              uint64_t cand_cap, uint64_t *__restrict__ fkey,
                                                              uint32_t *__restrict__ fprev,
                                                              uint2 *__restrict__ longq, int skip_diag0) {
    __shared__ uint32_t tab[GROUP_TAB];
    for (int i = threadIdx.x; i < GROUP_TAB; i += EXT_THREADS) tab[i] = group_tab[i];
    __syncthreads();
    const uint64_t nhits = nhits_dev ? (uint64_t)*nhits_dev : nhits_arg;  // count produced by an earlier kernel
    for (uint64_t gid = (uint64_t)blockIdx.x * EXT_THREADS + threadIdx.x; gid < nhits;
         gid += (uint64_t)gridDim.x * EXT_THREADS) {
        const uint2 h = hits[gid];
        if (skip_diag0 && h.x == h.y) continue;
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
        if (is_long) {
            unsigned long long i = atomicAdd(&ctr->nlong, 1ull);
            longq[i] = h;
            continue;
        }
        if (L.found) {
            unsigned long long i = atomicAdd(&ctr->nfollow, 1ull);
            fkey[i] = ((uint64_t)(uint32_t)(d + (int32_t)Q.len) << 32) | (uint32_t)et;
            fprev[i] = (uint32_t)et - L.found_step;  // position of the base just summed = that seed's end
            continue;
        }
        // ---- right walk
        WalkState R{0, 0, 0, 0, false, false, 0};
        const uint32_t maxr = min(T.len - (uint32_t)et, Q.len - (uint32_t)eq);
        for (int win = 0; !R.done; win++) {
            if (win == LONG_WINDOWS) { is_long = true; break; }
            const int32_t P = et + 32 * win, Pq = P - d;
            const Win32 tw = win32(T, P), qw = win32(Q, Pq);
            walk_window(tab, R, tw.lo ^ qw.lo, tw.hi ^ qw.hi, tw.lo ^ tw.hi, tw.nm | qw.nm, 0u, maxr, xdrop);
        }
        if (is_long) {
            unsigned long long i = atomicAdd(&ctr->nlong, 1ull);
            longq[i] = h;
            continue;
        }
        const int32_t score = L.best + R.best;
        if (score >= hspthresh) {
            unsigned long long i = atomicAdd(&ctr->ncand, 1ull);
            if (i < cand_cap) cand[i] = Cand{(uint32_t)et - L.bk, (uint32_t)eq - L.bk, L.bk + R.bk, score};
        }
    }
}

// The same 32 steps as walk_window, as straight-line code without per-lane branches: the eight table
// groups are applied to running values unconditionally, and the state in front of the first group that
// cannot be applied — x-drop inside it, an N / earlier seed hit / the sequence end in it, or a walk that
// was already finished — is kept aside (one select per value and group).  A lane stopped by a blocked
// group finishes the window in walk_window (rare; taken under a wave-uniform branch).
__device__ __forceinline__ void walk_window_pred(const uint32_t *__restrict__ tab, WalkState &w, uint32_t mdl,
                                                 uint32_t mdh, uint32_t mcg, uint32_t mnn, uint32_t mH, uint32_t limit,
                                                 int xdrop) {
    const uint32_t blocked = mnn | mH;
    // the eight table entries depend only on the masks: fetch them back to back, then run the
    // dependent score arithmetic on registers
    uint32_t ent[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int pos = 4 * c;
        ent[c] = tab[((mdl >> pos) & 0xFu) | (((mdh >> pos) & 0xFu) << 4) | (((mcg >> pos) & 0xFu) << 8)];
    }
    uint32_t nz = blocked | (blocked >> 1);
    nz |= nz >> 2;                                // bit 4c: group c holds a blocked step ...
    const uint32_t ng = (limit - w.k) >> 2;       // ... or does not fit below the limit any more (one test per
    nz |= ng >= 8u ? 0u : (0xFFFFFFFFu << (4u * ng));  // group instead of two: compares issue at half rate)
    int32_t R = w.run, B = w.best, sR = R, sB = B;
    uint32_t BK = 0, sBK = 0, sC = 0;             // BK: steps at the best prefix relative to w.k (0 = unchanged)
    bool stopped = w.done, slow = false, brkdone = false;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const uint32_t e = ent[c];
        const int32_t S = ((int32_t)(e << 22)) >> 22, M = ((int32_t)(e << 12)) >> 22, mn = ((int32_t)(e << 2)) >> 22;
        const bool blk = (nz >> (4 * c)) & 1u;
        const bool brk = R + mn + xdrop < B;
        const bool first = (blk || brk) && !stopped;
        sR = first ? R : sR;
        sB = first ? B : sB;
        sBK = first ? BK : sBK;
        sC = first ? (uint32_t)c : sC;
        slow = slow || (first && blk);            // a blocked group wins over the x-drop test, as in walk_window
        brkdone = brkdone || (first && !blk);
        stopped = stopped || blk || brk;
        const int32_t cand = R + M;
        BK = cand > B ? (e >> 30) + (uint32_t)(4 * c + 1) : BK;
        B = max(B, cand);
        R += S;
    }
    if (!stopped) { sR = R; sB = B; sBK = BK; sC = 8; }
    w.run = sR;
    w.best = sB;
    w.bk = sBK ? w.k + sBK : w.bk;
    w.k += 4u * sC;
    w.done = w.done || brkdone;
    if (__ballot(slow)) {
        if (slow) walk_window(tab, w, mdl, mdh, mcg, mnn, mH, limit, xdrop, 4u * sC);
    }
}

// ---- K4a fast path: the whole neighbourhood of a hit is loaded once ---------------------------
// A random hit dies within ~45 bases to the left of its seed end and ~20 to the right.  The fast
// kernel therefore loads, per lane, six interleaved words of the target around the seed start and
// seven of the query (13 16-byte loads issued back to back, 2-3 cache lines per sequence), shifts
// the query into the target's bit frame once, and serves two left windows (64 bases) and two
// right windows (64 bases) from registers with compile-time word indices.  A walk that is still
// alive beyond that goes to the generic kernel through a queue (a few per cent of random hits,
// plus hits inside real similarity).
__device__ __forceinline__ uint32_t ext32(uint32_t a, uint32_t b, uint32_t c, uint32_t sh) {
    return sh < 32 ? __builtin_amdgcn_alignbit(b, a, sh) : __builtin_amdgcn_alignbit(c, b, sh - 32);
}

struct Frame {
    uint32_t dl[6], dh[6], cg[6], nn[6], st[6], sq[6];  // difference / class planes in the target's bit frame
};

// masks of left window WIN in step order (bit s <-> step 32*WIN + s of the left walk, which starts at the
// seed end and runs through the seed): difference / class planes, N, and H = "an earlier seed hit of this
// diagonal ends at the boundary this step reaches"
struct WinMasks { uint32_t dl, dh, cg, nn, H; };

template <int WIN>
__device__ __forceinline__ WinMasks left_masks(const Frame &F, uint32_t bt, int transitions) {
    constexpr int b = 1 - WIN;  // word holding seed start - 32*(WIN+1)
    const uint32_t dllo = __builtin_amdgcn_alignbit(F.dl[b + 1], F.dl[b], bt), dlhi = __builtin_amdgcn_alignbit(F.dl[b + 2], F.dl[b + 1], bt);
    const uint32_t dhlo = __builtin_amdgcn_alignbit(F.dh[b + 1], F.dh[b], bt), dhhi = __builtin_amdgcn_alignbit(F.dh[b + 2], F.dh[b + 1], bt);
    const uint32_t cglo = __builtin_amdgcn_alignbit(F.cg[b + 1], F.cg[b], bt), cghi = __builtin_amdgcn_alignbit(F.cg[b + 2], F.cg[b + 1], bt);
    const uint32_t nnlo = __builtin_amdgcn_alignbit(F.nn[b + 1], F.nn[b], bt), nnhi = __builtin_amdgcn_alignbit(F.nn[b + 2], F.nn[b + 1], bt);
    const uint32_t nlo = dllo | dhlo, nhi = dlhi | dhhi;
    uint32_t ones = 0, twos = 0, tv = 0;
#pragma unroll
    for (int c = 0; c < SEED_LEN; c++) {
        if (!((CARE19 >> c) & 1u)) continue;
        const uint32_t v = c ? __builtin_amdgcn_alignbit(nhi, nlo, c) : nlo;
        twos |= ones & v;
        ones |= v;
        tv |= c ? __builtin_amdgcn_alignbit(dlhi, dllo, c) : dllo;
    }
    const uint32_t bad = transitions ? (twos | tv) : ones;
    const uint32_t H = ~bad & __builtin_amdgcn_alignbit(F.st[b + 1], F.st[b], bt) & __builtin_amdgcn_alignbit(F.sq[b + 1], F.sq[b], bt);
    WinMasks m;
    m.dl = __brev(__builtin_amdgcn_alignbit(dlhi, dllo, SEED_LEN));
    m.dh = __brev(__builtin_amdgcn_alignbit(dhhi, dhlo, SEED_LEN));
    m.cg = __brev(__builtin_amdgcn_alignbit(cghi, cglo, SEED_LEN));
    m.nn = __brev(__builtin_amdgcn_alignbit(nnhi, nnlo, SEED_LEN));
    m.H = __brev(H);
    return m;
}

template <int WIN>
__device__ __forceinline__ void left_window(const uint32_t *__restrict__ tab, const Frame &F, uint32_t bt,
                                            int transitions, WalkState &L, uint32_t maxl, int xdrop) {
    const WinMasks m = left_masks<WIN>(F, bt, transitions);
    walk_window_pred(tab, L, m.dl, m.dh, m.cg, m.nn, m.H, maxl, xdrop);
}

// the neighbourhood of a hit: six interleaved target words from two words in front of the seed start, seven of
// the query, brought into the target's bit frame
__device__ __forceinline__ void load_frame(const StrandView &T, const StrandView &Q, const uint2 h, Frame &F) {
    const uint32_t bt = h.x & 31u, bq = h.y & 31u, sh = (bq - bt) & 31u;
    const int32_t wt = (int32_t)(h.x >> 5) - 2, wq = (int32_t)(h.y >> 5) - 2 - (bq < bt ? 1 : 0);
    uint4 tw[6], qw[7];
#pragma unroll
    for (int k = 0; k < 6; k++) tw[k] = T.pw[wt + k];
#pragma unroll
    for (int k = 0; k < 7; k++) qw[k] = Q.pw[wq + k];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const uint32_t qlo = __builtin_amdgcn_alignbit(qw[k + 1].x, qw[k].x, sh), qhi = __builtin_amdgcn_alignbit(qw[k + 1].y, qw[k].y, sh);
        F.dl[k] = tw[k].x ^ qlo;
        F.dh[k] = tw[k].y ^ qhi;
        F.cg[k] = tw[k].x ^ tw[k].y;
        F.nn[k] = tw[k].z | __builtin_amdgcn_alignbit(qw[k + 1].z, qw[k].z, sh);
        F.st[k] = T.svt ? T.svt[wt + k] : tw[k].w;
        F.sq[k] = __builtin_amdgcn_alignbit(qw[k + 1].w, qw[k].w, sh);
    }
}

// the score planes of the same neighbourhood from the two-plane copy (strands without N only): half the bytes
__device__ __forceinline__ void load_frame_slim(const StrandView &T, const StrandView &Q, const uint2 h, Frame &F) {
    const uint32_t bt = h.x & 31u, bq = h.y & 31u, sh = (bq - bt) & 31u;
    const int32_t wt = (int32_t)(h.x >> 5) - 2, wq = (int32_t)(h.y >> 5) - 2 - (bq < bt ? 1 : 0);
    uint2 tw[6], qw[7];
#pragma unroll
    for (int k = 0; k < 6; k++) tw[k] = T.p2[wt + k];
#pragma unroll
    for (int k = 0; k < 7; k++) qw[k] = Q.p2[wq + k];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        F.dl[k] = tw[k].x ^ __builtin_amdgcn_alignbit(qw[k + 1].x, qw[k].x, sh);
        F.dh[k] = tw[k].y ^ __builtin_amdgcn_alignbit(qw[k + 1].y, qw[k].y, sh);
        F.cg[k] = tw[k].x ^ tw[k].y;
        F.nn[k] = 0;
    }
}

// ---- K4a pre-filter: most seed hits of unrelated sequence are isolated and die at once -------------------
// A hit can be dropped without walking it when all of this holds inside its frame (64 steps to the left of the
// seed end, 64 to the right):
//   * no N, and no earlier seed hit of the diagonal ends at any boundary of the 64 left steps (so whatever the
//     left walk reaches, the hit is a HEAD: nothing can cover it);
//   * both walks provably stop inside their 64 steps: at some checkpoint n (a multiple of STEP) the prefix
//     score U_n is more than xdrop below a lower bound of the prefix score at an earlier checkpoint — the
//     running best is at least that, so the x-drop rule has fired at or before step n (or the sequence ended:
//     the zero padding behind a sequence end only adds matches, which loosen both bounds);
//   * an upper bound of best(left) + best(right) is below hspthresh: the best prefix inside a block is at most
//     the score at the block's start plus 100 per identical column in it.
// Such a hit yields nothing: the full walk would classify it as a head, finish both walks inside the frame and
// find a score below the threshold.  Everything else is queued (per wavefront, in LDS) and walked exactly, 64
// hits at a time.  Prefix scores come from popcounts of the class masks: with n columns, t transitions
// (dl=0,dh=1), v transversions (dl=1), a = C/G matches, b = transversions with dh=1 the HOXD70 sum is
// 91 n + 9 a - 122 t - 205 v - 9 b - 2 c, c = the C<->G transversions among b: U drops the last term (upper
// bound), the lower bound subtracts 2 b.
struct Bound {
    int32_t U, lomax, ub, nb;
    bool stop;
};
// 32 steps of a walk; REV: step s is bit 31 - s of the masks (left windows in position order), else bit s
template <int STEP, bool REV, int NBLOCKS = 32 / STEP>
__device__ __forceinline__ void bound_window(Bound &B, uint32_t mdl, uint32_t mdh, uint32_t mcg, int xdrop) {
    const uint32_t v = mdl, t = ~mdl & mdh, a = ~(mdl | mdh) & mcg, b = mdl & mdh;
#pragma unroll
    for (int j = 0; j < NBLOCKS; j++) {
        const uint32_t low = (1u << (STEP & 31)) - 1u;
        const uint32_t bm = REV ? (low << (32 - STEP * (j + 1))) : (low << (STEP * j));
        const int32_t dv = __popc(v & bm), dt = __popc(t & bm), da = __popc(a & bm), db = __popc(b & bm);
        B.ub = max(B.ub, B.U + 100 * (STEP - dv - dt));
        B.U += 91 * STEP + 9 * da - 122 * dt - 205 * dv - 9 * db;
        B.nb += db;
        B.stop = B.stop || (B.U + xdrop < B.lomax);
        B.lomax = max(B.lomax, B.U - 2 * B.nb);
    }
}

// left window WIN for the pre-filter: the three score planes in position order (bit 31 = the window's first
// step) and a SUPERSET of the boundaries that carry an earlier seed hit — eight of the twelve care positions,
// no seed-validity planes: a false alarm only sends the hit to the exact walk
constexpr uint32_t CARE8 = 0x2997u;  // offsets 0 1 2 4 7 8 11 13 of CARE19
struct FilterMasks {
    uint32_t dl, dh, cg, H;
    uint32_t dl2, dh2, cg2, H2;  // WIN == 1 only: steps 64..79 (bits 31..16) and the boundaries they reach
};
constexpr uint32_t CARE8_HIGH = 0x7A980u;  // offsets 7 8 11 13 15 16 17 18 of CARE19: the part of a window nearest the frame
template <int WIN>
__device__ __forceinline__ FilterMasks filter_left(const Frame &F, uint32_t bt, int transitions) {
    constexpr int b = 1 - WIN;
    const uint32_t dllo = __builtin_amdgcn_alignbit(F.dl[b + 1], F.dl[b], bt), dlhi = __builtin_amdgcn_alignbit(F.dl[b + 2], F.dl[b + 1], bt);
    const uint32_t dhlo = __builtin_amdgcn_alignbit(F.dh[b + 1], F.dh[b], bt), dhhi = __builtin_amdgcn_alignbit(F.dh[b + 2], F.dh[b + 1], bt);
    const uint32_t cglo = __builtin_amdgcn_alignbit(F.cg[b + 1], F.cg[b], bt), cghi = __builtin_amdgcn_alignbit(F.cg[b + 2], F.cg[b + 1], bt);
    const uint32_t nlo = dllo | dhlo, nhi = dlhi | dhhi;
    uint32_t ones = 0, twos = 0, tv = 0;
#pragma unroll
    for (int c = 0; c < SEED_LEN; c++) {
        if (!((CARE8 >> c) & 1u)) continue;
        const uint32_t v = c ? __builtin_amdgcn_alignbit(nhi, nlo, c) : nlo;
        twos |= ones & v;
        ones |= v;
        tv |= c ? __builtin_amdgcn_alignbit(dlhi, dllo, c) : dllo;
    }
    FilterMasks m;
    m.H = ~(transitions ? (twos | tv) : ones);
    m.dl = __builtin_amdgcn_alignbit(dlhi, dllo, SEED_LEN);
    m.dh = __builtin_amdgcn_alignbit(dhhi, dhlo, SEED_LEN);
    m.cg = __builtin_amdgcn_alignbit(cghi, cglo, SEED_LEN);
    m.dl2 = m.dh2 = m.cg2 = m.H2 = 0;
    if (WIN == 1) {
        // steps 64..79 of the left walk are bits 18..3 of the low words (position seed start - 64 + bit)
        m.dl2 = dllo << 13; m.dh2 = dhlo << 13; m.cg2 = cglo << 13;
        // seed windows that END where those steps arrive start 1..16 bases in front of the low words: the word
        // before them is only partly inside the frame (its low 32 - bt bits are not: taken as identical columns,
        // which can only add alarms), and the eight care positions nearest the frame are tested
        const uint32_t dlm = __builtin_amdgcn_alignbit(F.dl[0], 0u, bt), dhm = __builtin_amdgcn_alignbit(F.dh[0], 0u, bt);
        const uint32_t nm = dlm | dhm;
        uint32_t o2 = 0, w2 = 0, t2 = 0;
#pragma unroll
        for (int c = 0; c < SEED_LEN; c++) {
            if (!((CARE8_HIGH >> c) & 1u)) continue;
            const uint32_t v = __builtin_amdgcn_alignbit(nlo, nm, c);
            w2 |= o2 & v;
            o2 |= v;
            t2 |= __builtin_amdgcn_alignbit(dllo, dlm, c);
        }
        m.H2 = ~(transitions ? (w2 | t2) : o2) & 0xFFFF0000u;  // window starts 16..1 bases in front of the low words
    }
    return m;
}

template <int STEP, bool SLIM>
__device__ __forceinline__ bool hit_needs_walk(const StrandView &T, const StrandView &Q, const uint2 h, int xdrop,
                                               int hspthresh, int transitions) {
    Frame F;
    if (SLIM) load_frame_slim(T, Q, h, F);  // neither strand holds an N (the host checked)
    else load_frame(T, Q, h, F);
    const uint32_t bt = h.x & 31u;
    const FilterMasks l0 = filter_left<0>(F, bt, transitions), l1 = filter_left<1>(F, bt, transitions);
    const uint32_t rs = bt + SEED_LEN;
    Bound L{0, 0, 0, 0, false}, R{0, 0, 0, 0, false};
    bound_window<STEP, true>(L, l0.dl, l0.dh, l0.cg, xdrop);
    bound_window<STEP, true>(L, l1.dl, l1.dh, l1.cg, xdrop);
    // sixteen more steps on the left (the frame holds them): 99 % instead of 92 % of the left stops are proven.
    // Their boundaries only matter when the stop was not proven within 64 steps.
    const bool stop64 = L.stop;
    bound_window<16, true, 1>(L, l1.dl2, l1.dh2, l1.cg2, xdrop);
    const uint32_t veto2 = stop64 ? 0u : l1.H2;
    bound_window<STEP, false>(R, ext32(F.dl[2], F.dl[3], F.dl[4], rs), ext32(F.dh[2], F.dh[3], F.dh[4], rs),
                              ext32(F.cg[2], F.cg[3], F.cg[4], rs), xdrop);
    bound_window<STEP, false>(R, ext32(F.dl[3], F.dl[4], F.dl[5], rs), ext32(F.dh[3], F.dh[4], F.dh[5], rs),
                              ext32(F.cg[3], F.cg[4], F.cg[5], rs), xdrop);
    // an N anywhere in the frame (a superset of the 128 steps looked at) or a possible earlier seed hit: exact walk
    // (no early exit: the test is folded into the result so that nothing has to wait for all thirteen loads)
    const uint32_t veto = F.nn[0] | F.nn[1] | F.nn[2] | F.nn[3] | F.nn[4] | F.nn[5] | l0.H | l1.H | veto2;
    return !(L.stop && R.stop && L.ub + R.ub < hspthresh && veto == 0);
}

// the exact walk of one hit from its frame: classifies it (to the generic kernel / follower / candidate)
template <int VARIANT>
__device__ __forceinline__ void walk_hit(const uint32_t *__restrict__ tab, const StrandView &T, const StrandView &Q,
                                         const uint2 h, int xdrop, int hspthresh, int transitions, bool &q_med,
                                         bool &q_fol, bool &q_cd, uint64_t &r_fk, uint32_t &r_fp, Cand &r_cd) {
    const int32_t et = (int32_t)h.x + SEED_LEN, eq = (int32_t)h.y + SEED_LEN;
    const int32_t d = (int32_t)h.x - (int32_t)h.y;
    const uint32_t bt = h.x & 31u;
    Frame F;
    if (VARIANT == 8) {  // loads only, with the address pattern of a two-plane (8 bytes per 32 bases) copy: timing experiment
        const uint32_t bq = h.y & 31u;
        const int32_t wt = (int32_t)(h.x >> 5) - 2, wq = (int32_t)(h.y >> 5) - 2 - (bq < bt ? 1 : 0);
        const uint2 *t2 = reinterpret_cast<const uint2 *>(T.pw), *q2 = reinterpret_cast<const uint2 *>(Q.pw);
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) { const uint2 v = t2[wt + k]; acc ^= v.x ^ v.y; }
#pragma unroll
        for (int k = 0; k < 7; k++) { const uint2 v = q2[wq + k]; acc ^= v.x ^ v.y; }
        if (acc == 0x12345678u) q_med = true;
        return;
    }
    if (VARIANT == 3) {  // compute only (timing experiment, wrong results)
#pragma unroll
        for (int k = 0; k < 6; k++) {
            F.dl[k] = h.x * (k + 1); F.dh[k] = h.y + k; F.cg[k] = h.y * (k + 3); F.nn[k] = 0; F.st[k] = ~0u; F.sq[k] = ~0u;
        }
    } else {
        load_frame(T, Q, h, F);
    }
    if (VARIANT == 2) {  // loads only (timing experiment, wrong results)
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) acc ^= F.dl[k] ^ F.dh[k] ^ F.cg[k] ^ F.nn[k] ^ F.st[k] ^ F.sq[k];
        if (acc == 0x12345678u) q_med = true;
        return;
    }
    // ---- left walk: up to two windows from the frame (the seed starts at frame bit 64 + bt)
    WalkState L{0, 0, 0, 0, false, false, 0};
    const uint32_t maxl = (uint32_t)min(et, eq);
    left_window<0>(tab, F, bt, transitions, L, maxl, xdrop);
    if (!L.done) left_window<1>(tab, F, bt, transitions, L, maxl, xdrop);
    if (!L.done) {  // still alive after 64 bases: generic kernel
        q_med = true;
    } else if (L.found) {
        q_fol = true;
        r_fk = ((uint64_t)(uint32_t)(d + (int32_t)Q.len) << 32) | (uint32_t)et;
        r_fp = (uint32_t)et - L.found_step;
    } else {
        // ---- right walk: two windows from the frame (frame bit of the seed end = 64 + bt + 19)
        WalkState R{0, 0, 0, 0, false, false, 0};
        const uint32_t maxr = min(T.len - (uint32_t)et, Q.len - (uint32_t)eq);
        const uint32_t rs = bt + SEED_LEN;
        walk_window_pred(tab, R, ext32(F.dl[2], F.dl[3], F.dl[4], rs), ext32(F.dh[2], F.dh[3], F.dh[4], rs),
                         ext32(F.cg[2], F.cg[3], F.cg[4], rs), ext32(F.nn[2], F.nn[3], F.nn[4], rs), 0u, maxr, xdrop);
        if (!R.done)
            walk_window_pred(tab, R, ext32(F.dl[3], F.dl[4], F.dl[5], rs), ext32(F.dh[3], F.dh[4], F.dh[5], rs),
                             ext32(F.cg[3], F.cg[4], F.cg[5], rs), ext32(F.nn[3], F.nn[4], F.nn[5], rs), 0u, maxr, xdrop);
        if (!R.done) {
            q_med = true;
        } else if (L.best + R.best >= hspthresh) {
            q_cd = true;
            r_cd = Cand{(uint32_t)et - L.bk, (uint32_t)eq - L.bk, L.bk + R.bk, L.best + R.best};
        }
    }
}

// VARIANT 9 / 5 = production: pre-filter with checkpoints every 16 steps, reading the two-plane copy (9: neither
// strand holds an N) or the full planes (5); 4 = every 8 steps; 1 = no pre-filter, every hit is walked; 2 = loads only,
// 3 = compute only, 8 = loads only from the two-plane copy (timing experiments, wrong results)
template <int VARIANT>
__global__ __launch_bounds__(FAST_THREADS) void k4_extend_hits(StrandView T, StrandView Q,
                                                              const uint2 *__restrict__ hits, uint64_t nhits_arg,
                                                              int xdrop, int hspthresh, int transitions,
                                                              const uint32_t *__restrict__ group_tab,
                                                              ExtCounters *__restrict__ ctr, Cand *__restrict__ cand,
                                                              uint64_t cand_cap, uint64_t *__restrict__ fkey,
                                                              uint32_t *__restrict__ fprev,
                                                              uint2 *__restrict__ medq, int skip_diag0,
                                                              const unsigned long long *__restrict__ nhits_dev) {
    constexpr bool FILTER = VARIANT == 4 || VARIANT == 5 || VARIANT == 9;
    // speculative launch: the count comes from the seed scan on the device, nhits_arg is the buffer capacity
    // (a count beyond it means the scan wrote nothing: no work)
    uint64_t nhits = nhits_arg;
    if (nhits_dev) { const uint64_t t = *nhits_dev; nhits = t > nhits_arg ? 0 : t; }
    __shared__ uint32_t tab[GROUP_TAB];
    // per-wave staging of the three output queues: one global atomic per >= 64 records instead of
    // one per wavefront iteration (same-address atomics serialise at ~15 ns each)
    __shared__ uint2 s_med[FAST_THREADS / 64][QCAP];
    __shared__ uint64_t s_fk[FAST_THREADS / 64][QCAP];
    __shared__ uint32_t s_fp[FAST_THREADS / 64][QCAP];
    __shared__ Cand s_cd[FAST_THREADS / 64][QCAP];
    __shared__ uint2 s_walk[FILTER ? FAST_THREADS / 64 : 1][QCAP];  // hits that passed the pre-filter, per wavefront
    for (int i = threadIdx.x; i < GROUP_TAB; i += FAST_THREADS) tab[i] = group_tab[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint32_t n_med = 0, n_fol = 0, n_cd = 0, n_walk = 0, n_walked = 0;  // wave-uniform fill levels

    // walk one hit per lane exactly and stage the records; a queue is flushed with one atomic when the new
    // records would not fit (QCAP = 64 = the most one batch can add); `final` flushes what is left
    auto walk_batch = [&](bool active, uint2 h, bool final) {
        bool q_med = false, q_fol = false, q_cd = false;
        uint64_t r_fk = 0;
        uint32_t r_fp = 0;
        Cand r_cd{0, 0, 0, 0};
        if (active) walk_hit<VARIANT>(tab, T, Q, h, xdrop, hspthresh, transitions, q_med, q_fol, q_cd, r_fk, r_fp, r_cd);
        uint64_t m = __ballot(q_med);
        if (m || (final && n_med)) {
            const uint32_t add = (uint32_t)__popcll(m);
            if (n_med + add > (uint32_t)QCAP || (final && !m)) {
                __builtin_amdgcn_wave_barrier();  // LDS accesses of one wavefront execute in order
                unsigned long long b = 0;
                if (lane == 0) b = atomicAdd(&ctr->nmed, (unsigned long long)n_med);
                b = __shfl(b, 0);
                if (lane < n_med) medq[b + lane] = s_med[wv][lane];
                n_med = 0;
                __builtin_amdgcn_wave_barrier();
            }
            if (q_med) s_med[wv][n_med + __popcll(m & lt_mask)] = h;
            n_med += add;
            if (final && n_med) {
                __builtin_amdgcn_wave_barrier();
                unsigned long long b = 0;
                if (lane == 0) b = atomicAdd(&ctr->nmed, (unsigned long long)n_med);
                b = __shfl(b, 0);
                if (lane < n_med) medq[b + lane] = s_med[wv][lane];
                n_med = 0;
            }
        }
        m = __ballot(q_fol);
        if (m || (final && n_fol)) {
            const uint32_t add = (uint32_t)__popcll(m);
            if (n_fol + add > (uint32_t)QCAP || (final && !m)) {
                __builtin_amdgcn_wave_barrier();
                unsigned long long b = 0;
                if (lane == 0) b = atomicAdd(&ctr->nfollow, (unsigned long long)n_fol);
                b = __shfl(b, 0);
                if (lane < n_fol) { fkey[b + lane] = s_fk[wv][lane]; fprev[b + lane] = s_fp[wv][lane]; }
                n_fol = 0;
                __builtin_amdgcn_wave_barrier();
            }
            if (q_fol) { const uint32_t i = n_fol + __popcll(m & lt_mask); s_fk[wv][i] = r_fk; s_fp[wv][i] = r_fp; }
            n_fol += add;
            if (final && n_fol) {
                __builtin_amdgcn_wave_barrier();
                unsigned long long b = 0;
                if (lane == 0) b = atomicAdd(&ctr->nfollow, (unsigned long long)n_fol);
                b = __shfl(b, 0);
                if (lane < n_fol) { fkey[b + lane] = s_fk[wv][lane]; fprev[b + lane] = s_fp[wv][lane]; }
                n_fol = 0;
            }
        }
        m = __ballot(q_cd);
        if (m || (final && n_cd)) {
            const uint32_t add = (uint32_t)__popcll(m);
            if (n_cd + add > (uint32_t)QCAP || (final && !m)) {
                __builtin_amdgcn_wave_barrier();
                unsigned long long b = 0;
                if (lane == 0) b = atomicAdd(&ctr->ncand, (unsigned long long)n_cd);
                b = __shfl(b, 0);
                if (lane < n_cd && b + lane < cand_cap) cand[b + lane] = s_cd[wv][lane];
                n_cd = 0;
                __builtin_amdgcn_wave_barrier();
            }
            if (q_cd) s_cd[wv][n_cd + __popcll(m & lt_mask)] = r_cd;
            n_cd += add;
            if (final && n_cd) {
                __builtin_amdgcn_wave_barrier();
                unsigned long long b = 0;
                if (lane == 0) b = atomicAdd(&ctr->ncand, (unsigned long long)n_cd);
                b = __shfl(b, 0);
                if (lane < n_cd && b + lane < cand_cap) cand[b + lane] = s_cd[wv][lane];
                n_cd = 0;
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    const uint64_t stride = (uint64_t)gridDim.x * FAST_THREADS;
    for (uint64_t g0 = (uint64_t)blockIdx.x * FAST_THREADS + wv * 64u; g0 < nhits; g0 += stride) {
        const uint64_t gid = g0 + lane;
        uint2 h = make_uint2(0, 0);
        if (gid < nhits) h = hits[gid];
        const bool valid = gid < nhits && !(skip_diag0 && h.x == h.y);  // the main diagonal of a self unit belongs to k4_diag0
        if (!FILTER) {
            walk_batch(valid, h, false);
        } else {
            bool need = false;
            if (valid)
                need = VARIANT == 9 ? hit_needs_walk<16, true>(T, Q, h, xdrop, hspthresh, transitions)
                                    : hit_needs_walk<VARIANT == 4 ? 8 : 16, false>(T, Q, h, xdrop, hspthresh, transitions);
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
    if (FILTER && n_walked && lane == 0) atomicAdd(&ctr->nwalked, (unsigned long long)n_walked);
    walk_batch(false, make_uint2(0, 0), true);  // flush the staged records
}

// ---- wave-cooperative walk: 64 bases per step ---------------------------------------------
struct WalkResult {
    int64_t best;     // best prefix score
    uint32_t bsteps;  // number of steps in the best prefix
    bool found;       // (left + detect) an earlier seed hit ends at a reached boundary
    uint32_t prev_end;
};

__device__ __forceinline__ int64_t wave_incl_sum(int64_t v, uint32_t lane) {
    for (int o = 1; o < 64; o <<= 1) {
        int64_t u = __shfl_up(v, o);
        if (lane >= (uint32_t)o) v += u;
    }
    return v;
}
__device__ __forceinline__ int64_t wave_incl_max(int64_t v, uint32_t lane) {
    for (int o = 1; o < 64; o <<= 1) {
        int64_t u = __shfl_up(v, o);
        if (lane >= (uint32_t)o) v = max(v, u);
    }
    return v;
}
__device__ __forceinline__ int64_t wave_max(int64_t v) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}

// dir = -1: steps visit et-1, et-2, ...;  dir = +1: et, et+1, ...   All lanes return the same value.
__device__ WalkResult wave_walk(const StrandView &T, const StrandView &Q, int32_t et, int32_t d, int dir,
                                uint32_t maxsteps, int xdrop, bool detect, int transitions) {
    const uint32_t lane = threadIdx.x & 63u;
    WalkResult r{0, 0, false, 0};
    int64_t run = 0;
    for (uint32_t k0 = 0; k0 < maxsteps; k0 += 64) {
        uint32_t i = k0 + lane;
        bool active = i < maxsteps;
        int32_t pt = dir < 0 ? et - 1 - (int32_t)i : et + (int32_t)i;
        int32_t pq = pt - d;
        int64_t s = 0;
        if (active) {
            const Base1 ta = base_at(T, pt), qa = base_at(Q, pq);
            s = sub_score(ta.lo ^ qa.lo, ta.hi ^ qa.hi, ta.lo ^ ta.hi, ta.nm | qa.nm);
        }
        int64_t P = run + wave_incl_sum(s, lane);
        int64_t M = max(r.best, wave_incl_max(P, lane));
        bool brk = active && (P < M - xdrop);
        uint64_t bmask = __ballot(brk);
        uint32_t f = bmask ? (uint32_t)__builtin_ctzll(bmask) : 64u;  // first breaking lane
        uint32_t nact = min(64u, maxsteps - k0);
        if (detect) {
            bool hit = active && lane < f && seed_hit_at(T, Q, pt - SEED_LEN, d, transitions);
            uint64_t hmask = __ballot(hit);
            if (hmask) {
                uint32_t g = (uint32_t)__builtin_ctzll(hmask);
                r.found = true;
                r.prev_end = (uint32_t)(et - 1 - (int32_t)(k0 + g));
                return r;
            }
        }
        uint32_t lim = min(f, nact - 1);  // last executed lane (the breaking step itself is executed)
        int64_t cm = wave_max(lane <= lim ? P : INT64_MIN);
        if (cm > r.best) {
            uint64_t em = __ballot(lane <= lim && P == cm);
            r.best = cm;
            r.bsteps = k0 + (uint32_t)__builtin_ctzll(em) + 1;
        }
        run = __shfl(P, (int)lim);
        if (f < 64) break;
    }
    return r;
}

// Word-granular walk without seed detection: 64 lanes x 8 words x 32 bases = 16384 bases per step.
// Each lane reduces its 128 bases to (sum S, best prefix M and its position, lowest prefix mn,
// deepest drop a below the running in-lane maximum); a wave prefix sum / prefix max turns these
// into the exact running score and best at every lane boundary, and the walk can only stop inside
// the first lane with  a < -xdrop  or  run_in + mn < best_in - xdrop, which then replays its
// bases one by one.  Words that are all matches (the common case on long diagonals) need no
// per-base loop at all.
constexpr int WALK_WORDS = 8;

__device__ WalkResult wave_walk_fast(const StrandView &T, const StrandView &Q, int32_t et, int32_t d, int dir,
                                     uint32_t maxsteps, int xdrop) {
    const uint32_t lane = threadIdx.x & 63u;
    WalkResult r{0, 0, false, 0};
    int64_t run = 0;
    // the first step covers 2048 bases with one word per lane (most walks end there), later steps
    // take WALK_WORDS words per lane
    uint32_t nw = 1;
    for (uint32_t base = 0; base < maxsteps; base += 64u * 32u * nw, nw = WALK_WORDS) {
        const uint32_t off = base + lane * 32u * nw;
        uint32_t dl[WALK_WORDS], dh[WALK_WORDS], cg[WALK_WORDS], nn[WALK_WORDS], nst[WALK_WORDS];
#pragma unroll
        for (int j = 0; j < WALK_WORDS; j++) {
            const uint32_t o = off + 32u * j;
            nst[j] = ((uint32_t)j < nw && o < maxsteps) ? min(32u, maxsteps - o) : 0u;
            dl[j] = dh[j] = cg[j] = nn[j] = 0;
            if (nst[j]) {  // bit b of every mask <-> step o + b
                int32_t pt = dir > 0 ? et + (int32_t)o : et - (int32_t)o - 32, pq = pt - d;
                const Win32 tw = win32(T, pt), qw = win32(Q, pq);
                dl[j] = tw.lo ^ qw.lo; dh[j] = tw.hi ^ qw.hi; cg[j] = tw.lo ^ tw.hi; nn[j] = tw.nm | qw.nm;
                if (dir < 0) { dl[j] = __brev(dl[j]); dh[j] = __brev(dh[j]); cg[j] = __brev(cg[j]); nn[j] = __brev(nn[j]); }
            }
        }
        // lane summary over its (up to) 128 steps
        int32_t S = 0, M = INT32_MIN / 2, mn = INT32_MAX / 2, a = 0;
        uint32_t posM = 0, ntot = 0;
#pragma unroll
        for (int j = 0; j < WALK_WORDS; j++) {
            if (!nst[j]) continue;
            const uint32_t valid = nst[j] == 32 ? 0xFFFFFFFFu : ((1u << nst[j]) - 1u);
            int32_t wS, wM, wmn, wa = 0;
            uint32_t wpos;
            if (((dl[j] | dh[j] | nn[j]) & valid) == 0) {  // all matches: prefixes strictly increase
                wS = 91 * (int32_t)nst[j] + 9 * __popc(cg[j] & valid);
                wM = wS; wpos = nst[j]; wmn = (cg[j] & 1u) ? 100 : 91;
            } else {
                int32_t p = 0;
                wM = INT32_MIN / 2; wmn = INT32_MAX / 2; wpos = 0;
                for (uint32_t bb = 0; bb < nst[j]; bb++) {
                    p += sub_score((dl[j] >> bb) & 1u, (dh[j] >> bb) & 1u, (cg[j] >> bb) & 1u, (nn[j] >> bb) & 1u);
                    if (p > wM) { wM = p; wpos = bb + 1; }
                    wmn = min(wmn, p);
                    wa = min(wa, p - wM);
                }
                wS = p;
            }
            if (ntot == 0) { S = wS; M = wM; mn = wmn; a = wa; posM = wpos; }
            else {  // compose (lane so far) then (word j)
                a = min(a, min(S - M + wmn, wa));
                mn = min(mn, S + wmn);
                if (S + wM > M) { M = S + wM; posM = ntot + wpos; }
                S += wS;
            }
            ntot += nst[j];
        }
        // running score entering my lane, best entering my lane
        int64_t incS = wave_incl_sum((int64_t)S, lane);
        int64_t run_in = run + incS - S;
        int64_t cand = ntot ? run_in + M : INT64_MIN;  // best reached inside my lane
        int64_t incB = wave_incl_max(cand, lane);
        int64_t prevB = __shfl_up(incB, 1);
        int64_t best_in = lane ? max(r.best, prevB) : r.best;
        bool mb = ntot && (a < -xdrop || run_in + mn < best_in - xdrop);
        uint64_t bmask = __ballot(mb);
        uint32_t first = bmask ? (uint32_t)__builtin_ctzll(bmask) : 64u;
        // accept every lane before `first`: best = earliest lane reaching the maximum
        int64_t cm = wave_max((lane < first && ntot) ? cand : INT64_MIN);
        if (cm > r.best) {
            uint64_t em = __ballot(lane < first && ntot && cand == cm);
            uint32_t wl = (uint32_t)__builtin_ctzll(em);
            r.best = cm;
            r.bsteps = base + wl * 32u * nw + (uint32_t)__shfl((int)posM, (int)wl);
        }
        if (first < 64u) {
            // the walk ends inside lane `first`: replay its bases exactly
            int64_t lb = r.best;
            uint32_t lbs = r.bsteps;
            if (lane == first) {
                int64_t p = run_in;
                bool stop = false;
#pragma unroll
                for (int j = 0; j < WALK_WORDS; j++) {
                    for (uint32_t bb = 0; bb < nst[j] && !stop; bb++) {
                        p += sub_score((dl[j] >> bb) & 1u, (dh[j] >> bb) & 1u, (cg[j] >> bb) & 1u, (nn[j] >> bb) & 1u);
                        if (p > lb) { lb = p; lbs = off + 32u * j + bb + 1; }
                        if (p < lb - xdrop) stop = true;
                    }
                }
            }
            r.best = __shfl(lb, (int)first);
            r.bsteps = (uint32_t)__shfl((int)lbs, (int)first);
            return r;
        }
        run += __shfl(incS, 63);
    }
    return r;
}

// full extension of one hit by one wavefront; emits candidate or follower record (lane 0)
__device__ void wave_extend_emit(const StrandView &T, const StrandView &Q, uint2 h, int xdrop, int hspthresh,
                                 int transitions, bool detect, ExtCounters *ctr, Cand *cand, uint64_t cand_cap,
                                 uint64_t *fkey, uint32_t *fprev, uint32_t *rext_out) {
    const int32_t et = (int32_t)h.x + SEED_LEN, eq = (int32_t)h.y + SEED_LEN, d = (int32_t)h.x - (int32_t)h.y;
    WalkResult L = detect ? wave_walk(T, Q, et, d, -1, (uint32_t)min(et, eq), xdrop, true, transitions)
                          : wave_walk_fast(T, Q, et, d, -1, (uint32_t)min(et, eq), xdrop);
    if (L.found) {
        if ((threadIdx.x & 63) == 0) {
            unsigned long long i = atomicAdd(&ctr->nfollow, 1ull);
            fkey[i] = ((uint64_t)(uint32_t)(d + (int32_t)Q.len) << 32) | (uint32_t)et;
            fprev[i] = L.prev_end;
        }
        return;
    }
    WalkResult R = wave_walk_fast(T, Q, et, d, +1, min(T.len - (uint32_t)et, Q.len - (uint32_t)eq), xdrop);
    if (rext_out) *rext_out = R.bsteps;
    int64_t score = L.best + R.best;
    if (score >= hspthresh && (threadIdx.x & 63) == 0) {
        unsigned long long i = atomicAdd(&ctr->ncand, 1ull);
        if (i < cand_cap)
            cand[i] = Cand{(uint32_t)et - L.bsteps, (uint32_t)eq - L.bsteps, L.bsteps + R.bsteps,
                           score >= (int64_t)RAW_SATURATED ? RAW_SATURATED : (int32_t)score};  // k4_entropy recounts a saturated score
    }
}

// ---- K4b: long hits, one wavefront each ------------------------------------------------
__global__ __launch_bounds__(EXT_THREADS) void k4_extend_long(StrandView T, StrandView Q,
                                                              const uint2 *__restrict__ longq, int xdrop,
                                                              int hspthresh, int transitions,
                                                              ExtCounters *__restrict__ ctr, Cand *__restrict__ cand,
                                                              uint64_t cand_cap, uint64_t *__restrict__ fkey,
                                                              uint32_t *__restrict__ fprev) {
    const uint64_t nlong = ctr->nlong, nwaves = ((uint64_t)gridDim.x * EXT_THREADS) >> 6;
    for (uint64_t wid = ((uint64_t)blockIdx.x * EXT_THREADS + threadIdx.x) >> 6; wid < nlong; wid += nwaves)
        wave_extend_emit(T, Q, longq[wid], xdrop, hspthresh, transitions, true, ctr, cand, cand_cap, fkey, fprev, nullptr);
}

// ---- K4c: follower segments ----------------------------------------------------------------
// flag[i] = 1 iff sorted follower i starts a segment (its predecessor is a head, not follower i-1)
__global__ void k4_segment_flags(const uint64_t *__restrict__ key, const uint32_t *__restrict__ prev, uint64_t n,
                                 uint8_t *__restrict__ flag) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool start = true;
    if (i > 0) {
        uint64_t a = key[i - 1], b = key[i];
        start = (a >> 32) != (b >> 32) || (uint32_t)a != prev[i];
    }
    flag[i] = start ? 1 : 0;
}

// per-lane walk without seed detection, windows loaded on demand; false = still alive after
// LONG_WINDOWS windows (the caller hands the work to a wavefront)
__device__ __forceinline__ bool lane_walk(const uint32_t *__restrict__ tab, const StrandView &T, const StrandView &Q,
                                          int32_t et, int32_t d, int dir, uint32_t limit, int xdrop, WalkState &w) {
    for (int win = 0; !w.done; win++) {
        if (win == LONG_WINDOWS) return false;
        const int32_t P = dir < 0 ? et - 32 * (win + 1) : et + 32 * win, Pq = P - d;
        const Win32 tw = win32(T, P), qw = win32(Q, Pq);
        uint32_t mdl = tw.lo ^ qw.lo, mdh = tw.hi ^ qw.hi, mcg = tw.lo ^ tw.hi, mnn = tw.nm | qw.nm;
        if (dir < 0) { mdl = __brev(mdl); mdh = __brev(mdh); mcg = __brev(mcg); mnn = __brev(mnn); }
        walk_window(tab, w, mdl, mdh, mcg, mnn, 0u, limit, xdrop);
    }
    return true;
}

// One LANE per segment: almost every segment is a pair of neighbouring random hits (one head, one
// follower, walks of a few dozen bases).  Segments with more than SMALL_SEG members or with a walk
// longer than LONG_WINDOWS windows are queued for k4_resolve_segments (one wavefront each).
constexpr uint32_t SMALL_SEG = 8;
__global__ __launch_bounds__(EXT_THREADS) void k4_resolve_small(StrandView T, StrandView Q,
                                                                const uint64_t *__restrict__ key,
                                                                const uint32_t *__restrict__ prev, uint64_t nfollow,
                                                                const uint64_t *__restrict__ seg_start,
                                                                const uint64_t *__restrict__ nseg_dev,
                                                                int xdrop, int hspthresh,
                                                                const uint32_t *__restrict__ group_tab,
                                                                ExtCounters *__restrict__ ctr, Cand *__restrict__ cand,
                                                                uint64_t cand_cap, uint64_t *__restrict__ bigseg) {
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
    if (!big) {
        const int32_t d = (int32_t)(uint32_t)(key[beg] >> 32) - (int32_t)Q.len;
        const int32_t het = (int32_t)prev[beg];
        WalkState H{0, 0, 0, 0, false, false, 0};
        big = !lane_walk(tab, T, Q, het, d, +1, min(T.len - (uint32_t)het, Q.len - (uint32_t)(het - d)), xdrop, H);
        uint32_t reach = (uint32_t)het + H.bk;
        for (uint64_t i = beg; i < end && !big; i++) {
            const uint32_t et = (uint32_t)key[i];
            if (et <= reach) continue;  // inside the region the previous extension reached
            const int32_t eq = (int32_t)et - d;
            WalkState L{0, 0, 0, 0, false, false, 0}, R{0, 0, 0, 0, false, false, 0};
            if (!lane_walk(tab, T, Q, (int32_t)et, d, -1, (uint32_t)min((int32_t)et, eq), xdrop, L) ||
                !lane_walk(tab, T, Q, (int32_t)et, d, +1, min(T.len - et, Q.len - (uint32_t)eq), xdrop, R)) {
                big = true;
                break;
            }
            if (L.best + R.best >= hspthresh) out[nout++] = Cand{et - L.bk, (uint32_t)eq - L.bk, L.bk + R.bk, L.best + R.best};
            reach = et + R.bk;
        }
    }
    if (big) {  // nothing has been emitted for this segment yet: the wavefront kernel redoes it
        bigseg[atomicAdd(&ctr->nbig, 1ull)] = sid;
        return;
    }
    for (uint32_t k = 0; k < nout; k++) {
        unsigned long long i = atomicAdd(&ctr->ncand, 1ull);
        if (i < cand_cap) cand[i] = out[k];
    }
}

// one wavefront per segment: replay "skip while seed end <= reach, else extend" (lastz diagEnd rule)
__global__ __launch_bounds__(EXT_THREADS) void k4_resolve_segments(StrandView T, StrandView Q,
                                                                   const uint64_t *__restrict__ key,
                                                                   const uint32_t *__restrict__ prev, uint64_t nfollow,
                                                                   const uint64_t *__restrict__ seg_start,
                                                                   const uint64_t *__restrict__ nseg_dev,
                                                                   const uint64_t *__restrict__ list, int xdrop,
                                                                   int hspthresh,
                                                                   int transitions, ExtCounters *__restrict__ ctr,
                                                                   Cand *__restrict__ cand, uint64_t cand_cap) {
    const uint64_t nseg = *nseg_dev, nlist = ctr->nbig, nwaves = ((uint64_t)gridDim.x * EXT_THREADS) >> 6;
    for (uint64_t wid = ((uint64_t)blockIdx.x * EXT_THREADS + threadIdx.x) >> 6; wid < nlist; wid += nwaves) {
    const uint64_t sid = list[wid];
    uint64_t beg = seg_start[sid], end = sid + 1 < nseg ? seg_start[sid + 1] : nfollow;
    uint64_t k0 = key[beg];
    const int32_t d = (int32_t)(uint32_t)(k0 >> 32) - (int32_t)Q.len;
    // head: seed end = prev[beg]; only its right extent matters (it was emitted by K4a/K4b)
    int32_t het = (int32_t)prev[beg];
    WalkResult R = wave_walk_fast(T, Q, het, d, +1, min(T.len - (uint32_t)het, Q.len - (uint32_t)(het - d)), xdrop);
    uint32_t reach = (uint32_t)het + R.bsteps;
    uint64_t i = beg;
    while (i < end) {
        // first member at or after i whose seed end lies beyond reach (members are sorted by seed end)
        uint64_t lo = i, hi = end;
        while (lo < hi) {
            uint64_t mid = (lo + hi) >> 1;
            if ((uint32_t)key[mid] > reach) hi = mid; else lo = mid + 1;
        }
        uint64_t nxt = lo;
        if (nxt >= end) break;
        uint32_t et = (uint32_t)key[nxt];
        uint2 h = make_uint2(et - SEED_LEN, (uint32_t)((int32_t)et - d) - SEED_LEN);
        uint32_t rext = 0;
        wave_extend_emit(T, Q, h, xdrop, hspthresh, transitions, false, ctr, cand, cand_cap, nullptr, nullptr, &rext);
        reach = et + rext;
        i = nxt + 1;
    }
    }
}

// ---- K4e: the main diagonal of a strand aligned to itself -------------------------------------
// When target and query are the same strand every valid seed position is a hit on diagonal 0
// (millions of followers of one head).  K4a leaves those hits alone and this kernel replays the
// sequential rule directly on the seed-validity planes with one wavefront: extend the first seed,
// skip every seed whose end lies inside the reach, extend the next one, ...
__global__ __launch_bounds__(64) void k4_diag0(StrandView T, StrandView Q, int xdrop, int hspthresh, int transitions,
                                               ExtCounters *__restrict__ ctr, Cand *__restrict__ cand, uint64_t cand_cap) {
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
        wave_extend_emit(T, Q, make_uint2(found, found), xdrop, hspthresh, transitions, false, ctr, cand, cand_cap, nullptr,
                         nullptr, &rext);
        const uint32_t reach = found + SEED_LEN + rext;  // a later seed is extended iff its end lies beyond
        p = max(found + 1u, reach - (uint32_t)(SEED_LEN - 1));
    }
}

// ---- K4d: entropy adjustment + threshold, one wavefront per candidate --------------------
__global__ __launch_bounds__(EXT_THREADS) void k4_entropy(StrandView T, StrandView Q, const Cand *__restrict__ cand,
                                                          uint64_t cand_cap, int hspthresh, int entropy,
                                                          ExtCounters *__restrict__ ctr, mimeo_hsp *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t ncand = min((uint64_t)ctr->ncand, cand_cap), nwaves = ((uint64_t)gridDim.x * EXT_THREADS) >> 6;
    for (uint64_t cid = ((uint64_t)blockIdx.x * EXT_THREADS + threadIdx.x) >> 6; cid < ncand; cid += nwaves) {
    Cand c = cand[cid];
    int64_t raw = c.raw;
    const int32_t d = (int32_t)c.tstart - (int32_t)c.qstart;
    if (c.raw == RAW_SATURATED) {  // recount the column scores of the segment in 64 bits (wave-uniform, rare)
        int64_t sum = 0;
        for (uint32_t o = lane * 32u; o < c.len; o += 64u * 32u) {
            const int32_t pt = (int32_t)(c.tstart + o);
            const Win32 tw = win32(T, pt), qw = win32(Q, pt - d);
            const uint32_t rem = c.len - o, valid = rem < 32 ? (1u << rem) - 1u : 0xFFFFFFFFu;
            const uint32_t nn = (tw.nm | qw.nm) & valid, dl = (tw.lo ^ qw.lo) & ~nn & valid, dh = (tw.hi ^ qw.hi) & ~nn & valid;
            const uint32_t cg = tw.lo ^ tw.hi, ok = valid & ~nn;
            sum += 91ll * __popc(ok) + 9ll * __popc(ok & ~(dl | dh) & cg) - 122ll * __popc(~dl & dh) - 205ll * __popc(dl) -
                   9ll * __popc(dl & dh) - 2ll * __popc(dl & dh & cg) - 100ll * __popc(nn);
        }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        raw = sum;
    }
    int64_t adj = raw;
    if (entropy) {
        uint32_t cnt[4] = {0, 0, 0, 0};
        // four independent windows per iteration: the loads of a long HSP overlap instead of queueing
        for (uint32_t w0 = lane * 32u; w0 < c.len; w0 += 4u * 64u * 32u) {
            Win32 tw[4], qw[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t o = w0 + (uint32_t)j * 64u * 32u;
                if (o < c.len) {
                    const int32_t pt = (int32_t)(c.tstart + o);
                    tw[j] = win32(T, pt);
                    qw[j] = win32(Q, pt - d);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t o = w0 + (uint32_t)j * 64u * 32u;
                if (o >= c.len) break;
                const uint32_t tlo = tw[j].lo, thi = tw[j].hi;
                uint32_t m = ~((tlo ^ qw[j].lo) | (thi ^ qw[j].hi)) & ~(tw[j].nm | qw[j].nm);
                const uint32_t rem = c.len - o;
                if (rem < 32) m &= (1u << rem) - 1u;
                cnt[0] += __popc(m & ~tlo & ~thi);
                cnt[1] += __popc(m & tlo & ~thi);
                cnt[2] += __popc(m & ~tlo & thi);
                cnt[3] += __popc(m & tlo & thi);
            }
        }
        for (int b = 0; b < 4; b++)
            for (int o = 32; o > 0; o >>= 1) cnt[b] += __shfl_xor(cnt[b], o);
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
    if (adj >= hspthresh && lane == 0) {
        unsigned long long i = atomicAdd(&ctr->nhsp, 1ull);
        mimeo_hsp h;
        h.tstart = c.tstart; h.qstart = c.qstart; h.length = c.len; h.flags = 0; h.score = adj; h.raw_score = raw;
        out[i] = h;
    }
    }
}

// ---- host orchestration ---------------------------------------------------------------------
static uint32_t *g_group_tab = nullptr;  // device copy of the 4-base group table (read-only, shared by all lanes)
static std::once_flag g_group_once;

void ExtWork::release() {
    if (ctr) (void)hipFree(ctr);
    ctr = nullptr;
    for (DeviceBuf *b : {&cand, &fkey, &fkey2, &fprev, &fprev2, &longq, &medq, &flags, &segs, &tmp, &nsel, &bigseg}) b->release();
}

int ungapped_hsps_device(ExtWork &W, const StrandView &T, const StrandView &Q, const uint2 *hits, uint64_t nhits,
                         const mimeo_params *p, DeviceBuf &out_hsps, uint64_t *nhsp, float *ms,
                         const std::function<void()> *after_fast, const unsigned long long *d_nhits, uint64_t *nhits_out,
                         ExtChunk *chunk) {
    hipStream_t st = stream();
    *nhsp = 0;
    const bool first = !chunk || chunk->first, last = !chunk || chunk->last;
    if (T.len >= 0x7FFFFF00u || Q.len >= 0x7FFFFF00u) { set_error("scaffold longer than 2^31 bases"); return MIMEO_ERR_LIMIT; }
    if (!W.ctr) HIP_TRY(hipMalloc((void **)&W.ctr, sizeof(ExtCounters)));
    int tab_rc = 0;
    std::call_once(g_group_once, [&] {
        std::vector<uint32_t> tab(GROUP_TAB);
        build_group_table(tab.data());
        if (hipMalloc((void **)&g_group_tab, GROUP_TAB * 4) != hipSuccess ||
            hipMemcpy(g_group_tab, tab.data(), GROUP_TAB * 4, hipMemcpyHostToDevice) != hipSuccess) {
            g_group_tab = nullptr;
            tab_rc = MIMEO_ERR_HIP;
        }
    });
    if (tab_rc || !g_group_tab) { set_error("group table upload failed"); return MIMEO_ERR_HIP; }
    if (first) HIP_TRY(hipMemsetAsync(W.ctr, 0, sizeof(ExtCounters), st));
    if (!nhits && !d_nhits && !chunk) { return out_hsps.reserve(sizeof(mimeo_hsp)); }
    unsigned long long total_h = nhits;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, st));
    // Candidates are rare (~1e-4 of the hits on random sequence): a modest buffer, and a rerun with the
    // exact size in the rare overflow case.  Kernels that consume a queue read its length from device
    // memory and run on fixed grids, so one host synchronisation (follower count, needed to size the
    // sort) and a final one are all a unit costs.
    uint64_t cand_cap = chunk ? chunk->cand_cap : nhits / 32 + 65536;
    int rc;
    ExtCounters c;
    for (int attempt = 0;; attempt++) {
        if (first) {
            if ((rc = W.cand.reserve(cand_cap * sizeof(Cand)))) return rc;
            if ((rc = out_hsps.reserve(cand_cap * sizeof(mimeo_hsp)))) return rc;
        }
        if (chunk && !first) {
            // the followers of the earlier chunks of this unit stay in front: grow with their content kept
            const uint64_t keep = chunk->nfollow_before, want = keep + nhits;
            for (auto bw : {std::make_pair(&W.fkey, (size_t)8), std::make_pair(&W.fprev, (size_t)4)}) {
                if (bw.first->cap >= want * bw.second) continue;
                DeviceBuf bigger;
                if ((rc = bigger.reserve(want * bw.second))) return rc;
                if (keep) HIP_TRY(hipMemcpyAsync(bigger.p, bw.first->p, keep * bw.second, hipMemcpyDeviceToDevice, st));
                HIP_TRY(hipStreamSynchronize(st));
                bw.first->release();
                *bw.first = bigger;
            }
        } else {
            if ((rc = W.fkey.reserve(nhits * 8))) return rc;
            if ((rc = W.fprev.reserve(nhits * 4))) return rc;
        }
        if ((rc = W.longq.reserve(nhits * 8))) return rc;
        if ((rc = W.medq.reserve(nhits * 8))) return rc;
        if ((rc = W.nsel.reserve(16))) return rc;
        if (first) HIP_TRY(hipMemsetAsync(W.ctr, 0, sizeof(ExtCounters), st));
        else {  // the per-chunk queues restart; candidates and followers go on
            HIP_TRY(hipMemsetAsync(&W.ctr->nlong, 0, sizeof(unsigned long long), st));
            HIP_TRY(hipMemsetAsync(&W.ctr->nmed, 0, sizeof(unsigned long long), st));
        }
        HIP_TRY(hipMemsetAsync(W.nsel.p, 0, 16, st));
        uint64_t nb = (nhits + FAST_THREADS - 1) / FAST_THREADS;
        // grid-stride over one resident set of workgroups (4 per CU): each loads the 16 KiB group table once.  Four
        // times as many workgroups cost 12 % of the kernel (0.87 -> 0.76 ms per C2 unit with 1024).
        static const uint64_t nb_cap = getenv("MIMEO_K4_BLOCKS") ? (uint64_t)atol(getenv("MIMEO_K4_BLOCKS")) : 1024;
        if (nb > nb_cap) nb = nb_cap;
        // target and query are the same strand of the same scaffold: diagonal 0 is handled by k4_diag0
        const int same_strand = (T.pw == Q.pw && T.len == Q.len && !getenv("MIMEO_NO_DIAG0")) ? 1 : 0;
        static int variant = getenv("MIMEO_K4_VARIANT") ? atoi(getenv("MIMEO_K4_VARIANT")) : 5;
        static const bool force_variant = getenv("MIMEO_K4_VARIANT") != nullptr;  // production: 9 when neither strand holds an N, else 5
#define K4_LAUNCH(V) hipLaunchKernelGGL(k4_extend_hits<V>, dim3((uint32_t)nb), dim3(FAST_THREADS), 0, st, T, Q, hits, nhits, p->xdrop, \
                           p->hspthresh, p->transitions, (const uint32_t *)g_group_tab, W.ctr, (Cand *)W.cand.p, cand_cap, \
                           (uint64_t *)W.fkey.p, (uint32_t *)W.fprev.p, (uint2 *)W.medq.p, same_strand, d_nhits)
        if (variant == 0 && d_nhits) return MIMEO_RETRY_EXACT;  // the development variant has no capacity guard
        if (nb == 0) {
            // an empty chunk of a chunked unit: nothing to launch
        } else if (variant == 0)
            hipLaunchKernelGGL(k4_extend_generic, dim3((uint32_t)(nb * 2)), dim3(EXT_THREADS), 0, st, T, Q, hits, nhits,
                               (const unsigned long long *)nullptr, p->xdrop, p->hspthresh, p->transitions,
                               (const uint32_t *)g_group_tab, W.ctr, (Cand *)W.cand.p, cand_cap, (uint64_t *)W.fkey.p,
                               (uint32_t *)W.fprev.p, (uint2 *)W.longq.p, same_strand);
        else if (variant == 2) K4_LAUNCH(2);
        else if (variant == 3) K4_LAUNCH(3);
        else if (variant == 4) K4_LAUNCH(4);
        else if (variant == 9 || (variant == 5 && !force_variant && !T.has_n && !Q.has_n)) K4_LAUNCH(9);
        else if (variant == 5) K4_LAUNCH(5);
        else if (variant == 8) K4_LAUNCH(8);
        else K4_LAUNCH(1);
        if (after_fast && attempt == 0) (*after_fast)();
        // walks still alive after the frame -> generic kernel; beyond LONG_WINDOWS -> wavefront kernel
        hipLaunchKernelGGL(k4_extend_generic, dim3(512), dim3(EXT_THREADS), 0, st, T, Q, (const uint2 *)W.medq.p,
                           (uint64_t)0, (const unsigned long long *)&W.ctr->nmed, p->xdrop, p->hspthresh, p->transitions,
                           (const uint32_t *)g_group_tab, W.ctr, (Cand *)W.cand.p, cand_cap, (uint64_t *)W.fkey.p,
                           (uint32_t *)W.fprev.p, (uint2 *)W.longq.p, 0);
        if (same_strand && first)
            hipLaunchKernelGGL(k4_diag0, dim3(1), dim3(64), 0, st, T, Q, p->xdrop, p->hspthresh, p->transitions, W.ctr,
                               (Cand *)W.cand.p, cand_cap);
        hipLaunchKernelGGL(k4_extend_long, dim3(64), dim3(EXT_THREADS), 0, st, T, Q, (const uint2 *)W.longq.p, p->xdrop,
                           p->hspthresh, p->transitions, W.ctr, (Cand *)W.cand.p, cand_cap, (uint64_t *)W.fkey.p,
                           (uint32_t *)W.fprev.p);
        HIP_TRY(hipMemcpyAsync(&c, W.ctr, sizeof c, hipMemcpyDeviceToHost, st));
        if (d_nhits) HIP_TRY(hipMemcpyAsync(&total_h, d_nhits, sizeof total_h, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (nhits_out) *nhits_out = total_h;
        if (d_nhits && total_h > nhits) return MIMEO_RETRY_EXACT;  // nothing was written; the caller repeats the unit
        if (chunk) {
            chunk->nfollow_after = c.nfollow;
            if (c.ncand > cand_cap) { set_error("candidate buffer overflow in a chunked unit"); return MIMEO_ERR_LIMIT; }
            if (!last) {  // more chunks of this unit follow: followers and candidates wait for the last one
                (void)hipEventDestroy(e0);
                (void)hipEventDestroy(e1);
                return 0;
            }
        }
        if (c.nfollow) {
            uint64_t nf = c.nfollow;
            if ((rc = W.fkey2.reserve(nf * 8))) return rc;
            if ((rc = W.fprev2.reserve(nf * 4))) return rc;
            if ((rc = W.flags.reserve(nf))) return rc;
            if ((rc = W.segs.reserve(nf * 8))) return rc;
            if ((rc = W.bigseg.reserve(nf * 8))) return rc;
            size_t t1 = 0, t2 = 0;
            HIP_TRY(rocprim::radix_sort_pairs(nullptr, t1, (uint64_t *)W.fkey.p, (uint64_t *)W.fkey2.p,
                                              (uint32_t *)W.fprev.p, (uint32_t *)W.fprev2.p, (size_t)nf, 0, 64, st));
            rocprim::counting_iterator<uint64_t> iota(0);
            HIP_TRY(rocprim::select(nullptr, t2, iota, (uint8_t *)W.flags.p, (uint64_t *)W.segs.p,
                                    (uint64_t *)W.nsel.p, (size_t)nf, st));
            if ((rc = W.tmp.reserve(std::max(t1, t2) + 16))) return rc;
            HIP_TRY(rocprim::radix_sort_pairs(W.tmp.p, t1, (uint64_t *)W.fkey.p, (uint64_t *)W.fkey2.p,
                                              (uint32_t *)W.fprev.p, (uint32_t *)W.fprev2.p, (size_t)nf, 0, 64, st));
            hipLaunchKernelGGL(k4_segment_flags, dim3((uint32_t)((nf + 255) / 256)), dim3(256), 0, st,
                               (const uint64_t *)W.fkey2.p, (const uint32_t *)W.fprev2.p, nf, (uint8_t *)W.flags.p);
            HIP_TRY(rocprim::select(W.tmp.p, t2, iota, (uint8_t *)W.flags.p, (uint64_t *)W.segs.p,
                                    (uint64_t *)W.nsel.p, (size_t)nf, st));
            // segments: at most nf of them; the kernels read the real number from W.nsel
            hipLaunchKernelGGL(k4_resolve_small, dim3((uint32_t)((nf + EXT_THREADS - 1) / EXT_THREADS)), dim3(EXT_THREADS), 0,
                               st, T, Q, (const uint64_t *)W.fkey2.p, (const uint32_t *)W.fprev2.p, nf,
                               (const uint64_t *)W.segs.p, (const uint64_t *)W.nsel.p, p->xdrop, p->hspthresh,
                               (const uint32_t *)g_group_tab, W.ctr, (Cand *)W.cand.p, cand_cap, (uint64_t *)W.bigseg.p);
            hipLaunchKernelGGL(k4_resolve_segments, dim3(256), dim3(EXT_THREADS), 0, st, T, Q, (const uint64_t *)W.fkey2.p,
                               (const uint32_t *)W.fprev2.p, nf, (const uint64_t *)W.segs.p, (const uint64_t *)W.nsel.p,
                               (const uint64_t *)W.bigseg.p, p->xdrop, p->hspthresh, p->transitions, W.ctr,
                               (Cand *)W.cand.p, cand_cap);
        }
        hipLaunchKernelGGL(k4_entropy, dim3(512), dim3(EXT_THREADS), 0, st, T, Q, (const Cand *)W.cand.p, cand_cap,
                           p->hspthresh, p->entropy, W.ctr, (mimeo_hsp *)out_hsps.p);
        HIP_TRY(hipMemcpyAsync(&c, W.ctr, sizeof c, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (getenv("MIMEO_K4_STATS"))
            fprintf(stderr, "[k4] hits %llu walked %llu (%.2f%%) generic %llu long %llu followers %llu candidates %llu hsps %llu\n",
                    (unsigned long long)total_h, c.nwalked, total_h ? 100.0 * c.nwalked / total_h : 0.0, c.nmed, c.nlong, c.nfollow,
                    c.ncand, c.nhsp);
        if (c.ncand <= cand_cap) break;
        if (attempt || chunk) { set_error("candidate buffer overflow"); return MIMEO_ERR_LIMIT; }
        cand_cap = c.ncand + 1024;  // rerun with room for every candidate
    }
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    *nhsp = c.nhsp;
    if (ms) {
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, e0, e1));
        *ms += t;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
}

}  // namespace mimeo

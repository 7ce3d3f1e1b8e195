// k4_device.h — device-side building blocks of the gap-free extension (K4): the exact x-drop walks, the pre-filter
// bounds, the wavefront-cooperative long walks and the record emitters.  Shared by the tail kernels of
// k4_extend.hip and by the fused seed-scan / extension kernel (k34_fused.hip).
#pragma once
#include "device_util.h"

namespace mimeo {

constexpr uint32_t CARE19 = 0x7A997u;  // care positions of 1110100110010101111 (bit i = offset i)
constexpr int EXT_THREADS = 256;
// K34's first pass when MIMEO_K34_FORM is not set (k4_extend.hip reads it per batch): 0 = tiles cut at the middle key + level
// emission (k34_fused.hip; C4 row: K34 147.1 -> 136.1 ms per launch, every GPU test green under it); 96 = tiles cut by entry
// count, prefix sum + lane-major descriptor emission: the form of rounds 2 and 3 up to profiles/r03_bench_c4_rows_line.json
constexpr uint32_t K34_FORM_DEFAULT = 0u;
constexpr int LONG_WINDOWS = 8;  // per-lane walks give up after 8*32 bases per direction

// The extension stage works on a BATCH of units (one unit = one (target scaffold, query scaffold, strand)): the
// heavy kernel of every unit appends to batch-wide queues, and the small latency-bound kernels behind it (walks
// beyond the frame, follower sort and resolution, entropy) run once per batch.  A record names its unit; the
// kernels look the two strands up in the unit table.
struct ExtCounters {
    unsigned long long ncand, nfollow, nlong, nhsp, nmed, nbig, nwalked;  // nwalked: hits the pre-filter let through
    // The two busy queues — followers and hits whose walk outlives the frame — are filled in eight shards (workgroup
    // number mod 8, i.e. per XCD: region r = [r * cap, ...)): a same-address atomic serialises at ~13 ns and a C4 unit
    // makes ~10^4 flushes.  nfollow / nmed hold the totals once k4_compact has gathered the follower shards.
    unsigned long long nfollow8[8], nmed8[8];
    unsigned long long nbigcand;   // candidates longer than ENT_LONG columns: their entropy is counted by the whole grid
    unsigned long long long_next, seg_next;  // k4_extend_long / k4_resolve_segments: next long hit / large segment to hand out
    unsigned long long nwalk_total, nwalk_over;  // walk-queue entries of the batch; fullest shard of any unit, in 1/1024 of its capacity (> 1024: overflow)
    unsigned long long dbg[8];  // development (MIMEO_K34_DEBUG & 8): why the pre-filter passed a hit on
};

struct HeavyPlan {                       // the listed tiles of the whole batch, dense (k34_plan)
    const unsigned long long *est;       // per listed slot (unit * NTILE + i): the first pass's estimate of the tile's hits
    uint2 *tile;                         // dense k -> (unit index in the launch, tile)
    uint32_t *base;                      // dense k -> first part; base[ntiles] = all parts
    uint4 *split;                        // dense k -> (target shares, query ranges, entries per query range, 0)
    unsigned int *ctr;                   // [0] parts handed out, [1] listed tiles, [2] parts
};
struct Cand {
    uint32_t tstart, qstart, len;
    int32_t raw;  // RAW_SATURATED: the score does not fit (a gap-free segment beyond ~21 Mbp); k4_entropy recounts it in 64 bits
    uint32_t unit;
};
constexpr int32_t RAW_SATURATED = 0x7FFFFFFF;

// Batch-wide output queues of the heavy kernels and of the walks behind them.  Every append is guarded by the queue's
// capacity while the counter keeps counting, so the host sees an overflow as "counter > capacity" and repeats the
// batch with room for everything.  Follower key = unit << (dbits + ebits) | (diagonal + Q.len) << ebits | seed end:
// one radix sort over dbits + ebits + unit bits orders the followers of every unit of the batch.
struct ExtQueues {
    ExtCounters *ctr;
    Cand *cand;
    uint64_t *fkey;
    uint32_t *fprev;
    uint2 *medq, *longq;      // hits whose walk outlives the frame / LONG_WINDOWS windows
    uint32_t *medu, *longu;   // ... and their units
    uint2 *walkq;             // hits that the pre-filter of K34 could not dismiss: a region of eight shards per unit (FusedUnit)
    unsigned long long *nwalk_u;    // ... entries per unit and shard (workgroup number mod 8, i.e. per XCD)
    unsigned long long *nheavy_u;   // tiles per unit that K34's first pass left to its split pass
    uint32_t *heavy;                // ... their numbers: NTILE slots per unit
    unsigned long long *heavy_est;  // ... and the first pass's estimates of their hits (the split pass is planned from them)
    unsigned long long *bigcand;    // indices of the long candidates (ENT_BIGCAP) and their accumulators: 5 per candidate
    unsigned long long *bigacc;     // ... matched A / C / G / T columns, raw score
    unsigned long long *unit_hits;  // seed hits per unit (statistics)
    unsigned long long *tile_hits;  // ... per unit and tile (K34 writes them, k34_sum_hits folds them into unit_hits)
    uint64_t cand_cap, follow_cap, med_cap, long_cap, walk_cap;   // follow_cap, med_cap, walk_cap: per shard (shard r = base[r * cap ...])
    uint32_t ebits, dbits;
};
__device__ __forceinline__ uint32_t queue_shard() { return blockIdx.x & 7u; }
// a slot of an append-only queue for every lane that calls this together: one atomic per wavefront (same-address
// atomics serialise at ~13 ns each)
__device__ __forceinline__ unsigned long long wave_slot(unsigned long long *counter) {
    const uint64_t m = __ballot(true);   // the lanes active here
    const uint32_t lane = threadIdx.x & 63u, lead = (uint32_t)__builtin_ctzll(m);
    unsigned long long b = 0;
    if (lane == lead) b = atomicAdd(counter, (unsigned long long)__popcll(m));
    b = __shfl(b, (int)lead);
    return b + __popcll(m & ((1ull << lane) - 1ull));
}
__device__ __forceinline__ uint64_t follow_key(const ExtQueues &q, uint32_t unit, int32_t d, uint32_t qlen, uint32_t et) {
    return ((uint64_t)unit << (q.dbits + q.ebits)) | ((uint64_t)(uint32_t)(d + (int32_t)qlen) << q.ebits) | (uint64_t)et;
}
__device__ __forceinline__ uint32_t key_unit(const ExtQueues &q, uint64_t k) { return (uint32_t)(k >> (q.dbits + q.ebits)); }
__device__ __forceinline__ uint32_t key_end(const ExtQueues &q, uint64_t k) { return (uint32_t)(k & ((1ull << q.ebits) - 1ull)); }
__device__ __forceinline__ uint64_t key_unit_diag(const ExtQueues &q, uint64_t k) { return k >> q.ebits; }
__device__ __forceinline__ int32_t key_diag(const ExtQueues &q, uint64_t k, uint32_t qlen) {
    return (int32_t)(uint32_t)((k >> q.ebits) & ((1ull << q.dbits) - 1ull)) - (int32_t)qlen;
}

// ---- runs of consecutive seed hits -------------------------------------------------------------------------------
// Inside real similarity — and wholesale inside microsatellites, where two arrays of one motif are a rectangle of seed hits —
// a diagonal carries seed hits at CONSECUTIVE positions: e, e + 1, e + 2, ...  Every one of them but the first has its
// left neighbour as nearest earlier seed hit (the left walk reaches the boundary one step back whatever the bases: no
// single column can trip the x-drop), i.e. is a follower with predecessor end e - 1, and a C5 unit has 3e8 of those to sort.
// A run is therefore recorded by its two ends only: the first of these followers as an ordinary follower record
// (seed end s, predecessor s - 1), the last one as a RUN_END record (seed end t, predecessor field RUN_END): between two
// such records EVERY position s + 1 .. t is a follower naming its left neighbour.  A member decides what it is from
// three exact seed tests on its own diagonal: the hits one back (it is such a follower at all), two back (it is not the
// first) and one ahead (it is not the last); interior members leave no record.  The segment flags and the two resolution
// kernels read a RUN_END record as the range it closes.
constexpr uint32_t RUN_END = 0xFFFFFFFFu;

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
static inline void build_group_table(uint32_t *tab) {
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

// up to 32 steps of a walk whose masks are in step order (bit s <-> step s of this window).  G = columns per table
// group: 4 (4096-entry table, 16 KiB) or 2 (64 entries: fits beside the frames in the LDS of the fused kernel).
template <int G = 4>
__device__ __forceinline__ void walk_window(const uint32_t *__restrict__ tab, WalkState &w, uint32_t mdl, uint32_t mdh,
                                            uint32_t mcg, uint32_t mnn, uint32_t mH, uint32_t limit, int xdrop,
                                            uint32_t start = 0) {
    constexpr uint32_t GM = (1u << G) - 1u;
    for (uint32_t pos = start; pos < 32;) {
        const uint32_t rem = limit - w.k;
        if (rem == 0) { w.done = true; return; }
        if (rem >= (uint32_t)G && pos <= 32u - G && !(((mnn | mH) >> pos) & GM)) {
            const uint32_t idx = ((mdl >> pos) & GM) | (((mdh >> pos) & GM) << G) | (((mcg >> pos) & GM) << (2 * G));
            const uint32_t e = tab[idx];
            const int32_t S = ((int32_t)(e << 22)) >> 22, M = ((int32_t)(e << 12)) >> 22, mn = ((int32_t)(e << 2)) >> 22;
            if (w.run + mn < w.best - xdrop) { w.done = true; return; }
            if (w.run + M > w.best) { w.best = w.run + M; w.bk = w.k + (e >> 30) + 1; }
            w.run += S;
            w.k += G;
            pos += G;
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

// The same 32 steps as walk_window, as straight-line code without per-lane branches: the table
// groups are applied to running values unconditionally, and the state in front of the first group that
// cannot be applied — x-drop inside it, an N / earlier seed hit / the sequence end in it, or a walk that
// was already finished — is kept aside (one select per value and group).  A lane stopped by a blocked
// group finishes the window in walk_window (rare; taken under a wave-uniform branch).
template <int G = 4>
__device__ __forceinline__ void walk_window_pred(const uint32_t *__restrict__ tab, WalkState &w, uint32_t mdl,
                                                 uint32_t mdh, uint32_t mcg, uint32_t mnn, uint32_t mH, uint32_t limit,
                                                 int xdrop) {
    constexpr uint32_t GM = (1u << G) - 1u;
    constexpr int NG = 32 / G, LOG = G == 4 ? 2 : 1;
    static_assert(G == 4 || G == 2, "group size");
    const uint32_t blocked = mnn | mH;
    // the table entries depend only on the masks: fetch them back to back, then run the
    // dependent score arithmetic on registers
    uint32_t ent[NG];
#pragma unroll
    for (int c = 0; c < NG; c++) {
        const int pos = G * c;
        ent[c] = tab[((mdl >> pos) & GM) | (((mdh >> pos) & GM) << G) | (((mcg >> pos) & GM) << (2 * G))];
    }
    uint32_t nz = blocked | (blocked >> 1);
    if (G == 4) nz |= nz >> 2;                    // bit G c: group c holds a blocked step ...
    const uint32_t ng = (limit - w.k) >> LOG;     // ... or does not fit below the limit any more (one test per
    nz |= ng >= (uint32_t)NG ? 0u : (0xFFFFFFFFu << ((uint32_t)G * ng));  // group instead of two: compares issue at half rate)
    int32_t R = w.run, B = w.best, sR = R, sB = B;
    uint32_t BK = 0, sBK = 0, sC = 0;             // BK: steps at the best prefix relative to w.k (0 = unchanged)
    bool stopped = w.done, slow = false, brkdone = false;
#pragma unroll
    for (int c = 0; c < NG; c++) {
        const uint32_t e = ent[c];
        const int32_t S = ((int32_t)(e << 22)) >> 22, M = ((int32_t)(e << 12)) >> 22, mn = ((int32_t)(e << 2)) >> 22;
        const bool blk = (nz >> (G * c)) & 1u;
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
        BK = cand > B ? (e >> 30) + (uint32_t)(G * c + 1) : BK;
        B = max(B, cand);
        R += S;
    }
    if (!stopped) { sR = R; sB = B; sBK = BK; sC = NG; }
    w.run = sR;
    w.best = sB;
    w.bk = sBK ? w.k + sBK : w.bk;
    w.k += (uint32_t)G * sC;
    w.done = w.done || brkdone;
    if (__ballot(slow)) {
        if (slow) walk_window<G>(tab, w, mdl, mdh, mcg, mnn, mH, limit, xdrop, (uint32_t)G * sC);
    }
}

// entry idx of the group table for G columns per group, computed on the device (the fused kernel builds the
// 64-entry table of G = 2 in LDS): score sum S, best prefix M and where, lowest prefix mn, packed as above
template <int G>
__device__ __forceinline__ uint32_t group_table_entry(uint32_t idx) {
    int32_t p = 0, M = -100000, mn = 100000, posM = 0;
#pragma unroll
    for (int k = 0; k < G; k++) {
        p += sub_score((idx >> k) & 1u, (idx >> (G + k)) & 1u, (idx >> (2 * G + k)) & 1u, 0u);
        if (p > M) { M = p; posM = k; }
        if (p < mn) mn = p;
    }
    return ((uint32_t)p & 0x3FFu) | (((uint32_t)M & 0x3FFu) << 10) | (((uint32_t)mn & 0x3FFu) << 20) | ((uint32_t)posM << 30);
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
        if (!((CARE19 >> c) & 1u)) continue;
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
        r_fk = ((uint64_t)(uint32_t)(d + (int32_t)Q.len) << 32) | (uint32_t)et;  // the batch key is composed when the record is flushed
        r_fp = (uint32_t)et - L.found_step;
        if (L.found_step == 1) {   // a member of a run of consecutive seed hits: only the run's ends leave records
            const bool two_back = seed_hit_at(T, Q, (int32_t)h.x - 2, d, transitions), ahead = seed_hit_at(T, Q, (int32_t)h.x + 1, d, transitions);
            if (two_back && ahead) q_fol = false;     // interior
            else if (two_back) r_fp = RUN_END;        // the last one (and not the first)
        }
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
            r_cd = Cand{(uint32_t)et - L.bk, (uint32_t)eq - L.bk, L.bk + R.bk, L.best + R.best, 0u};
        }
    }
}

// ---- per-wavefront staging of the records the exact walks produce ---------------------------------------
// One global atomic serves >= 64 records instead of one per wavefront iteration (same-address atomics serialise
// at ~12-15 ns each).  s_med / s_fk / s_fp (and s_cd when STAGE_CAND) are QCAP-entry LDS arrays private to the
// wavefront (CAP entries for the frequent kinds, generic-walk hits and followers); the fill levels are wave-uniform
// registers of the caller.  A queue is flushed when the new records would not fit (a batch adds at most 64); `final`
// flushes what is left.  Follower records are
// staged as (diagonal + Q.len) << 32 | seed end and get their batch key (unit, bit widths) at the flush.
struct WaveFill { uint32_t n_med, n_fol, n_cd; };

__device__ __forceinline__ uint64_t batch_key(const ExtQueues &q, uint32_t unit, uint64_t staged) {
    return ((uint64_t)unit << (q.dbits + q.ebits)) | ((staged >> 32) << q.ebits) | (staged & 0xFFFFFFFFull);
}

template <bool STAGE_CAND, int CAP = QCAP>
__device__ __forceinline__ void stage_records(const ExtQueues &q, uint32_t unit, uint2 *s_med, uint64_t *s_fk, uint32_t *s_fp,
                                              Cand *s_cd, WaveFill &f, bool q_med, uint2 h, bool q_fol, uint64_t r_fk,
                                              uint32_t r_fp, bool q_cd, Cand r_cd, bool final) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint64_t m = __ballot(q_med);
    if (m || (final && f.n_med)) {
        const uint32_t add = (uint32_t)__popcll(m);
        if (f.n_med + add > (uint32_t)CAP || (final && !m)) {
            __builtin_amdgcn_wave_barrier();  // LDS accesses of one wavefront execute in order
            unsigned long long b = 0;
            if (lane == 0) b = atomicAdd(&q.ctr->nmed8[queue_shard()], (unsigned long long)f.n_med);
            b = __shfl(b, 0);
            for (uint32_t i = lane; i < f.n_med; i += 64)
                if (b + i < q.med_cap) { const size_t at = (size_t)queue_shard() * q.med_cap + b + i; q.medq[at] = s_med[i]; q.medu[at] = unit; }
            f.n_med = 0;
            __builtin_amdgcn_wave_barrier();
        }
        if (q_med) s_med[f.n_med + __popcll(m & lt_mask)] = h;
        f.n_med += add;
        if (final && f.n_med) {
            __builtin_amdgcn_wave_barrier();
            unsigned long long b = 0;
            if (lane == 0) b = atomicAdd(&q.ctr->nmed8[queue_shard()], (unsigned long long)f.n_med);
            b = __shfl(b, 0);
            for (uint32_t i = lane; i < f.n_med; i += 64)
                if (b + i < q.med_cap) { const size_t at = (size_t)queue_shard() * q.med_cap + b + i; q.medq[at] = s_med[i]; q.medu[at] = unit; }
            f.n_med = 0;
        }
    }
    m = __ballot(q_fol);
    if (m || (final && f.n_fol)) {
        const uint32_t add = (uint32_t)__popcll(m);
        if (f.n_fol + add > (uint32_t)CAP || (final && !m)) {
            __builtin_amdgcn_wave_barrier();
            unsigned long long b = 0;
            if (lane == 0) b = atomicAdd(&q.ctr->nfollow8[queue_shard()], (unsigned long long)f.n_fol);
            b = __shfl(b, 0);
            for (uint32_t i = lane; i < f.n_fol; i += 64)
                if (b + i < q.follow_cap) { const size_t at = (size_t)queue_shard() * q.follow_cap + b + i; q.fkey[at] = batch_key(q, unit, s_fk[i]); q.fprev[at] = s_fp[i]; }
            f.n_fol = 0;
            __builtin_amdgcn_wave_barrier();
        }
        if (q_fol) { const uint32_t i = f.n_fol + __popcll(m & lt_mask); s_fk[i] = r_fk; s_fp[i] = r_fp; }
        f.n_fol += add;
        if (final && f.n_fol) {
            __builtin_amdgcn_wave_barrier();
            unsigned long long b = 0;
            if (lane == 0) b = atomicAdd(&q.ctr->nfollow8[queue_shard()], (unsigned long long)f.n_fol);
            b = __shfl(b, 0);
            for (uint32_t i = lane; i < f.n_fol; i += 64)
                if (b + i < q.follow_cap) { const size_t at = (size_t)queue_shard() * q.follow_cap + b + i; q.fkey[at] = batch_key(q, unit, s_fk[i]); q.fprev[at] = s_fp[i]; }
            f.n_fol = 0;
        }
    }
    m = __ballot(q_cd);
    if (!STAGE_CAND) {
        // candidates are rare (~1e-4 of the hits): one aggregated atomic per batch that holds any
        if (m) {
            unsigned long long b = 0;
            if (lane == (uint32_t)__builtin_ctzll(m)) b = atomicAdd(&q.ctr->ncand, (unsigned long long)__popcll(m));
            b = __shfl(b, __builtin_ctzll(m));
            const unsigned long long i = b + __popcll(m & lt_mask);
            if (q_cd && i < q.cand_cap) { r_cd.unit = unit; q.cand[i] = r_cd; }
        }
    } else if (m || (final && f.n_cd)) {
        const uint32_t add = (uint32_t)__popcll(m);
        if (f.n_cd + add > (uint32_t)QCAP || (final && !m)) {
            __builtin_amdgcn_wave_barrier();
            unsigned long long b = 0;
            if (lane == 0) b = atomicAdd(&q.ctr->ncand, (unsigned long long)f.n_cd);
            b = __shfl(b, 0);
            if (lane < f.n_cd && b + lane < q.cand_cap) { Cand c = s_cd[lane]; c.unit = unit; q.cand[b + lane] = c; }
            f.n_cd = 0;
            __builtin_amdgcn_wave_barrier();
        }
        if (q_cd) s_cd[f.n_cd + __popcll(m & lt_mask)] = r_cd;
        f.n_cd += add;
        if (final && f.n_cd) {
            __builtin_amdgcn_wave_barrier();
            unsigned long long b = 0;
            if (lane == 0) b = atomicAdd(&q.ctr->ncand, (unsigned long long)f.n_cd);
            b = __shfl(b, 0);
            if (lane < f.n_cd && b + lane < q.cand_cap) { Cand c = s_cd[lane]; c.unit = unit; q.cand[b + lane] = c; }
            f.n_cd = 0;
        }
    }
    __builtin_amdgcn_wave_barrier();
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

// full extension of one hit by one wavefront: what it leaves (valid in lane 0) — nothing, a follower record or a candidate
struct HitRecord {
    int kind;   // 0 nothing, 1 follower, 2 candidate
    uint64_t fkey;
    uint32_t fprev;
    Cand c;
};
__device__ __forceinline__ HitRecord wave_extend_record(const StrandView &T, const StrandView &Q, uint2 h, int xdrop, int hspthresh,
                                                        int transitions, bool detect, const ExtQueues &q, uint32_t unit, uint32_t *rext_out) {
    HitRecord out;
    out.kind = 0;
    const int32_t et = (int32_t)h.x + SEED_LEN, eq = (int32_t)h.y + SEED_LEN, d = (int32_t)h.x - (int32_t)h.y;
    WalkResult L = detect ? wave_walk(T, Q, et, d, -1, (uint32_t)min(et, eq), xdrop, true, transitions)
                          : wave_walk_fast(T, Q, et, d, -1, (uint32_t)min(et, eq), xdrop);
    if (L.found) {
        out.kind = 1;
        out.fkey = follow_key(q, unit, d, Q.len, (uint32_t)et);
        out.fprev = L.prev_end;
        return out;
    }
    WalkResult R = wave_walk_fast(T, Q, et, d, +1, min(T.len - (uint32_t)et, Q.len - (uint32_t)eq), xdrop);
    if (rext_out) *rext_out = R.bsteps;
    int64_t score = L.best + R.best;
    if (score >= hspthresh) {
        out.kind = 2;
        out.c = Cand{(uint32_t)et - L.bsteps, (uint32_t)eq - L.bsteps, L.bsteps + R.bsteps,
                     score >= (int64_t)RAW_SATURATED ? RAW_SATURATED : (int32_t)score, unit};  // k4_entropy recounts a saturated score
    }
    return out;
}
// A wavefront's records, kept in LDS until 16 are waiting: a same-address atomic takes ~13 ns, a C5 batch has 3 * 10^6 long
// hits and as many large segments, and every one of them leaves records: one atomic per queue and 16 records instead
constexpr uint32_t STAGE = 16;
struct WaveStage {
    Cand c[STAGE];
    uint64_t fk[STAGE];
    uint32_t fp[STAGE];
};
__device__ __forceinline__ void stage_flush(WaveStage &S, uint32_t &nf, uint32_t &nc, const ExtQueues &q) {
    const uint32_t lane = threadIdx.x & 63u;
    __builtin_amdgcn_wave_barrier();   // lane 0's records, read by the other lanes of this wavefront
    if (nf) {
        unsigned long long i = 0;
        if (lane == 0) i = atomicAdd(&q.ctr->nfollow8[queue_shard()], (unsigned long long)nf);
        i = __shfl(i, 0) + lane;
        if (lane < nf && i < q.follow_cap) {
            i += (unsigned long long)queue_shard() * q.follow_cap;
            q.fkey[i] = S.fk[lane];
            q.fprev[i] = S.fp[lane];
        }
    }
    if (nc) {
        unsigned long long i = 0;
        if (lane == 0) i = atomicAdd(&q.ctr->ncand, (unsigned long long)nc);
        i = __shfl(i, 0) + lane;
        if (lane < nc && i < q.cand_cap) q.cand[i] = S.c[lane];
    }
    __builtin_amdgcn_wave_barrier();
    nf = nc = 0;
}
__device__ __forceinline__ void stage_push(WaveStage &S, uint32_t &nf, uint32_t &nc, const HitRecord &r, const ExtQueues &q) {
    const int kind = __builtin_amdgcn_readfirstlane(r.kind);
    if (kind == 1) { if ((threadIdx.x & 63u) == 0) { S.fk[nf] = r.fkey; S.fp[nf] = r.fprev; } nf++; }
    else if (kind == 2) { if ((threadIdx.x & 63u) == 0) S.c[nc] = r.c; nc++; }
    if (nf == STAGE || nc == STAGE) stage_flush(S, nf, nc, q);
}

// ... or appended at once (lane 0)
__device__ __forceinline__ void wave_extend_emit(const StrandView &T, const StrandView &Q, uint2 h, int xdrop, int hspthresh,
                                                 int transitions, bool detect, const ExtQueues &q, uint32_t unit, uint32_t *rext_out) {
    const HitRecord r = wave_extend_record(T, Q, h, xdrop, hspthresh, transitions, detect, q, unit, rext_out);
    if ((threadIdx.x & 63) != 0) return;
    if (r.kind == 1) {
        unsigned long long i = atomicAdd(&q.ctr->nfollow8[queue_shard()], 1ull);
        if (i < q.follow_cap) {
            i += (unsigned long long)queue_shard() * q.follow_cap;
            q.fkey[i] = r.fkey;
            q.fprev[i] = r.fprev;
        }
    } else if (r.kind == 2) {
        unsigned long long i = atomicAdd(&q.ctr->ncand, 1ull);
        if (i < q.cand_cap) q.cand[i] = r.c;
    }
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

}  // namespace mimeo

// K34 — seed scan and pre-filter of one unit in ONE kernel (SURVEY §8a A7 + A8: lastz's query scan with 13 probes per
// base and the first look at every seed hit; reference call site src/mimeo/wrappers.py:1025-1037,
// `--step=1 --strand=both --gfextend --entropy --hspthresh=H`).
//
// Round 1 materialised the hit flood: K3 wrote 77.8 M (tpos, qpos) records per 10 Mbp x 10 Mbp unit (0.9 GB of
// stores for 0.62 GB of records), K4 read them back and gathered 13 eight-byte words around every hit through L2
// — to prove for 95 % of them that nothing comes of them.  Here no hit is stored unless it matters:
//   * both seed indexes are CSR in the transition-closed key layout (common.h): a workgroup owns one tile of 4096
//     keys, stages the query tile's offsets in LDS (relative to the tile, 16 bits each: 8 KiB) and resolves all 13 probes of a
//     word there (K3's join);
//   * every index entry carries its seed FRAME (K2: 192 bases around the seed start, both planes, one common
//     alignment), so a tile's frames are contiguous: the query frames are STREAMED into LDS, coalesced, in segments
//     of QSEG entries, and each wavefront keeps the frames of 64 target entries in registers (one entry per lane,
//     the next chunk's already in flight);
//   * a wavefront enumerates the (target entry, query entry) pairs of its chunk into a small LDS queue of
//     descriptors — first pass: probe by probe and level by level (a ballot and v_mbcnt per level) into a ring of 128;
//     split pass: lane-major behind a prefix sum of the per-entry counts, long ranges densely — so that the rounds are
//     full wavefronts whatever the per-entry fan-out, which is Poisson(7.8) on a C4 unit; a lane takes one pair, fetches
//     the target frame from its owner lane
//     (ds_bpermute) and the query frame from LDS, and runs the pre-filter: twelve XORs bring the two frames
//     together, every shift of the popcount bounds and of the earlier-seed-hit test is a compile-time constant;
//   * the ~4 % of the pairs the filter cannot dismiss go, compacted per wavefront (ballot + prefix count), to the
//     unit's walk queue (k4_extend_hits on the queue: sharp filter + exact walk, right behind this kernel).
// HBM traffic per unit: the two offset arrays, positions + frames of both sides once (52 bytes per entry), and the target
// frames of a tile once more per further query segment (Infinity Cache).
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "k4_device.h"

namespace mimeo {

// the form of the first pass = bits 5 and 6 of the switch word (do_tile).  scripts/k34_isa_mix.py compiles ONE form alone
// (-DK34_ONLY_FORM=0) to read its ISA: the blocks of the form that does not run would dilute the opcode mix it prices.
#ifdef K34_ONLY_FORM
#define K34_FORM(dbg) ((uint32_t)(K34_ONLY_FORM))
#else
#define K34_FORM(dbg) ((dbg) & 96u)
#endif

constexpr uint32_t TCH = 64;      // target entries per wavefront and chunk: one per lane
constexpr uint32_t DQ = 192;      // pair descriptors per wavefront and round
constexpr uint32_t CARE10 = 0x1A997u;  // offsets 0 1 2 4 7 8 11 13 15 16 of CARE19
constexpr unsigned long long HEAVY_HITS = 262144;   // hits of a tile beyond which it is split (a tile of a 10 Mbp x 10 Mbp unit averages 19 000)
// The split pass works the listed tiles off in PARTS of about PART_HITS seed hits each (k34_plan: a tile's parts = shares of its
// target chunk groups x ranges of its query entries), dealt to persistent workgroups by an atomic counter: a poly-A tile of
// 6000 x 6000 entries (3.6e7 hits) is 280 parts, a tile just above the threshold two — the static 8 x 8 split of round 2 left the
// machine to the 64 workgroups of the largest tile (C5: 55 ps per seed hit in this pass against 13.5 in the first).
constexpr unsigned long long PART_HITS = 131072;
constexpr uint32_t HEAVY_GRID = 2048;  // persistent workgroups of the split pass (four per workgroup slot of the device)

// One launch works off EVERY unit of a batch (grid.y = the batch's units that have seed hits to look for): the machine never
// drains between units (a launch per unit left ~6 % of it idle in the tails of 4096 workgroups over 512 slots, and cost three
// launches per unit).  A unit's own things come from a device table (FusedUnit, k4_device.h): its two seed indexes, its
// number in the batch, and its region of the walk queue.
struct FusedArgs {
    const FusedUnit *units;
    ExtQueues q;
    HeavyPlan plan;
    int xdrop, hspthresh, transitions;
    uint32_t dbg;  // switch word.  Development (MIMEO_K34_DEBUG): 1 = no pre-filter arithmetic, 2 = nothing is passed on, 4 = no pair
                   // rounds, 8 = no descriptors either (lane-major emission), 16 = run members are not dropped; bits 5 and 6 (32, 64) =
                   // the form of the first pass (MIMEO_K34_FORM, k4_extend.hip)
};

// ---- pre-filter on two frames in the common alignment ------------------------------------------------------
// left walk = frame bits 127 .. 32 (four 24-step blocks), right walk = bits 128 .. 191 (24 + 24 + 16); the seed windows
// that END where left step s arrives start at bit 108 - s.  A pair is dismissed when (i) both walks provably stop
// inside the frame, (ii) no earlier seed hit of the diagonal can end at a boundary the left walk reaches before its
// proven stop, so the hit is a head, and (iii) an upper bound of best(left) + best(right) is below hspthresh: the exact
// walk would then find an isolated head that scores too little, and emit nothing.  Everything else goes to the batch's
// walk queue (k4_walk_queue: the exact walk).  The proofs use the CHEAP bounds — two popcounts per block: n =
// mismatching columns, v = transversions; a match scores 91..100, a transition -31, a transversion -125..-114 — and
// a seed test on ten of the twelve care positions without the transition rule: 4.1 % of random hits are walked
// instead of 1.4 % with the sharpest bounds, for half the instructions per pair (scripts/dev_filter_mc.py; the
// kernel is bound by VALU issue, and popcounts, funnel shifts, compares and max issue at half rate).
struct Bound2 {
    int32_t up, lomax, ub;   // upper bound of the prefix score; best lower bound at an earlier checkpoint; bound of the best prefix
    bool stop;
};
// one block of W columns: v = its transversion bits, n = its mismatch bits
template <int W>
__device__ __forceinline__ void bound_block2(Bound2 &B, uint32_t v, uint32_t n, int32_t &lo, int xdrop) {
    const int32_t pv = __popc(v), pn = __popc(n);
    B.ub = max(B.ub, B.up + 100 * W - 100 * pn);
    B.up += 100 * W - 131 * pn - 83 * pv;
    lo += 91 * W - 122 * pn - 94 * pv;
    B.stop = B.stop || (B.up + xdrop < B.lomax);
    B.lomax = max(B.lomax, lo);
}

// "a seed hit of this diagonal may start at frame bit B + i" for i = 0 .. 31 (bit i of the result), B = 13 + 32 k.
// n0 / n1: the words of nm = dl | dh that hold bits B - 13 .. B + 50; d0 / d1: the same of dl.  CARE = the care
// positions looked at (all twelve + TV: the exact 12of19 rule with one transition; fewer / no TV: a superset).
template <uint32_t CARE, bool TV>
__device__ __forceinline__ uint32_t seed_starts32(uint32_t n0, uint32_t n1, uint32_t d0, uint32_t d1, int transitions) {
    uint32_t ones = 0, twos = 0, tv = 0;
#pragma unroll
    for (int c = 0; c < SEED_LEN; c++) {
        if (!((CARE >> c) & 1u)) continue;
        const uint32_t v = __builtin_amdgcn_alignbit(n1, n0, 13 + c);
        twos |= ones & v;
        ones |= v;
        if (TV) tv |= __builtin_amdgcn_alignbit(d1, d0, 13 + c);
    }
    return ~(transitions ? (twos | tv) : ones);
}

// true = the pair is walked exactly.  Blocks of 24 columns (left: four, right: 24 + 24 + 16): seven blocks instead of
// the ten 16-column ones pass 4.1 % of random hits on instead of 3.8 % (scripts/dev_filter_mc.py) for 15 % fewer
// instructions per pair.
__device__ __forceinline__ bool pair_needs_walk(const uint4 t0, const uint4 t1, const uint4 t2, const uint4 q0, const uint4 q1,
                                                const uint4 q2, int xdrop, int hspthresh, int transitions) {
    // planes: lo = {0.x 0.y 0.z 0.w 1.x 1.y}, hi = {1.z 1.w 2.x 2.y 2.z 2.w}
    const uint32_t dl0 = t0.x ^ q0.x, dl1 = t0.y ^ q0.y, dl2 = t0.z ^ q0.z, dl3 = t0.w ^ q0.w, dl4 = t1.x ^ q1.x, dl5 = t1.y ^ q1.y;
    const uint32_t n0 = dl0 | (t1.z ^ q1.z), n1 = dl1 | (t1.w ^ q1.w), n2 = dl2 | (t2.x ^ q2.x), n3 = dl3 | (t2.y ^ q2.y),
                   n4 = dl4 | (t2.z ^ q2.z), n5 = dl5 | (t2.w ^ q2.w);
    // earlier seed hits: starts 77..108 <-> left steps 31..0, 45..76 <-> steps 63..32, 13..44 <-> steps 95..64
    const uint32_t h0 = seed_starts32<CARE10, false>(n2, n3, 0u, 0u, transitions);   // bit 31 - s  <-> step s      (s = 0..31)
    const uint32_t h1 = seed_starts32<CARE10, false>(n1, n2, 0u, 0u, transitions);   // bit 63 - s  <-> step s      (s = 32..63)
    const uint32_t h2 = seed_starts32<CARE10, false>(n0, n1, 0u, 0u, transitions);   // bit 95 - s  <-> step s      (s = 64..95)
    Bound2 L{0, 0, 0, false}, R{0, 0, 0, false};
    int32_t llo = 0, rlo = 0;
    uint32_t veto;
    constexpr uint32_t M24 = 0xFFFFFFu;
    // left walk: columns 127 downwards; a boundary only counts while the walk is not yet proven to have stopped
    veto = h0 >> 8;                                                   // steps 0 .. 23
    bound_block2<24>(L, dl3 >> 8, n3 >> 8, llo, xdrop);                                                               // columns 104 .. 127
    veto |= L.stop ? 0u : ((h0 & 0xFFu) | (h1 >> 16));                // steps 24 .. 47
    bound_block2<24>(L, __builtin_amdgcn_alignbit(dl3, dl2, 16) & M24, __builtin_amdgcn_alignbit(n3, n2, 16) & M24, llo, xdrop);   // 80 .. 103
    veto |= L.stop ? 0u : ((h1 & 0xFFFFu) | (h2 >> 24));              // steps 48 .. 71
    bound_block2<24>(L, __builtin_amdgcn_alignbit(dl2, dl1, 24) & M24, __builtin_amdgcn_alignbit(n2, n1, 24) & M24, llo, xdrop);   // 56 .. 79
    veto |= L.stop ? 0u : (h2 & M24);                                 // steps 72 .. 95
    bound_block2<24>(L, dl1 & M24, n1 & M24, llo, xdrop);                                                              // 32 .. 55
    // right walk: columns 128 upwards
    bound_block2<24>(R, dl4 & M24, n4 & M24, rlo, xdrop);                                                              // 128 .. 151
    bound_block2<24>(R, __builtin_amdgcn_alignbit(dl5, dl4, 24) & M24, __builtin_amdgcn_alignbit(n5, n4, 24) & M24, rlo, xdrop);   // 152 .. 175
    bound_block2<16>(R, dl5 >> 16, n5 >> 16, rlo, xdrop);                                                              // 176 .. 191
    return !(L.stop && R.stop && L.ub + R.ub < hspthresh && veto == 0);
}

// Split pass only (the tiles microsatellites and high-copy repeats fill): is this pair an INTERIOR member of a run of consecutive
// seed hits of its diagonal — exact seed hits one and two positions back and one ahead (k4_device.h, "runs of consecutive seed
// hits")?  Such a hit leaves no record anywhere, so it is dropped here, before the pre-filter and the walk queue.  The three
// 19-windows start at frame bits 107, 108 and 110 (the hit's own at 109): words 3 and 4 of the planes.  Exact for frames
// without N of a unit whose target has no soft-masked bases, with the windows inside both scaffolds (the caller checks):
// the same predicate seed_hit_at() evaluates on the planes for the members that do reach the walk kernel.
__device__ __forceinline__ bool run_interior(const uint4 t0, const uint4 t1, const uint4 t2, const uint4 q0, const uint4 q1, const uint4 q2,
                                             int transitions) {
    // planes: lo = {0.x 0.y 0.z 0.w 1.x 1.y}, hi = {1.z 1.w 2.x 2.y 2.z 2.w}
    const uint32_t dl3 = t0.w ^ q0.w, dl4 = t1.x ^ q1.x;
    const uint32_t n3 = dl3 | (t2.y ^ q2.y), n4 = dl4 | (t2.z ^ q2.z);
    bool all = true;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int off = k == 0 ? 11 : (k == 1 ? 12 : 14);   // frame bit of the window start - 96
        const uint32_t wn = __builtin_amdgcn_alignbit(n4, n3, off) & CARE19, wd = __builtin_amdgcn_alignbit(dl4, dl3, off) & CARE19;
        all = all && (transitions ? (wd == 0u && __popc(wn) <= 1) : (wn == 0u));
    }
    return all;
}

constexpr uint32_t DENSE_MIN = 48;   // split pass: a probe's range of this many entries or more is enumerated densely
__device__ __forceinline__ uint4 readlane4(const uint4 v, int lane) {   // lane: wave-uniform
    return make_uint4((uint32_t)__builtin_amdgcn_readlane((int)v.x, lane), (uint32_t)__builtin_amdgcn_readlane((int)v.y, lane),
                      (uint32_t)__builtin_amdgcn_readlane((int)v.z, lane), (uint32_t)__builtin_amdgcn_readlane((int)v.w, lane));
}
__device__ __forceinline__ uint4 bperm4(uint32_t src_lane, const uint4 v) {
    const int a = (int)(src_lane << 2);
    return make_uint4((uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)v.x), (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)v.y),
                      (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)v.z), (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)v.w));
}

// LDS of a workgroup: the query tile's offsets RELATIVE TO THE CURRENT SEGMENT and clamped to it (16 bits each: a key's
// range inside the segment is sQ[w] .. sQ[w + 1] whatever the size of the tile — one code path for tiles of one segment
// and of many, in both passes; rewritten from the offset array, L2-resident, for every segment), QSEG query frames
// (three 16-byte parts), one descriptor queue and one walk staging area per wavefront, the N flags of the segment.  First pass: 512 threads, two
// workgroups per CU, QSEG = 1280 (79 KiB each): the 2441 entries of an average C4 tile are two segments — every target
// chunk is visited once per segment (frames, key, 13 probes, prefix sum, descriptors: ~530 instructions for the
// visit), so three segments of 1024 cost a third more visits for the same pairs.
template <int THREADS, uint32_t QSEG, bool HEAVY>
struct FusedCfg {
    static constexpr int WAVES = THREADS / 64;
    static constexpr size_t OFF_BYTES = ((TILE_WORDS + 1) * 2 + 15) / 16 * 16;
    static constexpr size_t SMEM = OFF_BYTES + (size_t)QSEG * 48 + (size_t)WAVES * DQ * 4 + (size_t)WAVES * 64 * 8 + QSEG + 16;
};

// HEAVY = the split pass: the workgroups (x) loop over the tiles the first pass listed, y = share of a tile's target
// chunks, z = share of its query segments; 32-bit offsets.
template <int THREADS, uint32_t QSEG, bool HEAVY>
__global__ __launch_bounds__(THREADS, 4) void k34_scan_extend(FusedArgs A) {
    using Cfg = FusedCfg<THREADS, QSEG, HEAVY>;
    using qoff_t = uint16_t;
    constexpr int WAVES = THREADS / 64;
    static_assert((size_t)QSEG * 48 >= 2 * (TILE_WORDS + 4) * 4, "the count prologue stages both offset arrays in the frame area");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    qoff_t *sQ = reinterpret_cast<qoff_t *>(smem);                                      // TILE_WORDS + 1
    uint4 *sQF = reinterpret_cast<uint4 *>(smem + Cfg::OFF_BYTES);                       // 3 parts of QSEG
    uint32_t *sD_all = reinterpret_cast<uint32_t *>(sQF + QSEG * 3);                     // WAVES * DQ
    uint2 *s_walk_all = reinterpret_cast<uint2 *>(sD_all + WAVES * DQ);                  // WAVES * 64
    uint8_t *sQN = reinterpret_cast<uint8_t *>(s_walk_all + WAVES * 64);                 // QSEG
    unsigned long long *s_total = reinterpret_cast<unsigned long long *>(sQN + QSEG);
    __shared__ unsigned long long s_slot;

    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    // first pass: grid (tiles, units); split pass: grid (HEAVY_MAX x units, target shares, query shares)
    uint32_t ai = HEAVY ? 0u : blockIdx.y;
    FusedUnit U = A.units[ai];   // by value: wave-uniform registers, not a reload through the pointer at every use (split pass: set per part)
    // split pass: my part of the tile = target chunk groups p, p + Pt, ... and the query entries [qa, qb)
    uint32_t part_p = 0, part_pt = 1, part_qa = 0, part_qb = 0xFFFFFFFFu, part_no = 0;
    uint32_t *sD = sD_all + wv * DQ;
    uint2 *s_walk = s_walk_all + wv * 64;
    uint32_t n_walk = 0;

    // the pairs the filter passed on go to the batch's walk queue, 64 at a time (one atomic per flush)
    auto flush_walk = [&](uint32_t n) {
        __builtin_amdgcn_wave_barrier();
        unsigned long long b = 0;
        // (split pass: the 64 workgroups that share a tile have one blockIdx.x — spread them, or a microsatellite tile fills one shard)
        const uint32_t shard = (HEAVY ? blockIdx.x + part_no : blockIdx.x) & 7u;
        if (lane == 0) b = atomicAdd(&A.q.nwalk_u[(size_t)ai * 8 + shard], (unsigned long long)n);
        b = __shfl(b, 0);
        // (a staged pair names its query ENTRY: the positions of all 64 are fetched here, one gather per flush instead of one per round)
        if (lane < n && b + lane < U.walk_cap) {
            uint2 e = s_walk[lane];
            e.y = U.Q.pos[e.y] & POS_MASK;
            A.q.walkq[U.walk_base + (size_t)shard * U.walk_cap + b + lane] = e;
        }
        __builtin_amdgcn_wave_barrier();
    };

    // everything below returns for the whole workgroup at once (the conditions are uniform)
    auto do_tile = [&](const uint32_t tile) {
        const uint32_t *toff = U.T.off + (size_t)tile * TILE_WORDS, *qoff = U.Q.off + (size_t)tile * TILE_WORDS;
        const uint32_t t0 = toff[0], nT = toff[TILE_WORDS] - t0;
        const uint32_t q0 = qoff[0], nQ = qoff[TILE_WORDS] - q0;
        if (!nT || !nQ) return;   // tile_hits is zeroed per batch
        if (!HEAVY) {
            // A tile with far more hits than the average — a microsatellite's seed words: thousands of entries of ONE key on both
            // sides, 10^7-10^8 hits in one tile — or with more query entries than 16-bit offsets can name is not worked off by
            // one workgroup while the chip waits: it is listed for the split pass.  The test is an ESTIMATE made up front from
            // the exact-key hits alone (one probe per word instead of thirteen; both offset arrays pass through the frame area
            // of LDS once): 13 x sum nT(w) nQ(w).  Which pass takes a tile changes nothing in the results; the exact count of a
            // tile — the seed_hits statistic — is the sum of the pairs its chunk visits enumerate (below).
            uint32_t *sT = reinterpret_cast<uint32_t *>(sQF), *sQ32 = sT + TILE_WORDS + 4;
            {
                const uint4 *s0 = reinterpret_cast<const uint4 *>(toff), *s1 = reinterpret_cast<const uint4 *>(qoff);
                uint4 *d0 = reinterpret_cast<uint4 *>(sT), *d1 = reinterpret_cast<uint4 *>(sQ32);
                for (uint32_t k = threadIdx.x; k < TILE_WORDS / 4; k += THREADS) { d0[k] = s0[k]; d1[k] = s1[k]; }
                if (threadIdx.x == 0) { sT[TILE_WORDS] = t0 + nT; sQ32[TILE_WORDS] = q0 + nQ; *s_total = 0ull; }
            }
            __syncthreads();
            unsigned long long cnt = 0;
            for (uint32_t w = threadIdx.x; w < TILE_WORDS; w += THREADS)
                cnt += (unsigned long long)(sT[w + 1] - sT[w]) * (sQ32[w + 1] - sQ32[w]);
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
            if (lane == 0 && cnt) atomicAdd(s_total, cnt);
            __syncthreads();
            if (!*s_total && A.transitions) {   // no exact-key hit (sparse tiles of small scaffolds): any hit at all?  Otherwise the tile is done
                __syncthreads();
                unsigned long long c2 = 0;
                for (uint32_t w = threadIdx.x; w < TILE_WORDS; w += THREADS) {
                    const uint32_t nt = sT[w + 1] - sT[w];
                    if (nt)
                        for (int j = 0; j < SEED_WEIGHT; j++) { const uint32_t w2 = w ^ (1u << j); c2 += sQ32[w2 + 1] - sQ32[w2]; }
                }
                if (__syncthreads_or(c2 != 0) == 0) return;
            } else if (!*s_total) {
                return;
            }
            const unsigned long long tile_est = *s_total * (A.transitions ? (unsigned long long)(SEED_WEIGHT + 1) : 1ull);
            if (tile_est > HEAVY_HITS || nQ > 0xFFFFu) {   // (a tile of 65536 query entries and more: 52 segments for one workgroup)
                if (threadIdx.x == 0) {   // at most NTILE entries per unit
                    const size_t slot = (size_t)ai * NTILE + atomicAdd(&A.q.nheavy_u[ai], 1ull);
                    A.q.heavy[slot] = tile;
                    A.q.heavy_est[slot] = tile_est;
                }
                return;
            }
        }
        // my share of the tile's query offsets (relative to its first entry), kept in registers for all its segments: from the
        // copy the count prologue staged in LDS (first pass; read before the frames overwrite it: the segment loop opens with
        // a barrier), or from the offset array
        constexpr int NRO = (TILE_WORDS + THREADS) / THREADS;   // 4097 offsets over the workgroup's threads
        uint32_t ro[NRO];
#pragma unroll
        for (int i = 0; i < NRO; i++) {
            const uint32_t k = threadIdx.x + (uint32_t)i * THREADS;
            ro[i] = 0;
            if (k <= TILE_WORDS) ro[i] = (HEAVY ? qoff[k] : (k < TILE_WORDS ? (reinterpret_cast<const uint32_t *>(sQF) + TILE_WORDS + 4)[k] : q0 + nQ)) - q0;
        }
        // First pass, a tile of two segments (every C4 tile): the cut is put at key TILE_WORDS / 2 when both halves fit a segment.
        // A key and its 13 probes differ in one bit, so in the segment of half h an entry whose key lies in h finds ALL its
        // partners there but the one of bit 11, and an entry of the other half only that one: 13 probes per entry and tile
        // instead of 26 (a chunk of one kind takes 12 probes or 1, wave-uniformly; the one mixed chunk of a tile all 13 —
        // the clamped offsets make every probe right in any segment).  Bit 5 (32) of the switch word: off (tiles cut by entry count).
        bool aligned = false;
        uint32_t mid = 0;
        if (!HEAVY && nQ > QSEG && !(K34_FORM(A.dbg) & 32u)) {
            mid = (uint32_t)__builtin_amdgcn_readfirstlane((int)((reinterpret_cast<const uint32_t *>(sQF) + TILE_WORDS + 4)[TILE_WORDS / 2] - q0));
            aligned = mid <= QSEG && nQ - mid <= QSEG;   // (then 0 < mid < nQ: both segments hold entries)
        }
        const uint4 *tF0 = U.T.fr + t0, *tF1 = tF0 + U.T.fr_stride, *tF2 = tF1 + U.T.fr_stride;
        const uint32_t nchunks = (nT + TCH - 1) / TCH;
        // chunks of this workgroup: every one (first pass), or those of my share of the tile (split pass)
        const uint32_t ch_first = (HEAVY ? part_p * WAVES : 0u) + wv, ch_step = (HEAVY ? part_pt : 1u) * WAVES;
        // my first chunk's frames (one entry per lane): in flight while the first query segment is staged
        uint4 nf0 = make_uint4(0, 0, 0, 0), nf1 = nf0, nf2 = nf0;
        uint32_t npos = 0;
        if (ch_first < nchunks) {
            const uint32_t e = ch_first * TCH + lane;
            if (e < nT) { nf0 = tF0[e]; nf1 = tF1[e]; nf2 = tF2[e]; npos = U.T.pos[t0 + e]; }
        }
        unsigned long long hits_acc = 0;   // pairs this wavefront enumerates in this tile (wave-uniform): the tile's exact hit count
        // 64 pairs, one per lane: the target entry's frame and position, the query entry qi of the current segment (first entry qs)
        auto pair_round = [&](const bool valid, const uint4 ta, const uint4 tb, const uint4 tc, const uint32_t tpf, const uint32_t qi, const uint32_t qs) {
            const uint4 qa = sQF[qi], qb = sQF[QSEG + qi], qc = sQF[2 * QSEG + qi];
            const uint32_t qflag = sQN[qi];
            bool need = false;
            uint32_t qp = 0;
            bool live = valid;   // still to be looked at by the pre-filter
            if (HEAVY && !(A.dbg & 16u)) {
                // interior members of runs of consecutive seed hits leave no record: out, before the filter (a microsatellite
                // tile is nearly all of them: whole rounds skip the filter)
                bool inner = false;
                if (valid && ((tpf >> 31) | qflag) == 0 && U.Tv.svt == nullptr && run_interior(ta, tb, tc, qa, qb, qc, A.transitions)) {
                    const uint32_t tp = tpf & POS_MASK;
                    inner = tp >= 2u && tp + 1u + SEED_LEN <= U.Tv.len;   // (the query's ends: bit 1 of its flag)
                }
                live = valid && !inner;
            }
            if (__ballot(live)) {
                if (live) {
                    if (A.dbg & 1u) need = (ta.x ^ qa.x ^ tb.y ^ qb.y ^ tc.z ^ qc.z) == 0x12345u;
                    else need = ((tpf >> 31) | (qflag & 1u)) != 0 ||
                                pair_needs_walk(ta, tb, tc, qa, qb, qc, A.xdrop, A.hspthresh, A.transitions);
                    if (A.dbg & 2u) need = false;
                }
            }
            if (U.same) {  // the main diagonal of a self unit belongs to k4_diag0 (one unit in 2 S: the slow way will do)
                if (valid) {
                    qp = U.Q.pos[q0 + qs + qi] & POS_MASK;
                    if ((tpf & POS_MASK) == qp) need = false;
                }
            }
            const uint64_t m = __ballot(need);
            if (m) {
                const uint32_t add = (uint32_t)__popcll(m);
                if (n_walk + add > 64u) { flush_walk(n_walk); n_walk = 0; }
                if (need) s_walk[n_walk + __popcll(m & lt_mask)] = make_uint2(tpf & POS_MASK, q0 + qs + qi);   // the query entry: flush_walk looks its position up
                n_walk += add;
            }
        };
        const uint32_t q_begin = HEAVY ? part_qa : 0u, q_end = HEAVY ? min(part_qb, nQ) : nQ;
        for (uint32_t qs = q_begin, qe; qs < q_end; qs = qe) {
            qe = aligned ? (qs ? q_end : mid) : min(qs + QSEG, q_end);
            const uint32_t qn = qe - qs;
            const uint32_t half = qs ? 1u : 0u;   // aligned tiles: the half of the key space this segment holds
            __syncthreads();  // every wavefront is through with the previous segment (and with the count prologue's use of the frame area)
            // the tile's offsets relative to this segment, clamped to it: key w's entries inside the segment are sQ[w] .. sQ[w + 1]
#pragma unroll
            for (int i = 0; i < NRO; i++) {
                const uint32_t k = threadIdx.x + (uint32_t)i * THREADS;
                if (k <= TILE_WORDS) sQ[k] = (qoff_t)(min(max(ro[i], qs), qe) - qs);
            }
            {
                const uint4 *s0 = U.Q.fr + q0 + qs, *s1 = s0 + U.Q.fr_stride, *s2 = s1 + U.Q.fr_stride;
                for (uint32_t i = threadIdx.x; i < qn; i += THREADS) {
                    sQF[i] = s0[i];
                    sQF[QSEG + i] = s1[i];
                    sQF[2 * QSEG + i] = s2[i];
                    // bit 0: N in the frame; bit 1: too close to an end of the scaffold for the run test's three seed windows
                    const uint32_t pq = U.Q.pos[q0 + qs + i], pp = pq & POS_MASK;
                    sQN[i] = (uint8_t)((pq >> 31) | ((pp >= 2u && pp + 1u + SEED_LEN <= U.Qv.len) ? 0u : 2u));
                }
            }
            __syncthreads();
            for (uint32_t ch = ch_first; ch < nchunks; ch += ch_step) {
                const uint32_t e0 = ch * TCH, ne = min(TCH, nT - e0);
                const uint4 f0 = nf0, f1 = nf1, f2 = nf2;
                const uint32_t mypos = npos;
                {   // next chunk of this wavefront: the following one of this segment, or its first one for the next segment
                    uint32_t nx = ch + ch_step;
                    if (nx >= nchunks) nx = (qe < q_end) ? ch_first : 0xFFFFFFFFu;
                    if (nx != 0xFFFFFFFFu && nx != ch) {
                        const uint32_t e = nx * TCH + lane;
                        if (e < nT) { nf0 = tF0[e]; nf1 = tF1[e]; nf2 = tF2[e]; npos = U.T.pos[t0 + e]; }
                    }
                }
                const bool tvalid = lane < ne;
                // my entry's in-tile key from its own frame: hi plane, seed window = frame bits 109 .. 127 (hi3 = f2.y)
                uint32_t w = 0, c = 0, nmask = 0, dmask = 0;   // dmask (split pass): probes whose range is enumerated densely, below
                if (tvalid) w = pext12(f2.y >> 13);
                const int nn = A.transitions ? SEED_WEIGHT + 1 : 1;
                // the probes this chunk needs in this segment: all of them, or — aligned tiles, a chunk whose keys lie in one half of
                // the key space (all but one chunk of a tile) — the key itself and its partners of bits 0 .. 10 when that is the
                // segment's half, else the partner of bit 11 alone
                if (!HEAVY && !(K34_FORM(A.dbg) & 64u)) {
                    int jlo = 0, jhi = nn;
                    if (aligned) {
                        const uint64_t in_half = __ballot(tvalid && (w >> 11) == half), in_other = __ballot(tvalid && (w >> 11) != half);
                        if (!in_other) jhi = min(nn, SEED_WEIGHT);
                        else if (!in_half) jlo = SEED_WEIGHT;
                    }
                    jlo = __builtin_amdgcn_readfirstlane(jlo);   // (wave-uniform by construction: say so)
                    jhi = __builtin_amdgcn_readfirstlane(jhi);
                    // ---- first pass: descriptors written LEVEL BY LEVEL, probe-major.  For probe j, level k = the lanes whose range
                    // holds more than k entries: their slots are ring position + v_mbcnt of the ballot, one LDS write each; a probe's
                    // range holds 0.6 entries on average on a C4 tile, so a probe is two or three levels (the ballot of the next one is
                    // empty).  No prefix sum over the lanes, no second visit of the offsets, no per-lane loop that runs as long as the
                    // busiest lane's: the emission was 20 % of this kernel's time.  The ring holds 128 descriptors: a round of 64 pairs
                    // runs as soon as 64 are waiting (bit 6 (64) of the switch word: the lane-major emission below, as in the split pass).
                    constexpr uint32_t RING = 128;
                    static_assert(RING <= DQ, "the ring lives in the wavefront's descriptor queue");
                    const uint32_t l16 = lane << 16;
                    uint32_t head = 0, tail = 0, pend = 0;   // ring positions (below RING) and descriptors waiting: wave-uniform
                    uint32_t an = 0, cn = 0;
                    auto fetch = [&](const int jj) {   // the range of probe jj in this segment, fetched a probe ahead of its use
                        an = 0; cn = 0;
                        if (tvalid && jj < jhi) {
                            const uint32_t w2 = jj ? (w ^ (1u << (jj - 1))) : w;
                            an = sQ[w2];
                            cn = (uint32_t)sQ[w2 + 1] - an;
                        }
                    };
                    auto round = [&](const uint32_t n) {   // the n <= 64 oldest descriptors of the ring
                        __builtin_amdgcn_wave_barrier();
                        const bool valid = lane < n;
                        const uint32_t d = valid ? sD[(head + lane) & (RING - 1u)] : 0u;
                        const uint32_t owner = d >> 16, qi = d & 0xFFFFu;
                        if (!(A.dbg & 4u)) {
                            const uint4 ta = bperm4(owner, f0), tb = bperm4(owner, f1), tc = bperm4(owner, f2);
                            const uint32_t tpf = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)mypos);
                            pair_round(valid, ta, tb, tc, tpf, qi, qs);
                        }
                        __builtin_amdgcn_wave_barrier();
                        head = (head + 64u) & (RING - 1u);
                        pend -= n;
                    };
                    fetch(jlo);
#pragma unroll 1
                    for (int j = jlo; j < jhi; j++) {
                        const uint32_t ak = l16 | an, cnt = cn;
                        fetch(j + 1);
                        uint32_t k = 0;
#pragma unroll 1
                        for (uint64_t m = __ballot(cnt > 0u); m; m = __ballot(cnt > k)) {
                            if (cnt > k) {
                                const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, tail));
                                sD[slot & (RING - 1u)] = ak + k;
                            }
                            const uint32_t add = (uint32_t)__popcll(m);
                            tail = (tail + add) & (RING - 1u);
                            pend += add;
                            hits_acc += add;
                            k++;
                            if (pend >= 64u) round(64u);   // (fewer than 64 waited before this level: the ring never overflows)
                        }
                    }
                    if (pend) round(pend);   // the rest of this chunk's pairs
                    continue;   // next chunk
                }
                {
                    auto probes = [&](const int jlo, const int jhi) {   // (constant bounds: unrolled, the offsets of four probes in flight)
                        if (!tvalid) return;
                        for (int j = jlo; j < jhi; j++) {
                            const uint32_t w2 = j ? (w ^ (1u << (j - 1))) : w;
                            const uint32_t a = sQ[w2], b = sQ[w2 + 1];
                            if (HEAVY && b - a >= DENSE_MIN) { dmask |= 1u << j; continue; }
                            c += b - a;
                            nmask |= (b != a ? 1u : 0u) << j;
                        }
                    };
                    const uint64_t in_half = (!HEAVY && aligned) ? __ballot(tvalid && (w >> 11) == half) : 1ull;
                    const uint64_t in_other = (!HEAVY && aligned) ? __ballot(tvalid && (w >> 11) != half) : 1ull;
                    if (in_half && in_other) probes(0, nn);
                    else if (in_half) probes(0, min(nn, SEED_WEIGHT));
                    else probes(SEED_WEIGHT, nn);
                }
                // inclusive prefix sum over the lanes: DPP (VALU latency), not six trips through the LDS crossbar
                uint32_t inc = c;
                inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, false);   // row_shr:1
                inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, false);   // row_shr:2
                inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, false);   // row_shr:4
                inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, false);   // row_shr:8
                inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x142, 0xa, 0xf, false);   // row_bcast:15
                inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x143, 0xc, 0xf, false);   // row_bcast:31
                const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63), st = inc - c;
                hits_acc += tot;
                for (uint32_t rb = 0; rb < tot; rb += DQ) {
                    if (!(A.dbg & 8u) && c && st < rb + DQ && st + c > rb) {
                        // a probe's range holds one or two entries nearly always (0.3 on average on a C4 tile, the empty ones
                        // are not in nmask): they are written outright, the loop is for the rest; a range that straddles the
                        // window of this round is clipped entry by entry
                        uint32_t acc = st;
                        const uint32_t wend = rb + DQ;
                        for (uint32_t m = nmask; m; m &= m - 1u) {
                            const uint32_t j = (uint32_t)__builtin_ctz(m);
                            const uint32_t w2 = j ? (w ^ (1u << (j - 1u))) : w;
                            const uint32_t a = sQ[w2], cnt = (uint32_t)sQ[w2 + 1] - a, d0 = (lane << 16) | a;
                            if (acc >= rb && acc + cnt <= wend) {
                                uint32_t *dst = sD + (acc - rb);
                                dst[0] = d0;
                                if (cnt > 1) {
                                    dst[1] = d0 + 1u;
#pragma unroll 1
                                    for (uint32_t k = 2; k < cnt; k++) dst[k] = d0 + k;
                                }
                            } else {
                                const uint32_t g0 = max(acc, rb), g1 = min(acc + cnt, wend);
#pragma unroll 1
                                for (uint32_t g = g0; g < g1; g++) sD[g - rb] = d0 + (g - acc);
                            }
                            acc += cnt;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t n = (A.dbg & 4u) ? 0u : min(DQ, tot - rb);   // development: 4 = no pair rounds, 8 = no descriptors either
                    for (uint32_t i = 0; i < n; i += 64) {
                        const bool valid = i + lane < n;
                        const uint32_t d = valid ? sD[i + lane] : 0u;
                        const uint32_t owner = d >> 16, qi = d & 0xFFFFu;
                        // the target frame lives in its owner lane's registers, the query frame in LDS
                        const uint4 ta = bperm4(owner, f0), tb = bperm4(owner, f1), tc = bperm4(owner, f2);
                        const uint32_t tpf = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)mypos);
                        pair_round(valid, ta, tb, tc, tpf, qi, qs);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                if (HEAVY) {
                    // Long ranges — a microsatellite's key holds hundreds of the segment's entries, and the split pass is made of
                    // such tiles — are not written out as descriptors: for one target entry at a time (its frame in scalar
                    // registers) the lanes take 64 consecutive query entries of the range, whose frames are conflict-free LDS
                    // reads.  No descriptor is written or read and no lane shuffle fetches a frame: the enumeration was 57 % of
                    // the split pass.
                    for (uint64_t lm = __ballot(dmask != 0u); lm; lm &= lm - 1ull) {
                        const int o = __builtin_ctzll(lm);
                        const uint4 ta = readlane4(f0, o), tb = readlane4(f1, o), tc = readlane4(f2, o);
                        const uint32_t tpf = (uint32_t)__builtin_amdgcn_readlane((int)mypos, o), wo = (uint32_t)__builtin_amdgcn_readlane((int)w, o);
                        for (uint32_t dm = (uint32_t)__builtin_amdgcn_readlane((int)dmask, o); dm; dm &= dm - 1u) {
                            const uint32_t j = (uint32_t)__builtin_ctz(dm), w2 = j ? (wo ^ (1u << (j - 1u))) : wo;
                            const uint32_t a = sQ[w2], b = sQ[w2 + 1];
                            hits_acc += b - a;
                            if (A.dbg & 4u) continue;
                            for (uint32_t i = a; i < b; i += 64u) {
                                const bool valid = i + lane < b;
                                pair_round(valid, ta, tb, tc, tpf, valid ? i + lane : a, qs);
                            }
                        }
                    }
                }
            }
        }
        if (lane == 0 && hits_acc) atomicAdd(&A.q.tile_hits[(size_t)U.unit * NTILE + tile], hits_acc);
    };

    if (!HEAVY) {
        do_tile(blockIdx.x);
    } else {
        // persistent workgroups: parts are handed out by a counter until none is left (every workgroup reaches the end: the loop
        // bound is the plan's part count, fixed before this launch)
        __shared__ uint32_t s_part;
        const uint32_t nparts = A.plan.ctr[2], ntiles = A.plan.ctr[1];
        for (;;) {
            __syncthreads();   // the part before is finished in every wavefront (s_part, the tile's offsets and frames are free)
            if (threadIdx.x == 0) s_part = atomicAdd(&A.plan.ctr[0], 1u);
            __syncthreads();
            const uint32_t g = s_part;
            if (g >= nparts) break;
            // the listed tile that holds part g: base[k] <= g < base[k + 1]
            uint32_t lo = 0, hi = ntiles;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (A.plan.base[mid] <= g) lo = mid; else hi = mid;
            }
            const uint2 ut = A.plan.tile[lo];
            const uint4 sp = A.plan.split[lo];
            const uint32_t part = g - A.plan.base[lo];
            if (n_walk && ut.x != ai) { flush_walk(n_walk); n_walk = 0; }   // the staged hits belong to the unit before
            ai = ut.x;
            U = A.units[ai];
            part_no = part;
            part_p = part % sp.x; part_pt = sp.x;
            part_qa = (part / sp.x) * sp.z; part_qb = part_qa + sp.z;
            do_tile(ut.y);
        }
    }
    if (n_walk) flush_walk(n_walk);
    (void)s_slot;
}

// one workgroup per unit: unit_hits[u] = sum of its tile counts (once per batch)
__global__ __launch_bounds__(256) void k34_sum_hits(const unsigned long long *__restrict__ tile_hits, unsigned long long *__restrict__ unit_hits) {
    __shared__ unsigned long long red[256];
    unsigned long long s = 0;
    for (uint32_t i = threadIdx.x; i < NTILE; i += 256) s += tile_hits[(size_t)blockIdx.x * NTILE + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) unit_hits[blockIdx.x] = red[0];
}
void launch_sum_hits(const ExtQueues &q, uint32_t nunits, hipStream_t st) {
    hipLaunchKernelGGL(k34_sum_hits, dim3(nunits), dim3(256), 0, st, (const unsigned long long *)q.tile_hits, q.unit_hits);
}

// The plan of the split pass, one workgroup for the whole launch (a few thousand listed tiles; the worst case, every tile of every
// unit listed, is a few milliseconds): the units' lists made dense, every tile cut into parts of about PART_HITS hits — target chunk
// groups (512 entries: one chunk per wavefront of a workgroup) dealt round-robin over Pt shares, query entries in ranges of a
// multiple of 64 — and the parts' prefix sum.
__global__ __launch_bounds__(1024) void k34_plan(const FusedUnit *__restrict__ units, uint32_t nunits, ExtQueues q, HeavyPlan P) {
    __shared__ uint32_t s_scan[1024];
    __shared__ uint32_t s_carry, s_tiles;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) { s_carry = 0; s_tiles = 0; }
    __syncthreads();
    for (uint32_t u = 0; u < nunits; u++) {
        const uint32_t nlist = (uint32_t)min((unsigned long long)NTILE, q.nheavy_u[u]);
        const FusedUnit U = units[u];
        for (uint32_t i0 = 0; i0 < nlist; i0 += 1024) {
            const uint32_t i = i0 + tid;
            uint32_t parts = 0, pt = 1, pq = 1, qlen = 0, tile = 0;
            if (i < nlist) {
                tile = q.heavy[(size_t)u * NTILE + i];
                const unsigned long long est = P.est[(size_t)u * NTILE + i];
                const uint32_t *toff = U.T.off + (size_t)tile * TILE_WORDS, *qoff = U.Q.off + (size_t)tile * TILE_WORDS;
                const uint32_t nT = toff[TILE_WORDS] - toff[0], nQ = qoff[TILE_WORDS] - qoff[0];
                const uint32_t want = (uint32_t)min((unsigned long long)(1u << 20), (est + PART_HITS - 1) / PART_HITS);
                const uint32_t groups = max(1u, (nT + 511u) / 512u);
                pt = min(groups, max(1u, want));
                const uint32_t need_q = max(1u, (want + pt - 1) / pt);
                qlen = max(64u, (((nQ + need_q - 1) / need_q) + 63u) & ~63u);
                pq = max(1u, (nQ + qlen - 1) / qlen);
                parts = pt * pq;
            }
            // exclusive scan of the parts over the 1024 lanes, carried from iteration to iteration
            s_scan[tid] = parts;
            __syncthreads();
            for (uint32_t o = 1; o < 1024; o <<= 1) {
                const uint32_t v = tid >= o ? s_scan[tid - o] : 0u;
                __syncthreads();
                s_scan[tid] += v;
                __syncthreads();
            }
            if (i < nlist) {
                const uint32_t k = s_tiles + (i - i0);
                P.tile[k] = make_uint2(u, tile);
                P.split[k] = make_uint4(pt, pq, qlen, 0u);
                P.base[k] = s_carry + s_scan[tid] - parts;
            }
            __syncthreads();
            if (tid == 1023) { s_carry += s_scan[1023]; s_tiles += min(1024u, nlist - i0); }
            __syncthreads();
        }
    }
    if (tid == 0) { P.base[s_tiles] = s_carry; P.ctr[0] = 0; P.ctr[1] = s_tiles; P.ctr[2] = s_carry; }
}

constexpr uint32_t QSEG_FIRST = 1280, QSEG_HEAVY = 1024;

// the heavy phase of a batch: `nactive` units (table d_units), both passes
int launch_fused_batch(const FusedUnit *d_units, uint32_t nactive, const ExtQueues &q, const HeavyPlan &plan, const mimeo_params *p, hipStream_t st,
                       uint32_t dbg) {
    if (!nactive) return 0;
    FusedArgs A;
    A.units = d_units; A.q = q;
    A.xdrop = p->xdrop; A.hspthresh = p->hspthresh; A.transitions = p->transitions;
    A.dbg = dbg;   // MIMEO_K34_DEBUG, read once per batch by the caller
    constexpr size_t smem_heavy = FusedCfg<512, QSEG_HEAVY, true>::SMEM, smem_first = FusedCfg<512, QSEG_FIRST, false>::SMEM;
    static std::once_flag attr_once;   // the library may be driven from any one thread at a time: still set exactly once
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, [&] {
        attr_err = hipFuncSetAttribute(reinterpret_cast<const void *>(k34_scan_extend<512, QSEG_FIRST, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_first);
        if (attr_err == hipSuccess)
            attr_err = hipFuncSetAttribute(reinterpret_cast<const void *>(k34_scan_extend<512, QSEG_HEAVY, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_heavy);
    });
    HIP_TRY(attr_err);
    // measured on a C4 unit (two segments per tile at 1280, three at 1216 and 1024): 1.44 / 1.51 / 1.54 ms for the heavy phase
    for (uint32_t u0 = 0; u0 < nactive; u0 += 32768u) {   // grid.y limit
        const uint32_t nu = std::min(32768u, nactive - u0);
        FusedArgs B = A;
        B.units = d_units + u0;
        B.q.nwalk_u = q.nwalk_u + (size_t)u0 * 8; B.q.nheavy_u = q.nheavy_u + u0; B.q.heavy = q.heavy + (size_t)u0 * NTILE;
        B.q.heavy_est = q.heavy_est + (size_t)u0 * NTILE;
        B.plan = plan;
        B.plan.est = B.q.heavy_est;
        hipLaunchKernelGGL((k34_scan_extend<512, QSEG_FIRST, false>), dim3(NTILE, nu), dim3(512), smem_first, st, B);
        hipLaunchKernelGGL(k34_plan, dim3(1), dim3(1024), 0, st, B.units, nu, B.q, B.plan);
        hipLaunchKernelGGL((k34_scan_extend<512, QSEG_HEAVY, true>), dim3(HEAVY_GRID), dim3(512), smem_heavy, st, B);
    }
    return 0;
}

}  // namespace mimeo

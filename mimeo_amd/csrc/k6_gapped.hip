// K6 — lastz `--gapped` (reference call site src/mimeo/wrappers.py:1031):
// every chained HSP becomes an anchor (centre of its best 31-column window); anchors are taken
// by decreasing HSP score, an anchor that lies inside the box of an earlier alignment is
// skipped, every other anchor is extended in both directions by a y-drop affine-gap DP
// (gap open 400 / extend 30, y-drop 9400) and the two halves are joined.
//
// Decomposition:
//   * a half extension is one JOB executed by one wavefront.  The DP is evaluated row by row;
//     lane l owns the 16-column strip (j / 16) % 64 == l of a 1024-column window that slides with
//     the first live column, all state (score + match/mismatch counts of the C and D planes) in
//     registers.  Inside a row the only horizontal dependency is the insertion state, computed
//     as a max-plus prefix scan of u_k = H_k + k*E (in-lane over the strip, then one cross-lane
//     scan).  Pruning: a cell scoring below (best of the rows above) - ydrop is dead.  Cells carry
//     (score, matches, mismatches), so identity needs no traceback.
//   * exact shortcut: if the two sequences are identical and N-free from the anchor to the end of
//     the shorter one, the diagonal is optimal (every column already scores its maximum) and the
//     DP is skipped — this is what makes the trivial (A,A) self alignment cheap.
//   * jobs of different anchors are independent, only the skip rule is ordered: each round takes
//     the next <= nbatch unskipped anchors of every group (k6_pick), extends them all (k6_dp),
//     then replays the skip rule in order over the batch (k6_resolve).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "device_util.h"

namespace mimeo {

constexpr int32_t NEG = -(1 << 30);
constexpr int32_t NEGH = -(1 << 29);
// columns per lane: 16 (1024-column window) in k6_dp, 32 (2048 columns) in the second-chance kernel

struct Cell {
    int32_t s;
    uint32_t nm, nx;
};
struct HalfResult {
    int32_t score;
    uint32_t i, j, nm, nx, overflow;
    uint32_t maxcols, rows;  // widest live band (columns from the window base) and rows evaluated: tuning statistics
    uint32_t base_lo, base_hi;  // k6_dp_any: what its rebased 32-bit cells stand above (score = base + score), 0 elsewhere
};
struct DpJob {
    uint32_t group, at, aq;
    int32_t dir;
    uint32_t slot, pad;  // index of this half's HalfResult: 2 * (hsp_begin + anchor rank) + side
};

// score of a half extension: the identical-suffix shortcut (rows == 0, i > 0) carries 64 bits
__device__ __forceinline__ int64_t half_score(const HalfResult &r) {
    return (r.rows == 0 && r.i > 0) ? (int64_t)(((uint64_t)r.maxcols << 32) | (uint32_t)r.score)
                                    : (int64_t)r.score + (int64_t)(((uint64_t)r.base_hi << 32) | r.base_lo);
}

__device__ __forceinline__ Cell cmax_left(const Cell &l, const Cell &r) { return r.s > l.s ? r : l; }  // ties -> left

// Cross-lane movement with DPP (VALU latency) instead of ds_bpermute (LDS-crossbar latency): the DP
// keeps rank == lane, so every scan / neighbour access is a fixed lane pattern.
// gfx9 DPP controls: row_shr:n = 0x110+n, wave_shr:1 = 0x138, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
template <int CTRL, int RMASK>
__device__ __forceinline__ Cell dpp_cell(const Cell &c) {
    Cell o;  // lanes without a valid source keep the identity (NEG, 0, 0)
    o.s = __builtin_amdgcn_update_dpp(NEG, c.s, CTRL, RMASK, 0xf, false);
    o.nm = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c.nm, CTRL, RMASK, 0xf, false);
    o.nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c.nx, CTRL, RMASK, 0xf, false);
    return o;
}
// inclusive max-scan over the 64 lanes, ties to the lower lane
__device__ __forceinline__ Cell wave_incl_maxscan(Cell v) {
    v = cmax_left(dpp_cell<0x111, 0xf>(v), v);
    v = cmax_left(dpp_cell<0x112, 0xf>(v), v);
    v = cmax_left(dpp_cell<0x114, 0xf>(v), v);
    v = cmax_left(dpp_cell<0x118, 0xf>(v), v);
    v = cmax_left(dpp_cell<0x142, 0xa>(v), v);
    v = cmax_left(dpp_cell<0x143, 0xc>(v), v);
    return v;
}
struct Best4 {
    int32_t s;
    uint32_t j, nm, nx;
};
template <int CTRL, int RMASK>
__device__ __forceinline__ Best4 dpp_best(const Best4 &c) {
    Best4 o;
    o.s = __builtin_amdgcn_update_dpp(NEG, c.s, CTRL, RMASK, 0xf, false);
    o.j = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c.j, CTRL, RMASK, 0xf, false);
    o.nm = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c.nm, CTRL, RMASK, 0xf, false);
    o.nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c.nx, CTRL, RMASK, 0xf, false);
    return o;
}
__device__ __forceinline__ Best4 bmax_left(const Best4 &l, const Best4 &r) { return r.s > l.s ? r : l; }
// row maximum with the smallest column on ties: columns grow with the lane, so "ties to the lower
// lane" is "smallest column"; the total ends up in lane 63
__device__ __forceinline__ Best4 wave_best(Best4 v) {
    v = bmax_left(dpp_best<0x111, 0xf>(v), v);
    v = bmax_left(dpp_best<0x112, 0xf>(v), v);
    v = bmax_left(dpp_best<0x114, 0xf>(v), v);
    v = bmax_left(dpp_best<0x118, 0xf>(v), v);
    v = bmax_left(dpp_best<0x142, 0xa>(v), v);
    v = bmax_left(dpp_best<0x143, 0xc>(v), v);
    Best4 t;
    t.s = __builtin_amdgcn_readlane(v.s, 63); t.j = (uint32_t)__builtin_amdgcn_readlane((int)v.j, 63);
    t.nm = (uint32_t)__builtin_amdgcn_readlane((int)v.nm, 63); t.nx = (uint32_t)__builtin_amdgcn_readlane((int)v.nx, 63);
    return t;
}

// WSTRIP query bits for columns jb .. jb+WSTRIP-1 (bit s <-> column jb+s); column j consumes query base
// aq + j - 1 (dir > 0) or aq - j (dir < 0).  Out-of-range columns read padding and are never used.
template <int WSTRIP>
__device__ __forceinline__ void load_qbits(const StrandView &Q, uint32_t aq, int dir, uint32_t jb, uint32_t lenB,
                                           uint32_t &qlo, uint32_t &qhi, uint32_t &qn) {
    constexpr uint32_t SMASK = WSTRIP == 32 ? 0xFFFFFFFFu : ((1u << (WSTRIP & 31)) - 1u);  // WSTRIP in {4, 16, 32}
    if (jb > lenB) { qlo = qhi = qn = 0; return; }
    if (dir > 0) {
        int32_t p = (int32_t)(aq + jb) - 1;
        const Win32 w = win32(Q, p);
        qlo = w.lo & SMASK; qhi = w.hi & SMASK; qn = w.nm & SMASK;
    } else {
        // bit t <-> position p + t <-> column jb + WSTRIP - 1 - t
        int32_t p = (int32_t)aq - (int32_t)jb - (WSTRIP - 1);
        const Win32 w = win32(Q, p);
        qlo = __brev(w.lo & SMASK) >> (32 - WSTRIP); qhi = __brev(w.hi & SMASK) >> (32 - WSTRIP);
        qn = __brev(w.nm & SMASK) >> (32 - WSTRIP);
    }
}

// One-sided y-drop affine extension by one wavefront (all lanes return the same result).
// Target bases of 32 consecutive DP rows i0 .. i0+31 (bit b <-> row i0 + b): one window load per 32 rows
// instead of a dependent global load in every row; the caller fetches one block ahead.
struct RowBases { uint32_t lo, hi, nm; };
__device__ __forceinline__ RowBases load_row_bases(const StrandView &T, uint32_t at, int dir, uint32_t i0) {
    if (dir > 0) {
        const Win32 w = win32(T, (int32_t)(at + i0 - 1u));
        return RowBases{w.lo, w.hi, w.nm};
    }
    const Win32 w = win32(T, (int32_t)at - (int32_t)i0 - 31);
    return RowBases{__brev(w.lo), __brev(w.hi), __brev(w.nm)};
}

template <int WSTRIP>
__device__ HalfResult wave_half_extend(const StrandView &T, const StrandView &Q, uint32_t at, uint32_t aq, int dir,
                                       int32_t O, int32_t E, int32_t Y, int32_t cap) {
    constexpr int WINDOW = 64 * WSTRIP;  // columns in the sliding window
    constexpr int WSHIFT = WSTRIP == 32 ? 5 : 4;
    static_assert((1 << WSHIFT) == WSTRIP, "WSTRIP must be 16 or 32");
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t lenA = dir > 0 ? T.len - at : at, lenB = dir > 0 ? Q.len - aq : aq;
    HalfResult best{0, 0, 0, 0, 0, 0, 0, 0};
    // ---- exact shortcut: identical, N-free to the end of the shorter sequence
    {
        const uint32_t n = min(lenA, lenB);
        const int32_t st = dir > 0 ? (int32_t)at : (int32_t)(at - n), sq = dir > 0 ? (int32_t)aq : (int32_t)(aq - n);
        bool ok = true;
        uint64_t ncg = 0;
        for (uint32_t k0 = 0; k0 < n; k0 += 64u * 32u) {
            uint32_t k = k0 + lane * 32u;
            if (k < n) {
                const Win32 tw = win32(T, st + (int32_t)k), qw = win32(Q, sq + (int32_t)k);
                const uint32_t tlo = tw.lo, thi = tw.hi;
                uint32_t bad = (tlo ^ qw.lo) | (thi ^ qw.hi) | tw.nm | qw.nm;
                uint32_t rem = n - k, mask = rem < 32 ? (1u << rem) - 1u : 0xFFFFFFFFu;
                if (bad & mask) ok = false;
                ncg += __popc((tlo ^ thi) & mask);
            }
            if (__ballot(!ok)) break;
        }
        if (!__ballot(!ok)) {
            for (int o = 32; o > 0; o >>= 1) ncg += __shfl_xor(ncg, o);
            uint64_t sc = 100ull * ncg + 91ull * ((uint64_t)n - ncg);
            // rows == 0 marks a shortcut result: its score is 64 bits wide, high word in maxcols (k6_resolve)
            best.score = (int32_t)(uint32_t)sc; best.maxcols = (uint32_t)(sc >> 32); best.i = n; best.j = n; best.nm = n; best.nx = 0;
            return best;
        }
    }
    // ---- general row-by-row DP.  Lane l owns columns wb + 16*l .. wb + 16*l + 15 (rank == lane); when
    // the first live column crosses a strip boundary the whole state moves down by that many lanes.
    int32_t Cs[WSTRIP], Ds[WSTRIP];
    uint32_t Cm[WSTRIP], Cx[WSTRIP], Dm[WSTRIP], Dx[WSTRIP];
    uint32_t wb = 0, jb = lane * WSTRIP;
    uint32_t qlo, qhi, qn;
    load_qbits<WSTRIP>(Q, aq, dir, jb, lenB, qlo, qhi, qn);
    bool over = false;
#pragma unroll
    for (int s = 0; s < WSTRIP; s++) {
        uint32_t j = jb + s;
        int32_t v = j ? -O - (int32_t)j * E : 0;
        bool alive = j <= lenB && (j == 0 || v >= -Y);
        Cs[s] = alive ? v : NEG; Cm[s] = 0; Cx[s] = 0;
        Ds[s] = NEG; Dm[s] = 0; Dx[s] = 0;
        if (alive && j >= WINDOW - WSTRIP) over = true;
    }
    if (__ballot(over)) { best.overflow = 1; return best; }
    RowBases rbase{0, 0, 0}, rnext = load_row_bases(T, at, dir, 1u);
    for (uint32_t i = 1; i <= lenA; i++) {
        const int32_t thr = best.score - Y;
        const uint32_t rbit = (i - 1u) & 31u;
        if (rbit == 0) { rbase = rnext; rnext = load_row_bases(T, at, dir, i + 32u); }
        const uint32_t alo = (rbase.lo >> rbit) & 1u, ahi = (rbase.hi >> rbit) & 1u, an = (rbase.nm >> rbit) & 1u, acg = alo ^ ahi;
        // C of the column left of my strip (previous row): last slot of the previous lane
        const Cell p7 = dpp_cell<0x138, 0xf>(Cell{Cs[WSTRIP - 1], Cm[WSTRIP - 1], Cx[WSTRIP - 1]});
        // pass 1 (slots descending, in place): D(i,j) and H(i,j) = max(diagonal, D) overwrite the
        // previous row's D and C; slot s still sees the old C of slot s-1
#pragma unroll
        for (int s = WSTRIP - 1; s >= 0; s--) {
            const uint32_t j = jb + s;
            const bool exists = j <= lenB;
            Cell dd{NEG, 0, 0}, g{NEG, 0, 0};
            if (Ds[s] > NEGH) { dd.s = Ds[s] - E; dd.nm = Dm[s]; dd.nx = Dx[s]; }
            if (Cs[s] > NEGH && Cs[s] - O - E > dd.s) { dd.s = Cs[s] - O - E; dd.nm = Cm[s]; dd.nx = Cx[s]; }
            Cell pc = s ? Cell{Cs[s ? s - 1 : 0], Cm[s ? s - 1 : 0], Cx[s ? s - 1 : 0]} : p7;
            if (pc.s > NEGH && j >= 1) {
                uint32_t dl = alo ^ ((qlo >> s) & 1u), dh = ahi ^ ((qhi >> s) & 1u), nn = an | ((qn >> s) & 1u);
                bool m = !(dl | dh | nn);
                g.s = pc.s + sub_score(dl, dh, acg, nn);
                g.nm = pc.nm + (m ? 1u : 0u);
                g.nx = pc.nx + (m ? 0u : 1u);
            }
            if (!exists) { dd.s = NEG; g.s = NEG; }
            Ds[s] = dd.s; Dm[s] = dd.nm; Dx[s] = dd.nx;
            Cell hh = g;  // diagonal preferred on ties
            if (dd.s > g.s) hh = dd;
            Cs[s] = hh.s; Cm[s] = hh.nm; Cx[s] = hh.nx;
        }
        // pass 2: insertion state = exclusive max-plus scan of u_k = H_k + (k - wb) * E along the
        // row: lane aggregate, then one cross-lane scan
        Cell run{NEG, 0, 0};
#pragma unroll
        for (int s = 0; s < WSTRIP; s++) {
            Cell u{Cs[s] > NEGH ? Cs[s] + (int32_t)(lane * WSTRIP + s) * E : NEG, Cm[s], Cx[s]};
            run = cmax_left(run, u);
        }
        Cell acc = dpp_cell<0x138, 0xf>(wave_incl_maxscan(run));  // best u of every column left of my strip
        // pass 3: C = max(H, I), prune, row statistics
        uint32_t amask = 0;
        Best4 rb{NEG, 0xFFFFFFFFu, 0, 0};
#pragma unroll
        for (int s = 0; s < WSTRIP; s++) {
            Cell hh{Cs[s], Cm[s], Cx[s]};
            Cell I{NEG, acc.nm, acc.nx};
            if (acc.s > NEGH) I.s = acc.s - O - (int32_t)(lane * WSTRIP + s) * E;
            Cell u{hh.s > NEGH ? hh.s + (int32_t)(lane * WSTRIP + s) * E : NEG, hh.nm, hh.nx};
            acc = cmax_left(acc, u);
            Cell c = hh;  // H preferred over I on ties
            if (I.s > c.s) c = I;
            const bool alive = (jb + s <= lenB) && c.s >= thr && c.s > NEGH;
            Cs[s] = alive ? c.s : NEG; Cm[s] = c.nm; Cx[s] = c.nx;
            if (!alive) Ds[s] = NEG;
            if (alive) {
                amask |= 1u << s;
                if (c.s > rb.s) { rb.s = c.s; rb.j = jb + s; rb.nm = c.nm; rb.nx = c.nx; }
            }
        }
        const uint64_t ball = __ballot(amask != 0);
        if (!ball) break;
        const uint32_t rf = (uint32_t)__builtin_ctzll(ball), rl = 63u - (uint32_t)__builtin_clzll(ball);
        if (rl == 63u) { best.overflow = 1; break; }
        best.maxcols = max(best.maxcols, (rl + 1u) * WSTRIP);
        best.rows = i;
        // best cell of the row (only when some lane beats the best of the rows above)
        if (__ballot(rb.s > best.score)) {
            const Best4 t = wave_best(rb);
            if (t.s > best.score) { best.score = t.s; best.i = i; best.j = t.j; best.nm = t.nm; best.nx = t.nx; }
            // 32-bit cells and no limit on the rows here: a half extension that nears 2^31 (20 Mbp of near-identity without a
            // break) goes on to k6_dp_any, which rebases its cells
            if (best.score > cap) { best.overflow = 1; break; }
        }
        // slide the window so that it starts at the strip holding the first live column
        const uint32_t fmask = (uint32_t)__builtin_amdgcn_readlane((int)amask, (int)rf);
        const uint32_t plo = wb + rf * WSTRIP + (uint32_t)__builtin_ctz(fmask);
        const uint32_t nwb = plo & ~(uint32_t)(WSTRIP - 1);
        if (nwb != wb) {
            const uint32_t shift = (nwb - wb) >> WSHIFT;  // == rf
            wb = nwb;
            jb = wb + lane * WSTRIP;
            const int src = (int)((lane + shift) & 63u);
            const bool fresh = lane + shift >= 64u;  // strip re-enters on the right with new columns
#pragma unroll
            for (int s = 0; s < WSTRIP; s++) {
                int32_t cs = __shfl(Cs[s], src), ds = __shfl(Ds[s], src);
                Cm[s] = __shfl(Cm[s], src); Cx[s] = __shfl(Cx[s], src);
                Dm[s] = __shfl(Dm[s], src); Dx[s] = __shfl(Dx[s], src);
                Cs[s] = fresh ? NEG : cs;
                Ds[s] = fresh ? NEG : ds;
            }
            uint32_t a0 = __shfl(qlo, src), a1 = __shfl(qhi, src), a2 = __shfl(qn, src);
            if (fresh) load_qbits<WSTRIP>(Q, aq, dir, jb, lenB, qlo, qhi, qn);
            else { qlo = a0; qhi = a1; qn = a2; }
        }
    }
    return best;
}

// ---- four-wavefront variant of the same DP -----------------------------------------------------
// One workgroup of four wavefronts (one per SIMD) shares a half extension: wavefront of rank r owns
// the 256 columns wb + 256 r ..., four per lane.  The row costs each wavefront a quarter of the
// VALU work of the single-wavefront kernel plus two workgroup barriers, so a half extension finishes
// ~3-4x sooner — which is what matters when a K6 round has fewer jobs than the chip has SIMDs
// (multi-GPU shards, last rounds).  The window slides in whole 256-column blocks (the ring of four
// wavefronts rotates, no state moves), so only 769 columns are guaranteed; a band that does not fit
// is redone by the single-wavefront kernels.
constexpr int C4_WS = 4, C4_WCOLS = 64 * C4_WS, C4_THREADS = 256;

// k6_dp4 carries (matches, mismatches) packed into one word, 16 bits each: a third less to select, scan and
// exchange per cell.  Counts grow by one per row at most, so rows < 65535 cannot overflow them; longer half
// extensions fall back to the single-wavefront kernels (unpacked).
struct PCell {
    int32_t s;
    uint32_t c;  // matches | mismatches << 16
};
struct PBest {
    int32_t s;
    uint32_t j, c;
};
__device__ __forceinline__ PCell pcmax_left(const PCell &l, const PCell &r) { return r.s > l.s ? r : l; }
template <int CTRL, int RMASK>
__device__ __forceinline__ PCell dpp_pcell(const PCell &c) {
    PCell o;
    o.s = __builtin_amdgcn_update_dpp(NEG, c.s, CTRL, RMASK, 0xf, false);
    o.c = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c.c, CTRL, RMASK, 0xf, false);
    return o;
}
__device__ __forceinline__ PCell wave_incl_maxscan_p(PCell v) {
    v = pcmax_left(dpp_pcell<0x111, 0xf>(v), v);
    v = pcmax_left(dpp_pcell<0x112, 0xf>(v), v);
    v = pcmax_left(dpp_pcell<0x114, 0xf>(v), v);
    v = pcmax_left(dpp_pcell<0x118, 0xf>(v), v);
    v = pcmax_left(dpp_pcell<0x142, 0xa>(v), v);
    v = pcmax_left(dpp_pcell<0x143, 0xc>(v), v);
    return v;
}
template <int CTRL, int RMASK>
__device__ __forceinline__ PBest dpp_pbest(const PBest &c) {
    PBest o;
    o.s = __builtin_amdgcn_update_dpp(NEG, c.s, CTRL, RMASK, 0xf, false);
    o.j = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c.j, CTRL, RMASK, 0xf, false);
    o.c = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c.c, CTRL, RMASK, 0xf, false);
    return o;
}
__device__ __forceinline__ PBest pbmax_left(const PBest &l, const PBest &r) { return r.s > l.s ? r : l; }
__device__ __forceinline__ PBest wave_best_p(PBest v) {
    v = pbmax_left(dpp_pbest<0x111, 0xf>(v), v);
    v = pbmax_left(dpp_pbest<0x112, 0xf>(v), v);
    v = pbmax_left(dpp_pbest<0x114, 0xf>(v), v);
    v = pbmax_left(dpp_pbest<0x118, 0xf>(v), v);
    v = pbmax_left(dpp_pbest<0x142, 0xa>(v), v);
    v = pbmax_left(dpp_pbest<0x143, 0xc>(v), v);
    PBest t;
    t.s = __builtin_amdgcn_readlane(v.s, 63); t.j = (uint32_t)__builtin_amdgcn_readlane((int)v.j, 63);
    t.c = (uint32_t)__builtin_amdgcn_readlane((int)v.c, 63);
    return t;
}

struct C4Shared {
    PCell bnd[2][4];  // [row parity][wavefront]: C of the wavefront's last column
    PCell tot[4];     // per-wavefront maximum of u
    uint32_t first[4], last[4];
    PBest best[4];
    int flag;
};

__device__ HalfResult block_half_extend(C4Shared &sh, const StrandView &T, const StrandView &Q, uint32_t at, uint32_t aq,
                                        int dir, int32_t O, int32_t E, int32_t Y) {
    constexpr int WS = C4_WS;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t lenA = dir > 0 ? T.len - at : at, lenB = dir > 0 ? Q.len - aq : aq;
    HalfResult best{0, 0, 0, 0, 0, 0, 0, 0};
    // ---- exact shortcut: identical, N-free to the end of the shorter sequence
    {
        const uint32_t n = min(lenA, lenB);
        const int32_t st = dir > 0 ? (int32_t)at : (int32_t)(at - n), sq = dir > 0 ? (int32_t)aq : (int32_t)(aq - n);
        bool ok = true;
        uint32_t ncg = 0;
        for (uint32_t k0 = 0; k0 < n; k0 += C4_THREADS * 32u) {
            uint32_t k = k0 + tid * 32u;
            if (k < n) {
                const Win32 tw = win32(T, st + (int32_t)k), qw = win32(Q, sq + (int32_t)k);
                uint32_t bad = (tw.lo ^ qw.lo) | (tw.hi ^ qw.hi) | tw.nm | qw.nm;
                uint32_t rem = n - k, mask = rem < 32 ? (1u << rem) - 1u : 0xFFFFFFFFu;
                if (bad & mask) ok = false;
                ncg += __popc((tw.lo ^ tw.hi) & mask);
            }
            if (__syncthreads_or(!ok)) { ok = false; break; }
        }
        if (!__syncthreads_or(!ok)) {
            // block sum of ncg through the shared scratch (64-bit total)
            unsigned long long tsum = ncg;
            for (int o = 32; o > 0; o >>= 1) tsum += __shfl_xor(tsum, o);
            if (lane == 0) { sh.first[wave] = (uint32_t)tsum; sh.last[wave] = (uint32_t)(tsum >> 32); }
            __syncthreads();
            unsigned long long all = 0;
            for (int w = 0; w < 4; w++) all += ((unsigned long long)sh.last[w] << 32) | sh.first[w];
            __syncthreads();
            unsigned long long sc = 100ull * all + 91ull * ((unsigned long long)n - all);
            // rows == 0 marks a shortcut result: its score is 64 bits wide, high word in maxcols (k6_resolve)
            best.score = (int32_t)(uint32_t)sc; best.maxcols = (uint32_t)(sc >> 32); best.i = n; best.j = n; best.nm = n; best.nx = 0;
            return best;
        }
    }
    // ---- row-by-row DP
    int32_t Cs[WS], Ds[WS];
    uint32_t Cc[WS], Dc[WS];  // matches | mismatches << 16 (PCell)
    uint32_t wb = 0, wbase = 0;                      // window base column (multiple of 256), wavefront holding it
    uint32_t rw = wave, jb = rw * C4_WCOLS + lane * WS;  // my rank in the ring, my first column
    uint32_t qlo, qhi, qn;
    load_qbits<WS>(Q, aq, dir, jb, lenB, qlo, qhi, qn);
#pragma unroll
    for (int s = 0; s < WS; s++) {
        uint32_t j = jb + s;
        int32_t v = j ? -O - (int32_t)j * E : 0;
        bool alive = j <= lenB && (j == 0 || v >= -Y);
        Cs[s] = alive ? v : NEG; Cc[s] = 0;
        Ds[s] = NEG; Dc[s] = 0;
    }
    {   // row 0 must fit the guaranteed part of the window
        uint32_t hi0 = 0;
        if (Y >= O + E) hi0 = min(lenB, (uint32_t)((Y - O) / E));
        if (hi0 >= 4 * C4_WCOLS - WS) { best.overflow = 1; return best; }
    }
    if (lane == 63) sh.bnd[0][wave] = PCell{Cs[WS - 1], 0};
    __syncthreads();
    uint32_t par = 0;
    RowBases rbase{0, 0, 0}, rnext = load_row_bases(T, at, dir, 1u);
    for (uint32_t i = 1; i <= lenA; i++, par ^= 1u) {
        // the packed counts hold 16 bits each: a longer extension is redone by the single-wavefront kernel
        if (i >= 0xFFFFu) { best.overflow = 1; break; }
        const int32_t thr = best.score - Y;
        const uint32_t rbit = (i - 1u) & 31u;
        if (rbit == 0) { rbase = rnext; rnext = load_row_bases(T, at, dir, i + 32u); }
        const uint32_t alo = (rbase.lo >> rbit) & 1u, ahi = (rbase.hi >> rbit) & 1u, an = (rbase.nm >> rbit) & 1u, acg = alo ^ ahi;
        // C of the column left of my strip (previous row)
        PCell p7 = dpp_pcell<0x138, 0xf>(PCell{Cs[WS - 1], Cc[WS - 1]});
        if (lane == 0) p7 = rw ? sh.bnd[par][(wave + 3u) & 3u] : PCell{NEG, 0};
#pragma unroll
        for (int s = WS - 1; s >= 0; s--) {
            const uint32_t j = jb + s;
            const bool exists = j <= lenB;
            PCell dd{NEG, 0}, g{NEG, 0};
            if (Ds[s] > NEGH) { dd.s = Ds[s] - E; dd.c = Dc[s]; }
            if (Cs[s] > NEGH && Cs[s] - O - E > dd.s) { dd.s = Cs[s] - O - E; dd.c = Cc[s]; }
            PCell pc = s ? PCell{Cs[s ? s - 1 : 0], Cc[s ? s - 1 : 0]} : p7;
            if (pc.s > NEGH && j >= 1) {
                uint32_t dl = alo ^ ((qlo >> s) & 1u), dh = ahi ^ ((qhi >> s) & 1u), nn = an | ((qn >> s) & 1u);
                bool m = !(dl | dh | nn);
                g.s = pc.s + sub_score(dl, dh, acg, nn);
                g.c = pc.c + (m ? 1u : 0x10000u);
            }
            if (!exists) { dd.s = NEG; g.s = NEG; }
            Ds[s] = dd.s; Dc[s] = dd.c;
            PCell hh = g;
            if (dd.s > g.s) hh = dd;
            Cs[s] = hh.s; Cc[s] = hh.c;
        }
        // insertion state: u_k = H_k + (k - wb) * E; in-lane, in-wavefront (DPP), across wavefronts (LDS)
        const int32_t koff = (int32_t)(rw * C4_WCOLS + lane * WS);
        PCell run{NEG, 0};
#pragma unroll
        for (int s = 0; s < WS; s++) {
            PCell u{Cs[s] > NEGH ? Cs[s] + (koff + s) * E : NEG, Cc[s]};
            run = pcmax_left(run, u);
        }
        const PCell inc = wave_incl_maxscan_p(run);
        if (lane == 63) sh.tot[wave] = inc;
        __syncthreads();
        PCell acc{NEG, 0};
        for (uint32_t r = 0; r < rw; r++) acc = pcmax_left(acc, sh.tot[(wbase + r) & 3u]);
        acc = pcmax_left(acc, dpp_pcell<0x138, 0xf>(inc));
        uint32_t amask = 0;
        PBest rb{NEG, 0xFFFFFFFFu, 0};
#pragma unroll
        for (int s = 0; s < WS; s++) {
            PCell hh{Cs[s], Cc[s]};
            PCell I{NEG, acc.c};
            if (acc.s > NEGH) I.s = acc.s - O - (koff + s) * E;
            PCell u{hh.s > NEGH ? hh.s + (koff + s) * E : NEG, hh.c};
            acc = pcmax_left(acc, u);
            PCell c = hh;
            if (I.s > c.s) c = I;
            const bool alive = (jb + s <= lenB) && c.s >= thr && c.s > NEGH;
            Cs[s] = alive ? c.s : NEG; Cc[s] = c.c;
            if (!alive) Ds[s] = NEG;
            if (alive) {
                amask |= 1u << s;
                if (c.s > rb.s) { rb.s = c.s; rb.j = jb + s; rb.c = c.c; }
            }
        }
        // publish: boundary cell for the next row, first / last live column, best cell
        if (lane == 63) sh.bnd[par ^ 1u][wave] = PCell{Cs[WS - 1], Cc[WS - 1]};
        const uint64_t ball = __ballot(amask != 0);
        uint32_t wfirst = 0xFFFFFFFFu, wlast = 0;
        if (ball) {
            const uint32_t lf = (uint32_t)__builtin_ctzll(ball), ll = 63u - (uint32_t)__builtin_clzll(ball);
            const uint32_t mf = (uint32_t)__builtin_amdgcn_readlane((int)amask, (int)lf);
            const uint32_t ml = (uint32_t)__builtin_amdgcn_readlane((int)amask, (int)ll);
            const uint32_t cb = wb + rw * C4_WCOLS;
            wfirst = cb + lf * WS + (uint32_t)__builtin_ctz(mf);
            wlast = cb + ll * WS + (31u - (uint32_t)__builtin_clz(ml));
        }
        PBest wbest{NEG, 0xFFFFFFFFu, 0};
        if (__ballot(rb.s > best.score)) wbest = wave_best_p(rb);
        if (lane == 0) { sh.first[wave] = wfirst; sh.last[wave] = wlast; sh.best[wave] = wbest; }
        __syncthreads();
        uint32_t first = 0xFFFFFFFFu, last = 0;
        PBest tb{NEG, 0xFFFFFFFFu, 0};
#pragma unroll
        for (int w = 0; w < 4; w++) {
            first = min(first, sh.first[w]);
            last = max(last, sh.last[w]);
            const PBest o = sh.best[w];
            if (o.s > tb.s || (o.s == tb.s && o.j < tb.j)) tb = o;
        }
        if (first == 0xFFFFFFFFu) break;
        if (last - wb >= 4 * C4_WCOLS - WS) { best.overflow = 1; break; }
        best.maxcols = max(best.maxcols, last - wb + 1);
        best.rows = i;
        if (tb.s > best.score) { best.score = tb.s; best.i = i; best.j = tb.j; best.nm = tb.c & 0xFFFFu; best.nx = tb.c >> 16; }
        // slide the window by whole 256-column blocks: the ring of wavefronts rotates
        const uint32_t k = (first - wb) / C4_WCOLS;
        if (k) {
            const bool fresh = rw < k;  // my block left the window: I re-enter on the right with new columns
            wb += k * C4_WCOLS;
            wbase = (wbase + k) & 3u;
            rw = (wave - wbase) & 3u;
            jb = wb + rw * C4_WCOLS + lane * WS;
            if (fresh) {
#pragma unroll
                for (int s = 0; s < WS; s++) { Cs[s] = NEG; Ds[s] = NEG; }
                load_qbits<WS>(Q, aq, dir, jb, lenB, qlo, qhi, qn);
                if (lane == 63) sh.bnd[par ^ 1u][wave] = PCell{NEG, 0};
            }
            __syncthreads();
        }
    }
    return best;
}

// ---- lean single-wavefront DP (k6_dp1): the production kernel -----------------------------------------------
// Same recurrences, pruning and tie-breaks as wave_half_extend<16> (one wavefront, 16 columns per lane, 1024-column
// window that slides by whole strips), written for VALU issue, which is what bounds K6 (profiles/r02_*: both
// older kernels spend 1700+ issue slots per DP row):
//   * counts packed (matches | mismatches << 16): one select instead of two;
//   * no liveness guards: a dead cell is any value below NEGH, arithmetic on it stays below NEGH for the one row
//     until pruning resets it to NEG, so max / compare need no special cases;
//   * the substitution score of a cell is ONE v_perm_b32: the row's target base is wave-uniform, so the four
//     possible scores (+128, as bytes) sit in a scalar register and the column's query base is a precomputed byte
//     selector (selector 4 = the constant 28 = -100 + 128 of an N column; an N row is the table 0x1C1C1C1C);
//   * the insertion state is carried in the frame of the current column (acc = max(acc, H) - E) instead of
//     u_k = H_k + k E: no per-column constants; lanes are stitched with one max-scan of (aggregate + lane * 16 E);
//   * per cell the row maximum is one v_max; which cell it was (smallest column on ties) is found with scalar
//     reads only in rows that improve the best score; liveness is per strip (row maximum above NEGH), which is
//     all the window slide and the overflow test ever needed;
//   * columns beyond the end of the query only exist when the window touches it: rows of such windows run the
//     EDGE variant (one extra mask test per cell), selected wave-uniformly.
// Measured: ~40 VALU instructions per cell.  Strips of 14 columns (896-column window): the widest live band of a
// default-parameter extension is ~600 + 2 strips (y-drop 9400 / gap extend 30 on either side of the best cell; C4: all
// below 768), and a band that does not fit is redone by the 2048-column kernel.
constexpr int L_WS = 14, L_WINDOW = 64 * L_WS;
struct LeanState {
    int32_t C[L_WS], D[L_WS];
    uint32_t Cc[L_WS], Dc[L_WS], sel[L_WS];
};

__device__ __forceinline__ int32_t wave_max_i32(int32_t v) {
    v = max(v, __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}

// inclusive max-scan, ties to the lower lane; a lane without a source sees its own value (no identity moves)
template <int CTRL, int RMASK>
__device__ __forceinline__ PCell lean_scan_step(const PCell &v) {
    PCell o;
    o.s = __builtin_amdgcn_update_dpp(v.s, v.s, CTRL, RMASK, 0xf, false);
    o.c = (uint32_t)__builtin_amdgcn_update_dpp((int)v.c, (int)v.c, CTRL, RMASK, 0xf, false);
    return pcmax_left(o, v);
}
__device__ __forceinline__ PCell lean_incl_maxscan(PCell v) {
    v = lean_scan_step<0x111, 0xf>(v);
    v = lean_scan_step<0x112, 0xf>(v);
    v = lean_scan_step<0x114, 0xf>(v);
    v = lean_scan_step<0x118, 0xf>(v);
    v = lean_scan_step<0x142, 0xa>(v);
    v = lean_scan_step<0x143, 0xc>(v);
    return v;
}

// byte selectors of a strip from its query bits: 0..3 = base code (lo | hi << 1), 4 = N; upper bytes select zero
__device__ __forceinline__ void lean_selectors(LeanState &S, uint32_t qlo, uint32_t qhi, uint32_t qn) {
#pragma unroll
    for (int s = 0; s < L_WS; s++) {
        const uint32_t idx = ((qlo >> s) & 1u) | (((qhi >> s) & 1u) << 1);
        S.sel[s] = 0x0C0C0C00u | (((qn >> s) & 1u) ? 4u : idx);
    }
}

// one DP row; returns the lane's row maximum (NEG when none of its cells is live)
template <bool EDGE>
__device__ __forceinline__ int32_t lean_row(LeanState &S, uint32_t srow, int32_t O, int32_t E, int32_t thr, uint32_t exmask,
                                            int32_t lane_base, int32_t kneg128) {
    const int32_t OE = O + E;
    // C of the column left of my strip (previous row): last slot of the previous lane
    PCell pc = dpp_pcell<0x138, 0xf>(PCell{S.C[L_WS - 1], S.Cc[L_WS - 1]});
    // pass 1 (slots descending, in place): D and H = max(diagonal, D); slot s still sees the old C of slot s - 1
#pragma unroll
    for (int s = L_WS - 1; s >= 0; s--) {
        const int32_t t1 = S.D[s] - E, t2 = S.C[s] - OE;
        const bool open = t2 > t1;
        const int32_t ds = max(t1, t2);
        const uint32_t dc = open ? S.Cc[s] : S.Dc[s];
        const PCell left = s ? PCell{S.C[s ? s - 1 : 0], S.Cc[s ? s - 1 : 0]} : pc;
        const uint32_t scb = __builtin_amdgcn_perm(28u, srow, S.sel[s]);   // score + 128
        const int32_t gs = left.s + (int32_t)scb + kneg128;
        const uint32_t gc = left.c + 0x10000u + ((int32_t)scb > 128 ? 1u : 0u);   // diagonal steps << 16 | matches
        const bool vert = ds > gs;  // diagonal preferred on ties
        S.D[s] = ds; S.Dc[s] = dc;
        S.C[s] = max(gs, ds);
        S.Cc[s] = vert ? dc : gc;
    }
    // pass 2: the strip's aggregate of the insertion state as it arrives at the first column of the next strip
    PCell run{NEG, 0};
#pragma unroll
    for (int s = 0; s < L_WS; s++) {
        const bool take = S.C[s] > run.s;  // ties -> left
        run.c = take ? S.Cc[s] : run.c;
        run.s = max(run.s, S.C[s]) - E;
    }
    run.s += lane_base;  // common frame: column 0 of the window
    PCell acc = dpp_pcell<0x138, 0xf>(lean_incl_maxscan(run));  // best of every column left of my strip
    acc.s += L_WS * E - lane_base;                                   // ... as it arrives at my first column
    // pass 3: C = max(H, I), prune, row maximum
    int32_t rowmax = NEG;
#pragma unroll
    for (int s = 0; s < L_WS; s++) {
        const int32_t hs = S.C[s];
        const uint32_t hc = S.Cc[s];
        const int32_t is = acc.s - O;
        const bool ins = is > hs;       // H preferred over I on ties
        const int32_t cs = max(hs, is);
        const uint32_t cc = ins ? acc.c : hc;
        const bool take = hs > acc.s;   // ties -> left
        acc.c = take ? hc : acc.c;
        acc.s = max(acc.s, hs) - E;
        bool alive = cs >= thr;
        if (EDGE) alive = alive && ((exmask >> s) & 1u);
        S.C[s] = alive ? cs : NEG; S.Cc[s] = cc;
        S.D[s] = alive ? S.D[s] : NEG;
        rowmax = max(rowmax, S.C[s]);
    }
    return rowmax;
}

__device__ HalfResult wave_half_extend_lean(const StrandView &T, const StrandView &Q, uint32_t at, uint32_t aq, int dir,
                                            int32_t O, int32_t E, int32_t Y) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t lenA = dir > 0 ? T.len - at : at, lenB = dir > 0 ? Q.len - aq : aq;
    HalfResult best{0, 0, 0, 0, 0, 0, 0, 0};
    // ---- exact shortcut: identical, N-free to the end of the shorter sequence (as wave_half_extend)
    {
        const uint32_t n = min(lenA, lenB);
        const int32_t st = dir > 0 ? (int32_t)at : (int32_t)(at - n), sq = dir > 0 ? (int32_t)aq : (int32_t)(aq - n);
        bool ok = true;
        uint64_t ncg = 0;
        for (uint32_t k0 = 0; k0 < n; k0 += 64u * 32u) {
            uint32_t k = k0 + lane * 32u;
            if (k < n) {
                const Win32 tw = win32(T, st + (int32_t)k), qw = win32(Q, sq + (int32_t)k);
                uint32_t bad = (tw.lo ^ qw.lo) | (tw.hi ^ qw.hi) | tw.nm | qw.nm;
                uint32_t rem = n - k, mask = rem < 32 ? (1u << rem) - 1u : 0xFFFFFFFFu;
                if (bad & mask) ok = false;
                ncg += __popc((tw.lo ^ tw.hi) & mask);
            }
            if (__ballot(!ok)) break;
        }
        if (!__ballot(!ok)) {
            for (int o = 32; o > 0; o >>= 1) ncg += __shfl_xor(ncg, o);
            uint64_t sc = 100ull * ncg + 91ull * ((uint64_t)n - ncg);
            best.score = (int32_t)(uint32_t)sc; best.maxcols = (uint32_t)(sc >> 32); best.i = n; best.j = n; best.nm = n; best.nx = 0;
            return best;
        }
    }
    // biased score bytes of the four query bases for each target base (index lo | hi << 1)
    uint32_t tab[4];
#pragma unroll
    for (uint32_t a = 0; a < 4; a++) {
        uint32_t w = 0;
#pragma unroll
        for (uint32_t b = 0; b < 4; b++) {
            const uint32_t alo = a & 1u, ahi = a >> 1, dl = alo ^ (b & 1u), dh = ahi ^ (b >> 1);
            w |= (uint32_t)(sub_score(dl, dh, alo ^ ahi, 0u) + 128) << (8u * b);
        }
        tab[a] = w;
    }
    LeanState S;
    uint32_t wb = 0, jb = lane * L_WS;
    {
        uint32_t qlo, qhi, qn;
        load_qbits<L_WS>(Q, aq, dir, jb, lenB, qlo, qhi, qn);
        lean_selectors(S, qlo, qhi, qn);
    }
    bool over = false;
#pragma unroll
    for (int s = 0; s < L_WS; s++) {
        const uint32_t j = jb + s;
        const int32_t v = j ? -O - (int32_t)j * E : 0;
        const bool alive = j <= lenB && (j == 0 || v >= -Y);
        S.C[s] = alive ? v : NEG; S.Cc[s] = 0;
        S.D[s] = NEG; S.Dc[s] = 0;
        if (alive && j >= (uint32_t)(L_WINDOW - L_WS)) over = true;
    }
    if (__ballot(over)) { best.overflow = 1; return best; }
    const int32_t lane_base = (int32_t)(lane * L_WS) * E;
    const int32_t kneg128 = __builtin_amdgcn_readfirstlane(-128);
    uint32_t exmask = (1u << L_WS) - 1u;
    bool edge = wb + (uint32_t)L_WINDOW - 1u > lenB;
    if (edge) {
        exmask = 0;
#pragma unroll
        for (int s = 0; s < L_WS; s++) exmask |= (jb + s <= lenB ? 1u : 0u) << s;
    }
    RowBases rbase{0, 0, 0}, rnext = load_row_bases(T, at, dir, 1u);
    for (uint32_t i = 1; i <= lenA; i++) {
        // the packed counts hold 16 bits each: a longer extension is redone by the wide kernel (unpacked counts)
        if (i >= 0xFFFFu) { best.overflow = 1; break; }
        const int32_t thr = best.score - Y;
        const uint32_t rbit = (i - 1u) & 31u;
        if (rbit == 0) { rbase = rnext; rnext = load_row_bases(T, at, dir, i + 32u); }
        const uint32_t rlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)rbase.lo), rhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)rbase.hi),
                       rnm = (uint32_t)__builtin_amdgcn_readfirstlane((int)rbase.nm);
        const uint32_t a = ((rlo >> rbit) & 1u) | (((rhi >> rbit) & 1u) << 1);
        uint32_t srow = a & 2u ? (a & 1u ? tab[3] : tab[2]) : (a & 1u ? tab[1] : tab[0]);
        if ((rnm >> rbit) & 1u) srow = 0x1C1C1C1Cu;
        const int32_t rowmax = edge ? lean_row<true>(S, srow, O, E, thr, exmask, lane_base, kneg128)
                                    : lean_row<false>(S, srow, O, E, thr, exmask, lane_base, kneg128);
        const uint64_t ball = __ballot(rowmax > NEGH);
        if (!ball) break;
        const uint32_t rf = (uint32_t)__builtin_ctzll(ball), rl = 63u - (uint32_t)__builtin_clzll(ball);
        if (rl == 63u) { best.overflow = 1; break; }
        best.maxcols = max(best.maxcols, (rl + 1u) * L_WS);
        best.rows = i;
        const int32_t wmax = wave_max_i32(rowmax);
        if (wmax > best.score) {
            // the cell: lowest lane holding the maximum, smallest slot in it (smallest column on ties)
            const uint32_t L = (uint32_t)__builtin_ctzll(__ballot(rowmax == wmax));
            uint32_t slot = 0, cnt = 0;
#pragma unroll
            for (int s = L_WS - 1; s >= 0; s--) {
                const int32_t v = __builtin_amdgcn_readlane(S.C[s], (int)L);
                if (v == wmax) { slot = (uint32_t)s; cnt = (uint32_t)__builtin_amdgcn_readlane((int)S.Cc[s], (int)L); }
            }
            best.score = wmax; best.i = i; best.j = wb + L * L_WS + slot; best.nm = cnt & 0xFFFFu; best.nx = (cnt >> 16) - (cnt & 0xFFFFu);
        }
        // slide the window so that it starts at the strip holding the first live column
        if (rf) {
            wb += rf * L_WS;
            jb = wb + lane * L_WS;
            const int src = (int)((lane + rf) & 63u);
            const bool fresh = lane + rf >= 64u;  // strip re-enters on the right with new columns
#pragma unroll
            for (int s = 0; s < L_WS; s++) {
                const int32_t cs = __shfl(S.C[s], src), ds = __shfl(S.D[s], src);
                S.Cc[s] = __shfl(S.Cc[s], src); S.Dc[s] = __shfl(S.Dc[s], src);
                S.sel[s] = __shfl(S.sel[s], src);
                S.C[s] = fresh ? NEG : cs;
                S.D[s] = fresh ? NEG : ds;
            }
            if (fresh) {
                uint32_t qlo, qhi, qn;
                load_qbits<L_WS>(Q, aq, dir, jb, lenB, qlo, qhi, qn);
                lean_selectors(S, qlo, qhi, qn);
            }
            edge = wb + (uint32_t)L_WINDOW - 1u > lenB;
            if (edge) {
                exmask = 0;
#pragma unroll
                for (int s = 0; s < L_WS; s++) exmask |= (jb + s <= lenB ? 1u : 0u) << s;
            }
        }
    }
    return best;
}

__global__ __launch_bounds__(64) void k6_dp1(const Group *__restrict__ groups, const DpJob *__restrict__ jobs,
                                             HalfResult *__restrict__ res, int32_t O, int32_t E, int32_t Y) {
    const DpJob job = jobs[blockIdx.x];
    const Group &G = groups[job.group];
    HalfResult r = wave_half_extend_lean(G.T, G.Q, job.at, job.aq, job.dir, O, E, Y);
    if (threadIdx.x == 0) res[job.slot] = r;
}

__global__ __launch_bounds__(C4_THREADS) void k6_dp4(const Group *__restrict__ groups, const DpJob *__restrict__ jobs,
                                                     HalfResult *__restrict__ res, int32_t O, int32_t E, int32_t Y) {
    __shared__ C4Shared sh;
    const DpJob job = jobs[blockIdx.x];
    const Group &G = groups[job.group];
    HalfResult r = block_half_extend(sh, G.T, G.Q, job.at, job.aq, job.dir, O, E, Y);
    if (threadIdx.x == 0) res[job.slot] = r;
}

// Anchor = centre of the best 31-column window of the HSP (first maximum).  Windows are cut into
// chunks of ANCHOR_CHUNK starts; a wave scans one chunk (each lane slides over 64 consecutive
// starts) and folds its best (sum, start) into a packed 64-bit word with atomicMax — high half
// = biased sum, low half = ~start, so the maximum is the highest sum at the smallest start.
constexpr uint32_t ANCHOR_W = 31, ANCHOR_CHUNK = 4096;
constexpr int ANCHOR_THREADS = 1024;

__device__ void wave_anchor_chunk(const StrandView &T, const StrandView &Q, const mimeo_hsp &h, uint32_t chunk,
                                  unsigned long long *packed) {
    const uint32_t lane = threadIdx.x & 63u, nw = h.length - ANCHOR_W + 1;
    const int32_t d = (int32_t)h.tstart - (int32_t)h.qstart;
    const uint32_t w0 = chunk * ANCHOR_CHUNK + lane * (ANCHOR_CHUNK / 64u), w1 = min(nw, w0 + ANCHOR_CHUNK / 64u);
    int32_t bs = INT32_MIN;
    uint32_t bw = 0xFFFFFFFFu;
    if (w0 < w1) {
        int32_t sum = 0;
        bool m;
        for (uint32_t k = 0; k < ANCHOR_W; k++) {
            int32_t pt = (int32_t)(h.tstart + w0 + k);
            sum += pair_score(T, Q, pt, pt - d, &m);
        }
        bs = sum; bw = w0;
        for (uint32_t w = w0 + 1; w < w1; w++) {
            int32_t add = (int32_t)(h.tstart + w + ANCHOR_W - 1), sub = (int32_t)(h.tstart + w - 1);
            sum += pair_score(T, Q, add, add - d, &m) - pair_score(T, Q, sub, sub - d, &m);
            if (sum > bs) { bs = sum; bw = w; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        int32_t ob = __shfl_xor(bs, o);
        uint32_t ow = __shfl_xor(bw, o);
        if (ob > bs || (ob == bs && ow < bw)) { bs = ob; bw = ow; }
    }
    if (lane == 0 && bw != 0xFFFFFFFFu)
        atomicMax(packed, ((unsigned long long)(uint32_t)(bs + 8192) << 32) | (unsigned long long)(0xFFFFFFFFu - bw));
}

// packed[b0 + r] = best window of the r-th chained HSP of the group.  ANCHOR_SPLIT workgroups of 16
// wavefronts share the (HSP, chunk) items of a group round-robin (the 5 Mbp self HSP alone has 1220
// chunks); k6_anchor_final turns the packed maxima into anchor points.
constexpr uint32_t ANCHOR_SPLIT = 16;
__global__ __launch_bounds__(ANCHOR_THREADS) void k6_anchor_points(const Group *__restrict__ groups,
                                                                   const mimeo_hsp *__restrict__ hs,
                                                                   const uint32_t *__restrict__ order,
                                                                   unsigned long long *__restrict__ packed) {
    const Group &G = groups[blockIdx.x];
    const uint64_t b0 = G.hsp_begin;
    const uint32_t nw = gridDim.y * (ANCHOR_THREADS / 64), me = blockIdx.y * (ANCHOR_THREADS / 64) + (threadIdx.x >> 6);
    uint32_t item = 0;
    for (uint32_t r = 0; r < G.nchain; r++) {
        const mimeo_hsp h = hs[b0 + order[b0 + r]];
        if (h.length <= ANCHOR_W) continue;
        const uint32_t nch = (h.length - ANCHOR_W + 1 + ANCHOR_CHUNK - 1) / ANCHOR_CHUNK;
        // my chunks of this HSP: c = first, first + nw, ...
        const uint32_t first = (me + nw - item % nw) % nw;
        for (uint32_t c = first; c < nch; c += nw) wave_anchor_chunk(G.T, G.Q, h, c, &packed[b0 + r]);
        item += nch;
    }
}
__global__ __launch_bounds__(256) void k6_anchor_final(const Group *__restrict__ groups, const mimeo_hsp *__restrict__ hs,
                                                       const uint32_t *__restrict__ order,
                                                       const unsigned long long *__restrict__ packed,
                                                       uint2 *__restrict__ anchors) {
    const Group &G = groups[blockIdx.x];
    const uint64_t b0 = G.hsp_begin;
    for (uint32_t r = threadIdx.x; r < G.nchain; r += blockDim.x) {
        const mimeo_hsp h = hs[b0 + order[b0 + r]];
        uint32_t off = h.length / 2;
        if (h.length > ANCHOR_W) off = (0xFFFFFFFFu - (uint32_t)packed[b0 + r]) + ANCHOR_W / 2;
        anchors[b0 + r] = make_uint2(h.tstart + off, h.qstart + off);
    }
}

__device__ __forceinline__ bool in_boxes(const mimeo_alignment *aln, uint32_t n, uint2 a) {
    // wave-cooperative: is the anchor inside the box of any of the n earlier alignments?
    bool inside = false;
    for (uint32_t e = threadIdx.x & 63u; e < n; e += 64u) {
        const mimeo_alignment &o = aln[e];
        if (a.x >= o.tstart && a.x < o.tend && a.y >= o.qstart && a.y < o.qend) inside = true;
    }
    return __ballot(inside) != 0;
}

// Anchor states.  The skip rule is ordered (an anchor is skipped iff it lies in the box of an ACCEPTED
// anchor of lower rank), so anchors are finalised strictly in rank order by k6_resolve; what k6_pick
// may choose freely is which unfinalised anchors get their DP in this round.  Fragments of one repeat
// copy sit on neighbouring diagonals within a few kb and only the best-ranked one survives, so an
// anchor that is NEAR a lower-ranked anchor whose DP is pending is deferred (twice at most): in the
// next round it is normally inside that anchor's accepted box and never costs a DP.
enum : uint8_t { A_NEW = 0, A_DONE = 1, A_SKIPPED = 2, A_ACCEPTED = 3 };
constexpr uint32_t PICK_SCAN = 512;   // ranks looked at per round, from the first unfinalised one
constexpr uint32_t NEAR_DIAG = 256, NEAR_POS = 8192, MAX_DEFER = 2;

__device__ __forceinline__ bool anchors_near(uint2 a, uint2 b) {
    int32_t da = (int32_t)a.x - (int32_t)a.y, db = (int32_t)b.x - (int32_t)b.y;
    uint32_t dd = (uint32_t)abs(da - db), dp = a.x > b.x ? a.x - b.x : b.x - a.x;
    return dd <= NEAR_DIAG && dp <= NEAR_POS;
}

// one wave per group
__global__ __launch_bounds__(64) void k6_pick(Group *__restrict__ groups, const uint2 *__restrict__ anchors,
                                              const mimeo_alignment *__restrict__ aln, uint32_t bmax,
                                              uint8_t *__restrict__ astate, uint8_t *__restrict__ adefer,
                                              DpJob *__restrict__ jobs, unsigned int *__restrict__ njobs) {
    __shared__ uint32_t sb[MAX_BATCH];  // ranks scheduled in this round
    Group &G = groups[blockIdx.x];
    const uint64_t b0 = G.hsp_begin;
    const uint32_t first = G.next, lane = threadIdx.x;
    uint32_t nb = 0;
    for (uint32_t r = first; r < G.nchain && nb < bmax && r - first < PICK_SCAN; r++) {
        if (astate[b0 + r] != A_NEW) continue;  // wave-uniform; states of earlier rounds only
        const uint2 a = anchors[b0 + r];
        if (in_boxes(aln + b0, G.nacc, a)) {
            if (lane == 0) astate[b0 + r] = A_SKIPPED;
            continue;
        }
        // near an unfinalised lower-ranked anchor whose DP is done (earlier rounds) or scheduled (this round)?
        bool susp = false;
        for (uint32_t r2 = first + lane; r2 < r; r2 += 64)
            if (astate[b0 + r2] == A_DONE && anchors_near(a, anchors[b0 + r2])) susp = true;
        if (lane < nb && anchors_near(a, anchors[b0 + sb[lane]])) susp = true;
        if (__ballot(susp) && adefer[b0 + r] < MAX_DEFER) {
            if (lane == 0) adefer[b0 + r]++;
            continue;
        }
        if (lane == 0) sb[nb] = r;
        nb++;
        __syncthreads();
    }
    if (lane == 0) {
        G.nbatch = nb;
        unsigned int j0 = nb ? atomicAdd(njobs, 2u * nb) : 0u;
        G.job0 = j0;
        for (uint32_t k = 0; k < nb; k++) {
            const uint32_t r = sb[k];
            astate[b0 + r] = A_DONE;
            const uint2 a = anchors[b0 + r];
            const uint32_t slot = 2u * (uint32_t)(b0 + r);
            jobs[j0 + 2 * k] = DpJob{blockIdx.x, a.x, a.y, -1, slot, 0};
            jobs[j0 + 2 * k + 1] = DpJob{blockIdx.x, a.x, a.y, +1, slot + 1, 0};
        }
    }
}

__global__ __launch_bounds__(64) void k6_dp(const Group *__restrict__ groups, const DpJob *__restrict__ jobs,
                                            HalfResult *__restrict__ res, int32_t O, int32_t E, int32_t Y,
                                            int only_overflowed, int32_t cap) {
    const DpJob job = jobs[blockIdx.x];
    if (only_overflowed && !res[job.slot].overflow) return;
    const Group &G = groups[job.group];
    HalfResult r = wave_half_extend<16>(G.T, G.Q, job.at, job.aq, job.dir, O, E, Y, cap);
    if (threadIdx.x == 0) res[job.slot] = r;
}

// second chance for half extensions whose band outgrew the 1024-column window: 2048 columns
__global__ __launch_bounds__(64) void k6_dp_wide(const Group *__restrict__ groups, const DpJob *__restrict__ jobs,
                                                 HalfResult *__restrict__ res, int32_t O, int32_t E, int32_t Y,
                                                 unsigned int *__restrict__ novf, unsigned int *__restrict__ ovf_list, int32_t cap) {
    const DpJob job = jobs[blockIdx.x];
    if (!res[job.slot].overflow) return;
    const Group &G = groups[job.group];
    HalfResult r = wave_half_extend<32>(G.T, G.Q, job.at, job.aq, job.dir, O, E, Y, cap);
    if (threadIdx.x == 0) {
        res[job.slot] = r;
        if (r.overflow) ovf_list[atomicAdd(novf, 1u)] = blockIdx.x;  // band beyond 2048 columns: k6_dp_any
    }
}

// ---- last resort: a half extension whose band does not fit 2048 columns (tandem arrays: every shift by a
// period scores almost as well, so the live band grows with the array).  One workgroup of 1024 threads, the DP
// rows in global memory as a ring of ANY_COLS columns (two rows: previous / current), three passes per row
// with the same rules and tie-breaks as wave_half_extend.  Slow (a few microseconds per row plus ~1 ns per
// live cell) but exact; only jobs that overflowed the register kernels come here.
constexpr uint32_t ANY_COLS = 1u << 16;   // live band + one row's growth must stay below this
constexpr int ANY_THREADS = 1024;
struct AnyRow {  // one DP row in global memory, indexed by column & (ANY_COLS - 1)
    int32_t *cs, *ds;
    uint32_t *cm, *cx, *dm, *dx;
};
__device__ __forceinline__ AnyRow any_row(uint32_t *base, uint32_t parity) {
    uint32_t *b = base + (size_t)parity * 6u * ANY_COLS;
    return AnyRow{(int32_t *)b, (int32_t *)(b + ANY_COLS), b + 2u * ANY_COLS, b + 3u * ANY_COLS, b + 4u * ANY_COLS, b + 5u * ANY_COLS};
}
constexpr size_t ANY_SLOT_WORDS = 2u * 6u * (size_t)ANY_COLS;  // per job

__global__ __launch_bounds__(ANY_THREADS) void k6_dp_any(const Group *__restrict__ groups, const DpJob *__restrict__ jobs,
                                                         const unsigned int *__restrict__ list, uint32_t first,
                                                         HalfResult *__restrict__ res, uint32_t *__restrict__ scratch,
                                                         int32_t O, int32_t E, int32_t Y, int32_t cap) {
    __shared__ Cell s_scan[ANY_THREADS / 64];
    __shared__ Best4 s_best[ANY_THREADS / 64];
    __shared__ uint32_t s_first[ANY_THREADS / 64], s_last[ANY_THREADS / 64];
    const DpJob job = jobs[list[first + blockIdx.x]];
    const Group &G = groups[job.group];
    const StrandView &T = G.T, &Q = G.Q;
    const uint32_t at = job.at, aq = job.aq;
    const int dir = job.dir;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t lenA = dir > 0 ? T.len - at : at, lenB = dir > 0 ? Q.len - aq : aq;
    uint32_t *base = scratch + (size_t)blockIdx.x * ANY_SLOT_WORDS;
    const uint32_t M = ANY_COLS - 1u;
    HalfResult best{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long sbase = 0;   // the cells are 32 bits wide and stand above this (below)
    // row 0: C(0,j) = -O - j*E while that is within the y-drop
    uint32_t lo = 0, hi = 0;
    if (Y >= O + E) hi = min(lenB, (uint32_t)((Y - O) / E));
    const uint32_t ext = (uint32_t)((Y + 200) / E) + 2u;  // how far an insertion can carry a live cell to the right
    bool overflow = min(lenB, hi + ext) + 2u >= ANY_COLS;   // no row can hold more live columns than the query has left
    {
        AnyRow r0 = any_row(base, 0);
        for (uint32_t j = tid; j <= hi && !overflow; j += ANY_THREADS) {
            r0.cs[j & M] = j ? -O - (int32_t)j * E : 0; r0.cm[j & M] = 0; r0.cx[j & M] = 0;
            r0.ds[j & M] = NEG; r0.dm[j & M] = 0; r0.dx[j & M] = 0;
        }
    }
    __syncthreads();
    uint32_t par = 0;
    for (uint32_t i = 1; i <= lenA && !overflow; i++, par ^= 1u) {
        const AnyRow P = any_row(base, par), N = any_row(base, par ^ 1u);
        const int32_t thr = best.score - Y;
        const int32_t pa = dir > 0 ? (int32_t)(at + i - 1) : (int32_t)(at - i);
        const Base1 ab = base_at(T, pa);
        const uint32_t alo = ab.lo, ahi = ab.hi, an = ab.nm, acg = alo ^ ahi;
        const uint32_t hx = min(lenB, hi + 1u + ext);  // last column that can be alive in this row
        const uint32_t ncols = hx - lo + 1u;
        if (ncols + 2u >= ANY_COLS) { overflow = true; break; }
        // every thread owns one contiguous chunk of the row's columns (the insertion scan runs left to right)
        const uint32_t chunk = (ncols + ANY_THREADS - 1) / ANY_THREADS;
        const uint32_t j0 = lo + tid * chunk, j1 = min(hx + 1u, j0 + chunk);  // [j0, j1), possibly empty
        // pass 1: D and H = max(diagonal, D) from the previous row (dead outside [lo, hi])
        Cell run{NEG, 0, 0};
        for (uint32_t j = j0; j < j1; j++) {
            const bool in = j <= hi;  // j >= lo holds
            const int32_t cp = in ? P.cs[j & M] : NEG, dp = in ? P.ds[j & M] : NEG;
            Cell dd{NEG, 0, 0}, g{NEG, 0, 0};
            if (dp > NEGH) { dd.s = dp - E; dd.nm = P.dm[j & M]; dd.nx = P.dx[j & M]; }
            if (cp > NEGH && cp - O - E > dd.s) { dd.s = cp - O - E; dd.nm = P.cm[j & M]; dd.nx = P.cx[j & M]; }
            if (j >= 1 && j - 1 >= lo && j - 1 <= hi) {
                const int32_t pc = P.cs[(j - 1) & M];
                if (pc > NEGH) {
                    const Base1 qb = base_at(Q, dir > 0 ? (int32_t)(aq + j - 1) : (int32_t)(aq - j));
                    const uint32_t dl = alo ^ qb.lo, dh = ahi ^ qb.hi, nn = an | qb.nm;
                    const bool m = !(dl | dh | nn);
                    g.s = pc + sub_score(dl, dh, acg, nn);
                    g.nm = P.cm[(j - 1) & M] + (m ? 1u : 0u);
                    g.nx = P.cx[(j - 1) & M] + (m ? 0u : 1u);
                }
            }
            N.ds[j & M] = dd.s; N.dm[j & M] = dd.nm; N.dx[j & M] = dd.nx;
            Cell hh = g;  // diagonal preferred on ties
            if (dd.s > g.s) hh = dd;
            N.cs[j & M] = hh.s; N.cm[j & M] = hh.nm; N.cx[j & M] = hh.nx;
            const Cell u{hh.s > NEGH ? hh.s + (int32_t)(j - lo) * E : NEG, hh.nm, hh.nx};
            run = cmax_left(run, u);
        }
        // pass 2: exclusive max-plus scan of the chunk aggregates over the workgroup (ties to the left)
        const Cell winc = wave_incl_maxscan(run);
        if (lane == 63) s_scan[wave] = winc;
        __syncthreads();
        Cell acc{NEG, 0, 0};
        for (uint32_t w = 0; w < wave; w++) acc = cmax_left(acc, s_scan[w]);
        acc = cmax_left(acc, dpp_cell<0x138, 0xf>(winc));  // best u of every column left of my chunk
        // pass 3: C = max(H, I), prune, row statistics
        Best4 rb{NEG, 0xFFFFFFFFu, 0, 0};
        uint32_t myfirst = 0xFFFFFFFFu, mylast = 0;
        for (uint32_t j = j0; j < j1; j++) {
            const Cell hh{N.cs[j & M], N.cm[j & M], N.cx[j & M]};
            Cell I{NEG, acc.nm, acc.nx};
            if (acc.s > NEGH) I.s = acc.s - O - (int32_t)(j - lo) * E;
            const Cell u{hh.s > NEGH ? hh.s + (int32_t)(j - lo) * E : NEG, hh.nm, hh.nx};
            acc = cmax_left(acc, u);
            Cell c = hh;  // H preferred over I on ties
            if (I.s > c.s) c = I;
            const bool alive = c.s >= thr && c.s > NEGH;
            N.cs[j & M] = alive ? c.s : NEG; N.cm[j & M] = c.nm; N.cx[j & M] = c.nx;
            if (!alive) N.ds[j & M] = NEG;
            if (alive) {
                if (myfirst == 0xFFFFFFFFu) myfirst = j;
                mylast = j;
                if (c.s > rb.s) { rb.s = c.s; rb.j = j; rb.nm = c.nm; rb.nx = c.nx; }
            }
        }
        // workgroup reductions: first / last live column, best cell (smallest column on ties: chunks grow with tid)
        uint32_t wf = myfirst, wl = (myfirst == 0xFFFFFFFFu) ? 0u : mylast + 1u;  // last + 1 so that 0 = none
        for (int o = 32; o > 0; o >>= 1) { wf = min(wf, (uint32_t)__shfl_xor((int)wf, o)); wl = max(wl, (uint32_t)__shfl_xor((int)wl, o)); }
        const Best4 wb4 = wave_best(rb);
        if (lane == 0) { s_first[wave] = wf; s_last[wave] = wl; s_best[wave] = wb4; }
        __syncthreads();
        uint32_t first_alive = 0xFFFFFFFFu, last1 = 0;
        Best4 tb{NEG, 0xFFFFFFFFu, 0, 0};
        for (int w = 0; w < ANY_THREADS / 64; w++) {
            first_alive = min(first_alive, s_first[w]);
            last1 = max(last1, s_last[w]);
            const Best4 o = s_best[w];
            if (o.s > tb.s || (o.s == tb.s && o.j < tb.j)) tb = o;
        }
        __syncthreads();  // the LDS arrays are rewritten in the next row
        if (first_alive == 0xFFFFFFFFu) break;
        lo = first_alive;
        hi = last1 - 1u;
        best.maxcols = max(best.maxcols, hi - lo + 1u);
        best.rows = i;
        if (tb.s > best.score) { best.score = tb.s; best.i = i; best.j = tb.j; best.nm = tb.nm; best.nx = tb.nx; }
        // The cells are 32 bits wide (as lastz's own score_t, which wraps there).  Every live cell of a row lies within the
        // y-drop of the best score so far, so when that nears the cap the whole row — and the best score — is moved down by
        // half the cap and `sbase` up: comparisons inside a row and between neighbouring rows never see the difference, and
        // the half's score is sbase + best.score in 64 bits
        if (best.score > cap) {
            const int32_t K = cap / 2;
            for (uint32_t j = lo + tid; j <= hi; j += ANY_THREADS) {
                if (N.cs[j & M] > NEGH) N.cs[j & M] -= K;
                if (N.ds[j & M] > NEGH) N.ds[j & M] -= K;
            }
            best.score -= K;
            sbase += K;
            __syncthreads();
        }
    }
    best.base_lo = (uint32_t)(unsigned long long)sbase;
    best.base_hi = (uint32_t)((unsigned long long)sbase >> 32);
    best.overflow = overflow ? 1u : 0u;
    if (tid == 0) res[job.slot] = best;
}

// one wave per group: finalise anchors in rank order as far as DP results exist
constexpr uint32_t RESOLVE_NEW = 256;  // boxes accepted per invocation that fit the LDS list
__global__ __launch_bounds__(64) void k6_resolve(Group *__restrict__ groups, const uint2 *__restrict__ anchors,
                                                 const HalfResult *__restrict__ res, mimeo_alignment *__restrict__ aln,
                                                 uint8_t *__restrict__ astate, unsigned int *__restrict__ remaining) {
    __shared__ uint4 sbox[RESOLVE_NEW];  // boxes accepted in this invocation (tstart, tend, qstart, qend)
    Group &G = groups[blockIdx.x];
    const uint64_t b0 = G.hsp_begin;
    const uint32_t nacc0 = G.nacc;
    uint32_t nnew = 0, overflow = 0, r = G.next;
    for (; r < G.nchain; r++) {
        const uint8_t st = astate[b0 + r];
        if (st == A_SKIPPED || st == A_ACCEPTED) continue;
        if (nnew == RESOLVE_NEW) break;
        const uint2 a = anchors[b0 + r];
        bool inside = in_boxes(aln + b0, nacc0, a);  // boxes of earlier invocations (visible in global memory)
        if (!inside) {
            bool in2 = false;
            for (uint32_t e = threadIdx.x; e < nnew; e += 64) {
                uint4 o = sbox[e];
                if (a.x >= o.x && a.x < o.y && a.y >= o.z && a.y < o.w) in2 = true;
            }
            inside = __ballot(in2) != 0;
        }
        if (inside) {  // also for an anchor without DP result: a deferred anchor swallowed by now never needs one
            if (threadIdx.x == 0) astate[b0 + r] = A_SKIPPED;
            continue;
        }
        if (st == A_NEW) break;  // not swallowed and no DP result yet: it is scheduled in the next round
        const uint32_t slot = 2u * (uint32_t)(b0 + r);
        HalfResult L = res[slot], R = res[slot + 1];
        overflow |= L.overflow | R.overflow;
        if (threadIdx.x == 0) {
            mimeo_alignment m;
            m.tid = G.tid; m.qid = G.qid; m.qstrand = G.minus; m.reserved = 0;
            m.tstart = a.x - L.i; m.tend = a.x + R.i; m.qstart = a.y - L.j; m.qend = a.y + R.j;
            m.score = half_score(L) + half_score(R);
            m.id_n = L.nm + R.nm;
            m.id_d = L.nm + R.nm + L.nx + R.nx;
            aln[b0 + nacc0 + nnew] = m;
            sbox[nnew] = make_uint4(m.tstart, m.tend, m.qstart, m.qend);
            astate[b0 + r] = A_ACCEPTED;
        }
        nnew++;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        G.next = r;
        G.nacc = nacc0 + nnew;
        G.nbatch = 0;
        if (overflow) G.overflow = 1;
        if (r < G.nchain) atomicAdd(remaining, 1u);
    }
}

// gap-free mode (--gapped off): every chained HSP is reported as is; identity by popcount
__global__ __launch_bounds__(256) void k6_ungapped(Group *__restrict__ groups, const mimeo_hsp *__restrict__ hs,
                                                   const uint32_t *__restrict__ order, mimeo_alignment *__restrict__ aln) {
    Group &G = groups[blockIdx.x];
    const uint64_t b0 = G.hsp_begin;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t r = threadIdx.x >> 6; r < G.nchain; r += 4) {
        const mimeo_hsp h = hs[b0 + order[b0 + r]];
        const int32_t d = (int32_t)h.tstart - (int32_t)h.qstart;
        uint32_t nmatch = 0;
        for (uint32_t w0 = lane * 32u; w0 < h.length; w0 += 64u * 32u) {
            int32_t pt = (int32_t)(h.tstart + w0), pq = pt - d;
            const Win32 tw = win32(G.T, pt), qw = win32(G.Q, pq);
            uint32_t mm = ~((tw.lo ^ qw.lo) | (tw.hi ^ qw.hi)) & ~(tw.nm | qw.nm);
            uint32_t rem = h.length - w0;
            if (rem < 32) mm &= (1u << rem) - 1u;
            nmatch += __popc(mm);
        }
        for (int o = 32; o > 0; o >>= 1) nmatch += __shfl_xor(nmatch, o);
        if (lane == 0) {
            mimeo_alignment m;
            m.tid = G.tid; m.qid = G.qid; m.qstrand = G.minus; m.reserved = 0;
            m.tstart = h.tstart; m.tend = h.tstart + h.length; m.qstart = h.qstart; m.qend = h.qstart + h.length;
            m.score = h.score; m.id_n = nmatch; m.id_d = h.length;
            aln[b0 + r] = m;
        }
    }
    if (threadIdx.x == 0) G.nacc = G.nchain;
}

// threshold, minus-strand coordinates -> query plus strand (start2+/end2+), compaction
__global__ void k6_finish(Group *__restrict__ groups, uint32_t ngroups, mimeo_alignment *__restrict__ aln,
                          int32_t thresh) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    Group &G = groups[g];
    const uint64_t b0 = G.hsp_begin;
    uint32_t k = 0;
    for (uint32_t e = 0; e < G.nacc; e++) {
        mimeo_alignment a = aln[b0 + e];
        if (a.score < thresh) continue;
        if (G.minus) { uint32_t s = G.Q.len - a.qend, t2 = G.Q.len - a.qstart; a.qstart = s; a.qend = t2; }
        aln[b0 + k++] = a;
    }
    G.naln = k;
}

// The alignments of group g sit at d_aln[hsp_begin .. hsp_begin + naln): a few thousand records scattered over an array of
// one slot per HSP (70 MB on a C4 row).  Packed densely before the read-back: job0 = the group's first dense slot.
__global__ __launch_bounds__(1024) void k6_dense_offsets(Group *__restrict__ groups, uint32_t ngroups) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t g0 = 0; g0 < ngroups; g0 += 1024) {
        const uint32_t g = g0 + threadIdx.x, n = g < ngroups ? groups[g].naln : 0u;
        part[threadIdx.x] = n;
        __syncthreads();
        for (uint32_t o = 1; o < 1024; o <<= 1) {
            const uint32_t v = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
            __syncthreads();
            part[threadIdx.x] += v;
            __syncthreads();
        }
        if (g < ngroups) groups[g].job0 = carry + part[threadIdx.x] - n;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
}
__global__ __launch_bounds__(64) void k6_dense_copy(const Group *__restrict__ groups, const mimeo_alignment *__restrict__ aln,
                                                    mimeo_alignment *__restrict__ dense) {
    const Group &G = groups[blockIdx.x];
    for (uint32_t k = threadIdx.x; k < G.naln; k += 64) dense[G.job0 + k] = aln[G.hsp_begin + k];
}
void dense_alignments_device(Group *d_groups, uint32_t ngroups, const mimeo_alignment *d_aln, mimeo_alignment *d_dense) {
    hipLaunchKernelGGL(k6_dense_offsets, dim3(1), dim3(1024), 0, stream(), d_groups, ngroups);
    hipLaunchKernelGGL(k6_dense_copy, dim3(ngroups), dim3(64), 0, stream(), (const Group *)d_groups, d_aln, d_dense);
}

static DeviceBuf g_anchors, g_packed, g_jobs, g_res, g_cnt, g_astate, g_ovf_list, g_any;

int gapped_device(Group *d_groups, uint32_t ngroups, const mimeo_hsp *d_sorted, const uint32_t *d_order,
                  uint64_t nhsps, const mimeo_params *p, mimeo_alignment *d_aln) {
    if (!ngroups || !nhsps) return 0;
    hipStream_t st = stream();
    int rc;
    if (!p->gapped) {
        hipLaunchKernelGGL(k6_ungapped, dim3(ngroups), dim3(256), 0, st, d_groups, d_sorted, d_order, d_aln);
    } else {
        uint32_t bmax = 8192u / ngroups;  // anchors per group and round (200 units: 32 -> one large round and a short one)
        if (getenv("MIMEO_K6_BMAX")) bmax = (uint32_t)atoi(getenv("MIMEO_K6_BMAX"));
        const char *kmode = getenv("MIMEO_K6_KERNEL");       // development switches: read once per call, not per round
        const bool k6_stats = getenv("MIMEO_K6_STATS") != nullptr;
        // 32-bit DP cells: beyond this score a half extension goes to (or, in k6_dp_any, rebases its cells in) the last kernel;
        // MIMEO_K6_SCORE_CAP lowers it so that tests of ordinary size take that road
        const int32_t cap = getenv("MIMEO_K6_SCORE_CAP") ? std::max(100000, atoi(getenv("MIMEO_K6_SCORE_CAP"))) : 2000000000;
        bmax = bmax < 1 ? 1 : (bmax > MAX_BATCH ? MAX_BATCH : bmax);
        if ((rc = g_anchors.reserve(nhsps * sizeof(uint2)))) return rc;
        if ((rc = g_packed.reserve(nhsps * 8))) return rc;
        HIP_TRY(hipMemsetAsync(g_packed.p, 0, nhsps * 8, st));
        if ((rc = g_jobs.reserve((size_t)ngroups * bmax * 2 * sizeof(DpJob)))) return rc;
        if ((rc = g_res.reserve((size_t)nhsps * 2 * sizeof(HalfResult)))) return rc;
        if ((rc = g_astate.reserve((size_t)nhsps * 2))) return rc;  // state | defer count
        HIP_TRY(hipMemsetAsync(g_astate.p, 0, (size_t)nhsps * 2, st));
        if ((rc = g_cnt.reserve(16))) return rc;
        if ((rc = g_ovf_list.reserve((size_t)ngroups * bmax * 2 * sizeof(unsigned int)))) return rc;
        // many small groups (scaffold pairs of a packed fragmented assembly): one workgroup each will do
        const uint32_t asplit = ngroups > 4096 ? 1u : ANCHOR_SPLIT;
        hipLaunchKernelGGL(k6_anchor_points, dim3(ngroups, asplit), dim3(ANCHOR_THREADS), 0, st, (const Group *)d_groups,
                           d_sorted, d_order, (unsigned long long *)g_packed.p);
        hipLaunchKernelGGL(k6_anchor_final, dim3(ngroups), dim3(256), 0, st, (const Group *)d_groups, d_sorted, d_order,
                           (const unsigned long long *)g_packed.p, (uint2 *)g_anchors.p);
        for (;;) {
            HIP_TRY(hipMemsetAsync(g_cnt.p, 0, 16, st));
            unsigned int *njobs = (unsigned int *)g_cnt.p, *remaining = njobs + 1, *novf = njobs + 2;
            hipLaunchKernelGGL(k6_pick, dim3(ngroups), dim3(64), 0, st, d_groups, (const uint2 *)g_anchors.p,
                               (const mimeo_alignment *)d_aln, bmax, (uint8_t *)g_astate.p, (uint8_t *)g_astate.p + nhsps,
                               (DpJob *)g_jobs.p, njobs);
            unsigned int h[2] = {0, 0};
            HIP_TRY(hipMemcpyAsync(h, g_cnt.p, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (h[0]) {
                // MIMEO_K6_KERNEL (development A/B): "dp4" = the four-wavefront kernel then wave_half_extend<16>, "dp" = the
                // latter alone; default = the lean single-wavefront kernel (its dead-cell arithmetic needs the
                // penalties to stay far below 2^29 / 1024)
                const bool lean_ok = p->gap_extend <= (1 << 16) && p->gap_open <= (1 << 24) && p->ydrop <= (1 << 28);
                if (lean_ok && !kmode) {
                    hipLaunchKernelGGL(k6_dp1, dim3(h[0]), dim3(64), 0, st, (const Group *)d_groups, (const DpJob *)g_jobs.p,
                                       (HalfResult *)g_res.p, p->gap_open, p->gap_extend, p->ydrop);
                } else {
                    const bool use4 = !kmode || !strcmp(kmode, "dp4");
                    if (use4)
                        hipLaunchKernelGGL(k6_dp4, dim3(h[0]), dim3(C4_THREADS), 0, st, (const Group *)d_groups,
                                           (const DpJob *)g_jobs.p, (HalfResult *)g_res.p, p->gap_open, p->gap_extend, p->ydrop);
                    hipLaunchKernelGGL(k6_dp, dim3(h[0]), dim3(64), 0, st, (const Group *)d_groups, (const DpJob *)g_jobs.p,
                                       (HalfResult *)g_res.p, p->gap_open, p->gap_extend, p->ydrop, use4 ? 1 : 0, cap);
                }
                hipLaunchKernelGGL(k6_dp_wide, dim3(h[0]), dim3(64), 0, st, (const Group *)d_groups,
                                   (const DpJob *)g_jobs.p, (HalfResult *)g_res.p, p->gap_open, p->gap_extend, p->ydrop, novf,
                                   (unsigned int *)g_ovf_list.p, cap);
                // bands beyond 2048 columns (tandem arrays): the global-memory kernel, a few jobs at a time
                unsigned int nov = 0;
                HIP_TRY(hipMemcpyAsync(&nov, novf, 4, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                if (nov) {
                    const unsigned int slots = std::min<unsigned int>(nov, 32u);
                    if ((rc = g_any.reserve((size_t)slots * ANY_SLOT_WORDS * 4))) return rc;
                    for (unsigned int f = 0; f < nov; f += slots)
                        hipLaunchKernelGGL(k6_dp_any, dim3(std::min(slots, nov - f)), dim3(ANY_THREADS), 0, st, (const Group *)d_groups,
                                           (const DpJob *)g_jobs.p, (const unsigned int *)g_ovf_list.p, f, (HalfResult *)g_res.p,
                                           (uint32_t *)g_any.p, p->gap_open, p->gap_extend, p->ydrop, cap);
                }
            }
            if (k6_stats && h[0]) {
                std::vector<DpJob> hj(h[0]);
                std::vector<HalfResult> hall((size_t)nhsps * 2), hr(h[0]);
                HIP_TRY(hipStreamSynchronize(st));
                HIP_TRY(hipMemcpy(hj.data(), g_jobs.p, (size_t)h[0] * sizeof(DpJob), hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(hall.data(), g_res.p, hall.size() * sizeof(HalfResult), hipMemcpyDeviceToHost));
                for (size_t k = 0; k < hr.size(); k++) hr[k] = hall[hj[k].slot];
                unsigned long long hist[9] = {0}, rows = 0, maxr = 0, shortcut = 0;
                for (auto &r : hr) {
                    if (!r.rows) { shortcut++; continue; }
                    hist[std::min<uint32_t>(8, r.maxcols / 128)]++; rows += r.rows; maxr = std::max<unsigned long long>(maxr, r.rows);
                }
                unsigned long long rebased = 0;
                for (auto &r : hr) if (r.base_lo | r.base_hi) rebased++;
                fprintf(stderr, "[k6] jobs %u shortcut %llu (rebased in k6_dp_any: %llu) rows total %llu max %llu band<128..>=1024:", h[0], shortcut, rebased, rows, maxr);
                for (int b = 0; b < 9; b++) fprintf(stderr, " %llu", hist[b]);
                fprintf(stderr, "\n");
                {
                    unsigned long long fine[34] = {0};
                    for (auto &r : hr) if (r.rows) fine[std::min<uint32_t>(33, r.maxcols / 32)]++;
                    fprintf(stderr, "[k6] widest band, buckets of 32 columns from 448:");
                    for (int b = 14; b < 34; b++) fprintf(stderr, " %llu", fine[b]);
                    fprintf(stderr, "\n");
                }
                int shown = 0;
                for (auto &r : hr)
                    if (!r.rows && shown < 8) { fprintf(stderr, "  [k6] zero-row job: score %d i %u j %u nm %u nx %u ovf %u\n", r.score, r.i, r.j, r.nm, r.nx, r.overflow); shown++; }
            }
            hipLaunchKernelGGL(k6_resolve, dim3(ngroups), dim3(64), 0, st, d_groups, (const uint2 *)g_anchors.p,
                               (const HalfResult *)g_res.p, d_aln, (uint8_t *)g_astate.p, remaining);
            HIP_TRY(hipMemcpyAsync(h, g_cnt.p, 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (!h[1]) break;
        }
    }
    hipLaunchKernelGGL(k6_finish, dim3((ngroups + 63) / 64), dim3(64), 0, st, d_groups, ngroups, d_aln, p->hspthresh);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace mimeo

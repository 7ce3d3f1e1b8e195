// K5 + K6 — lastz `--chain` and `--gapped` (reference call site
// src/mimeo/wrappers.py:1031 `--chain --gapped`).
//
// Work unit = one group = one (target scaffold, query scaffold, strand).  Groups are independent;
// inside a group lastz's stages are order-dependent (the chain DP, and anchors processed by
// decreasing score with anchors inside an earlier alignment skipped), so one workgroup owns a
// group and the parallelism inside it is over HSPs (K5) and over the columns of one DP row (K6).
//
//   k5_chain  : rank-sort the group's HSPs by (tstart, qstart, length); chain DP
//               best[j] = score[j] + max(0, max{best[i] : i ends before j starts in both
//               sequences}), ties to the earliest i / earliest end; flag the chain; order the
//               chained HSPs by (score desc, tstart, qstart, length) for K6.
//   k6_gapped : per anchor (centre of the best 31-column window of the HSP) two one-sided
//               y-drop affine DPs, evaluated row by row: a row's cells are independent except for
//               the horizontal gap state, which is a max-plus prefix scan (u_k = H_k + k*E).
//               Rows live in an LDS ring of RING columns; every cell carries (score, matches,
//               mismatches) so identity needs no traceback.  Pruning: a cell whose score is below
//               (best score of the rows above) - ydrop is dead (DESIGN.md spec v1 §7).
#include "device_util.h"

namespace mimeo {

constexpr int CH_THREADS = 256;
constexpr int GP_THREADS = 512;
constexpr int GP_WAVES = GP_THREADS / 64;
constexpr int RING = 2048;
constexpr int32_t NEG = -(1 << 30);
constexpr int32_t NEGH = -(1 << 29);

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

__global__ __launch_bounds__(CH_THREADS) void k5_chain(Group *__restrict__ groups, const mimeo_hsp *__restrict__ in,
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
    if (n == 0) { if (tid == 0) G.nchain = 0; return; }
    // 1. rank sort into hs[b0 .. b0+n)
    for (uint32_t i = tid; i < n; i += CH_THREADS) {
        mimeo_hsp me = in[b0 + i];
        uint32_t rank = 0;
        for (uint32_t k = 0; k < n; k++) {
            mimeo_hsp o = in[b0 + k];
            if (hsp_less(o, me) || (!hsp_less(me, o) && k < i)) rank++;
        }
        me.flags = 0;
        hs[b0 + rank] = me;
    }
    __syncthreads();
    if (do_chain) {
        for (uint32_t k = tid; k < n; k += CH_THREADS) { cand[b0 + k] = 0; pred[b0 + k] = -1; }
        __syncthreads();
        for (uint32_t j = 0; j < n; j++) {
            mimeo_hsp hj = hs[b0 + j];
            long long bj = cand[b0 + j] + hj.score;
            if (tid == 0) best[b0 + j] = bj;
            uint32_t te = hj.tstart + hj.length, qe = hj.qstart + hj.length;
            // relax every later HSP that starts after hj ends (strict improvement keeps the earliest j)
            uint32_t k0 = j + 1 + ((tid + CH_THREADS - ((j + 1) % CH_THREADS)) % CH_THREADS);
            for (uint32_t k = k0; k < n; k += CH_THREADS) {
                const mimeo_hsp &hk = hs[b0 + k];
                if (te <= hk.tstart && qe <= hk.qstart && bj > cand[b0 + k]) { cand[b0 + k] = bj; pred[b0 + k] = (int)j; }
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
    // 2. anchor order of the flagged HSPs (rank among flagged by anchor_less)
    for (uint32_t i = tid; i < n; i += CH_THREADS) {
        mimeo_hsp me = hs[b0 + i];
        if (!(me.flags & 1u)) continue;
        uint32_t rank = 0;
        for (uint32_t k = 0; k < n; k++) {
            const mimeo_hsp &o = hs[b0 + k];
            if ((o.flags & 1u) && k != i && anchor_less(o, me)) rank++;
        }
        order[b0 + rank] = i;
    }
    if (tid == 0) G.nchain = s_m;
}

// ---- K6 ------------------------------------------------------------------------------------
struct Cell {
    int32_t s;
    uint32_t nm, nx;
};
struct HalfResult {
    int32_t score;
    uint32_t i, j, nm, nx;
};

struct GpShared {
    int32_t Cs[2][RING];
    uint32_t Cm[2][RING], Cx[2][RING];
    int32_t Ds[2][RING];
    uint32_t Dm[2][RING], Dx[2][RING];
    Cell wtot[GP_WAVES];
    uint32_t wfirst[GP_WAVES], wlast[GP_WAVES], wbj[GP_WAVES];
    Cell wbest[GP_WAVES];
    HalfResult res;
    int overflow;
    long long red64[GP_WAVES];
    uint32_t redu[GP_WAVES];
};

__device__ __forceinline__ Cell cmax_left(const Cell &l, const Cell &r) { return r.s > l.s ? r : l; }  // ties -> left

// One-sided y-drop affine extension, evaluated by the whole workgroup.  Result in sh.res.
__device__ void half_extend(GpShared &sh, const StrandView &T, const StrandView &Q, uint32_t at, uint32_t aq, int dir,
                            int32_t O, int32_t E, int32_t Y) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t lenA = dir > 0 ? T.len - at : at, lenB = dir > 0 ? Q.len - aq : aq;
    uint32_t hi0 = 0;
    if (Y >= O + E) hi0 = min(lenB, (uint32_t)((Y - O) / E));
    if (hi0 + 2 + GP_THREADS > RING) {  // uniform: row 0 alone would not fit the LDS ring
        if (tid == 0) { sh.overflow = 1; sh.res = HalfResult{0, 0, 0, 0, 0}; }
        __syncthreads();
        return;
    }
    for (uint32_t j = tid; j <= hi0; j += GP_THREADS) {
        sh.Cs[0][j & (RING - 1)] = j ? -O - (int32_t)j * E : 0;
        sh.Cm[0][j & (RING - 1)] = 0;
        sh.Cx[0][j & (RING - 1)] = 0;
        sh.Ds[0][j & (RING - 1)] = NEG;
    }
    if (tid == 0) sh.res = HalfResult{0, 0, 0, 0, 0};
    __syncthreads();
    uint32_t plo = 0, phi = hi0;
    int p = 0;  // index of the previous row's buffers
    HalfResult best{0, 0, 0, 0, 0};
    for (uint32_t i = 1; i <= lenA; i++) {
        const int32_t thr = best.score - Y;
        const int32_t pa = dir > 0 ? (int32_t)(at + i - 1) : (int32_t)(at - i);
        const uint32_t alo = getbit(T.lo, pa), ahi = getbit(T.hi, pa), an = getbit(T.nm, pa);
        const uint32_t jlo = plo, jmax = min(phi + 1, lenB);
        const int c = p ^ 1;
        Cell carry{NEG, 0, 0};
        uint32_t first = 0xFFFFFFFFu, last = 0, rbj = 0;
        Cell rb{NEG, 0, 0};
        for (uint32_t ch = jlo / GP_THREADS;; ch++) {
            const uint32_t j = ch * GP_THREADS + tid;
            const bool inrow = j >= jlo && j <= lenB;
            Cell d{NEG, 0, 0}, g{NEG, 0, 0};
            if (inrow && j <= jmax) {
                if (j >= plo && j <= phi) {
                    uint32_t sl = j & (RING - 1);
                    int32_t pds = sh.Ds[p][sl], pcs = sh.Cs[p][sl];
                    if (pds > NEGH) { d.s = pds - E; d.nm = sh.Dm[p][sl]; d.nx = sh.Dx[p][sl]; }
                    if (pcs > NEGH && pcs - O - E > d.s) { d.s = pcs - O - E; d.nm = sh.Cm[p][sl]; d.nx = sh.Cx[p][sl]; }
                }
                if (j >= 1 && j - 1 >= plo && j - 1 <= phi) {
                    uint32_t sl = (j - 1) & (RING - 1);
                    int32_t pcs = sh.Cs[p][sl];
                    if (pcs > NEGH) {
                        int32_t pb = dir > 0 ? (int32_t)(aq + j - 1) : (int32_t)(aq - j);
                        uint32_t dl = alo ^ getbit(Q.lo, pb), dh = ahi ^ getbit(Q.hi, pb);
                        uint32_t nn = an | getbit(Q.nm, pb);
                        g.s = pcs + sub_score(dl, dh, alo ^ ahi, nn);
                        bool m = !(dl | dh | nn);
                        g.nm = sh.Cm[p][sl] + (m ? 1u : 0u);
                        g.nx = sh.Cx[p][sl] + (m ? 0u : 1u);
                    }
                }
            }
            Cell h = g;  // diagonal preferred on ties
            if (d.s > h.s) h = d;
            // horizontal gap state: exclusive max-plus prefix scan of u_k = H_k + (k - jlo) * E
            Cell u = h;
            u.s = h.s > NEGH ? h.s + (int32_t)(j - jlo) * E : NEG;
            Cell inc = u;
            for (int o = 1; o < 64; o <<= 1) {
                Cell l;
                l.s = __shfl_up(inc.s, o); l.nm = __shfl_up(inc.nm, o); l.nx = __shfl_up(inc.nx, o);
                if (lane >= (uint32_t)o) inc = cmax_left(l, inc);
            }
            if (lane == 63) sh.wtot[wave] = inc;
            Cell ex;
            ex.s = __shfl_up(inc.s, 1); ex.nm = __shfl_up(inc.nm, 1); ex.nx = __shfl_up(inc.nx, 1);
            if (lane == 0) ex = Cell{NEG, 0, 0};
            __syncthreads();
            Cell pre = carry, tot = carry;
#pragma unroll
            for (int w = 0; w < GP_WAVES; w++) {
                Cell t = sh.wtot[w];
                if ((uint32_t)w < wave) pre = cmax_left(pre, t);
                tot = cmax_left(tot, t);
            }
            ex = cmax_left(pre, ex);
            Cell I{NEG, ex.nm, ex.nx};
            if (ex.s > NEGH) I.s = ex.s - O - (int32_t)(j - jlo) * E;
            Cell cc = h;  // H preferred over I on ties
            if (I.s > cc.s) cc = I;
            const bool alive = inrow && cc.s >= thr && cc.s > NEGH;
            if (inrow) {
                uint32_t sl = j & (RING - 1);
                sh.Cs[c][sl] = alive ? cc.s : NEG; sh.Cm[c][sl] = cc.nm; sh.Cx[c][sl] = cc.nx;
                sh.Ds[c][sl] = alive ? d.s : NEG; sh.Dm[c][sl] = d.nm; sh.Dx[c][sl] = d.nx;
            }
            if (alive) {
                if (first == 0xFFFFFFFFu) first = j;
                last = j;
                if (cc.s > rb.s) { rb = cc; rbj = j; }
            }
            carry = tot;
            const uint32_t chunk_end = ch * GP_THREADS + GP_THREADS - 1;
            bool stop = chunk_end >= lenB;
            if (!stop && chunk_end >= jmax) {
                int32_t inext = carry.s > NEGH ? carry.s - O - (int32_t)(chunk_end + 1 - jlo) * E : NEG;
                stop = inext < thr;
            }
            __syncthreads();  // wtot is reused by the next chunk
            if (stop) break;
        }
        // row reduce: first / last alive column, best cell (max score, smallest column)
        for (int o = 32; o > 0; o >>= 1) {
            first = min(first, (uint32_t)__shfl_xor(first, o));
            last = max(last, (uint32_t)__shfl_xor(last, o));
            Cell ob;
            ob.s = __shfl_xor(rb.s, o); ob.nm = __shfl_xor(rb.nm, o); ob.nx = __shfl_xor(rb.nx, o);
            uint32_t oj = __shfl_xor(rbj, o);
            if (ob.s > rb.s || (ob.s == rb.s && ob.s > NEG && oj < rbj)) { rb = ob; rbj = oj; }
        }
        if (lane == 0) { sh.wfirst[wave] = first; sh.wlast[wave] = last; sh.wbest[wave] = rb; sh.wbj[wave] = rbj; }
        __syncthreads();
        first = sh.wfirst[0]; last = sh.wlast[0]; rb = sh.wbest[0]; rbj = sh.wbj[0];
#pragma unroll
        for (int w = 1; w < GP_WAVES; w++) {
            first = min(first, sh.wfirst[w]);
            last = max(last, sh.wlast[w]);
            Cell ob = sh.wbest[w];
            uint32_t oj = sh.wbj[w];
            if (ob.s > rb.s || (ob.s == rb.s && ob.s > NEG && oj < rbj)) { rb = ob; rbj = oj; }
        }
        __syncthreads();  // everyone has read the w* arrays before the next row overwrites them
        if (first == 0xFFFFFFFFu) break;
        if (rb.s > best.score) best = HalfResult{rb.s, i, rbj, rb.nm, rb.nx};
        plo = first; phi = last; p = c;
        if (phi - plo + 2 + GP_THREADS > RING) { if (tid == 0) sh.overflow = 1; break; }
    }
    if (tid == 0) sh.res = best;
    __syncthreads();
}

// best 31-column window of an HSP: offset of its centre (length <= 31: length / 2)
__device__ uint32_t anchor_offset(GpShared &sh, const StrandView &T, const StrandView &Q, const mimeo_hsp &h) {
    const uint32_t Wn = 31;
    if (h.length <= Wn) return h.length / 2;
    const uint32_t tid = threadIdx.x, nw = h.length - Wn + 1;
    const int32_t d = (int32_t)h.tstart - (int32_t)h.qstart;
    // each thread slides over a contiguous range of window starts
    uint32_t per = (nw + GP_THREADS - 1) / GP_THREADS, w0 = tid * per, w1 = min(nw, w0 + per);
    long long bs = INT64_MIN;
    uint32_t bw = 0xFFFFFFFFu;
    if (w0 < w1) {
        long long sum = 0;
        bool m;
        for (uint32_t k = 0; k < Wn; k++) sum += pair_score(T, Q, (int32_t)(h.tstart + w0 + k), (int32_t)(h.tstart + w0 + k) - d, &m);
        bs = sum; bw = w0;
        for (uint32_t w = w0 + 1; w < w1; w++) {
            int32_t add = (int32_t)(h.tstart + w + Wn - 1), sub = (int32_t)(h.tstart + w - 1);
            sum += pair_score(T, Q, add, add - d, &m) - pair_score(T, Q, sub, sub - d, &m);
            if (sum > bs) { bs = sum; bw = w; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        long long ob = __shfl_xor(bs, o);
        uint32_t ow = __shfl_xor(bw, o);
        if (ob > bs || (ob == bs && ow < bw)) { bs = ob; bw = ow; }
    }
    if ((tid & 63) == 0) { sh.red64[tid >> 6] = bs; sh.redu[tid >> 6] = bw; }
    __syncthreads();
    bs = sh.red64[0]; bw = sh.redu[0];
    for (int w = 1; w < GP_WAVES; w++)
        if (sh.red64[w] > bs || (sh.red64[w] == bs && sh.redu[w] < bw)) { bs = sh.red64[w]; bw = sh.redu[w]; }
    __syncthreads();
    return bw + Wn / 2;
}

__global__ __launch_bounds__(GP_THREADS) void k6_gapped(Group *__restrict__ groups, const mimeo_hsp *__restrict__ hs,
                                                        const uint32_t *__restrict__ order,
                                                        mimeo_alignment *__restrict__ aln, int32_t O, int32_t E,
                                                        int32_t Y, int32_t thresh, int do_gapped) {
    __shared__ GpShared sh;
    Group &G = groups[blockIdx.x];
    const uint64_t b0 = G.hsp_begin;
    const uint32_t m = G.nchain, tid = threadIdx.x;
    const StrandView T = G.T, Q = G.Q;
    if (tid == 0) sh.overflow = 0;
    __syncthreads();
    uint32_t cnt = 0;  // alignments computed so far (uniform)
    for (uint32_t r = 0; r < m; r++) {
        const mimeo_hsp h = hs[b0 + order[b0 + r]];
        mimeo_alignment a;
        a.tid = G.tid; a.qid = G.qid; a.qstrand = G.minus; a.reserved = 0;
        if (!do_gapped) {
            // gap-free HSP reported as is; identity by popcount
            uint32_t nmatch = 0;
            const int32_t d = (int32_t)h.tstart - (int32_t)h.qstart;
            for (uint32_t w0 = tid * 32u; w0 < h.length; w0 += GP_THREADS * 32u) {
                int32_t pt = (int32_t)(h.tstart + w0), pq = pt - d;
                uint32_t mm = ~((get32(T.lo, pt) ^ get32(Q.lo, pq)) | (get32(T.hi, pt) ^ get32(Q.hi, pq))) &
                              ~(get32(T.nm, pt) | get32(Q.nm, pq));
                uint32_t rem = h.length - w0;
                if (rem < 32) mm &= (1u << rem) - 1u;
                nmatch += __popc(mm);
            }
            for (int o = 32; o > 0; o >>= 1) nmatch += __shfl_xor(nmatch, o);
            if ((tid & 63) == 0) sh.redu[tid >> 6] = nmatch;
            __syncthreads();
            nmatch = 0;
            for (int w = 0; w < GP_WAVES; w++) nmatch += sh.redu[w];
            __syncthreads();
            a.tstart = h.tstart; a.tend = h.tstart + h.length; a.qstart = h.qstart; a.qend = h.qstart + h.length;
            a.score = h.score; a.id_n = nmatch; a.id_d = h.length;
        } else {
            uint32_t off = anchor_offset(sh, T, Q, h);
            uint32_t at = h.tstart + off, aq = h.qstart + off;
            int inside = 0;
            for (uint32_t e = tid; e < cnt; e += GP_THREADS) {
                const mimeo_alignment &o = aln[b0 + e];
                if (at >= o.tstart && at < o.tend && aq >= o.qstart && aq < o.qend) inside = 1;
            }
            if (__syncthreads_or(inside)) continue;
            half_extend(sh, T, Q, at, aq, -1, O, E, Y);
            HalfResult L = sh.res;
            __syncthreads();
            half_extend(sh, T, Q, at, aq, +1, O, E, Y);
            HalfResult R = sh.res;
            __syncthreads();
            a.tstart = at - L.i; a.tend = at + R.i; a.qstart = aq - L.j; a.qend = aq + R.j;
            a.score = (int64_t)L.score + R.score;
            a.id_n = L.nm + R.nm;
            a.id_d = L.nm + R.nm + L.nx + R.nx;
        }
        if (tid == 0) aln[b0 + cnt] = a;
        cnt++;
        __syncthreads();
    }
    // threshold + minus-strand coordinates -> query plus strand (start2+/end2+); compact in place
    if (tid == 0) {
        uint32_t k = 0;
        for (uint32_t e = 0; e < cnt; e++) {
            mimeo_alignment a = aln[b0 + e];
            if (a.score < thresh) continue;
            if (G.minus) { uint32_t s = Q.len - a.qend, t2 = Q.len - a.qstart; a.qstart = s; a.qend = t2; }
            aln[b0 + k++] = a;
        }
        G.naln = k;
        G.overflow = (uint32_t)sh.overflow;
    }
}

int chain_gapped_device(Group *d_groups, uint32_t ngroups, const mimeo_hsp *d_hsps, uint64_t nhsps,
                        const mimeo_params *p, DeviceBuf &scratch, mimeo_alignment *d_aln, float *ms_chain,
                        float *ms_gapped) {
    if (!ngroups || !nhsps) return 0;
    hipStream_t st = stream();
    // scratch: hs | best | cand | pred | order
    size_t off_hs = 0, off_best = off_hs + nhsps * sizeof(mimeo_hsp), off_cand = off_best + nhsps * 8,
           off_pred = off_cand + nhsps * 8, off_order = off_pred + nhsps * 4, total = off_order + nhsps * 4;
    int rc = scratch.reserve(total);
    if (rc) return rc;
    char *b = (char *)scratch.p;
    hipEvent_t e0, e1, e2;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); HIP_TRY(hipEventCreate(&e2));
    HIP_TRY(hipEventRecord(e0, st));
    hipLaunchKernelGGL(k5_chain, dim3(ngroups), dim3(CH_THREADS), 0, st, d_groups, d_hsps, (mimeo_hsp *)(b + off_hs),
                       (long long *)(b + off_best), (long long *)(b + off_cand), (int *)(b + off_pred),
                       (uint32_t *)(b + off_order), p->chain);
    HIP_TRY(hipEventRecord(e1, st));
    hipLaunchKernelGGL(k6_gapped, dim3(ngroups), dim3(GP_THREADS), 0, st, d_groups, (const mimeo_hsp *)(b + off_hs),
                       (const uint32_t *)(b + off_order), d_aln, p->gap_open, p->gap_extend, p->ydrop, p->hspthresh,
                       p->gapped);
    HIP_TRY(hipEventRecord(e2, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    float a = 0, c = 0;
    HIP_TRY(hipEventElapsedTime(&a, e0, e1));
    HIP_TRY(hipEventElapsedTime(&c, e1, e2));
    if (ms_chain) *ms_chain += a;
    if (ms_gapped) *ms_gapped += c;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
    return 0;
}

}  // namespace mimeo

/*
 * mimeo_oracle.c — CPU restatement of the alignment half of mimeo's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under mimeo_amd/ may import, link or execute
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED.  The arithmetic of this path lives in LASTZ, an un-vendored
 * third-party binary (reference environment.yml:7, version not pinned) that is
 * absent from /root/reference and from this image, and the reference ships no test
 * or fixture for it (tests/test_dummy.py:1-4).  This file therefore restates
 * LASTZ's published stage semantics for exactly the flags the reference passes at
 * its call site (src/mimeo/wrappers.py:1025-1037):
 *     lastz T Q --entropy --format=general:... --markend --gfextend --chain
 *           --gapped --step=1 --strand=both --hspthresh=H
 * Every rule below that is taken from LASTZ's documentation rather than from the
 * reference tree is tagged [EXT]; DESIGN.md ("Alignment spec v1") lists where the
 * restatement fixes a choice LASTZ leaves to its implementation.
 *
 * The oracle is deliberately *sequential and LASTZ-shaped* (query scan in position
 * order, hash-table probe per word, one diagonal-extent array), i.e. a different
 * decomposition from the HIP engine (index join + stateless extension + chain
 * resolution), so that agreement between the two is evidence and not tautology.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SEED_LEN 19
#define SEED_WEIGHT 12
/* LASTZ default seed 12of19 = 1110100110010101111 [EXT] */
static const int CARE[SEED_WEIGHT] = {0, 1, 2, 4, 7, 8, 11, 13, 15, 16, 17, 18};

typedef struct {
    int32_t hspthresh, xdrop, ydrop, gap_open, gap_extend;
    int32_t transitions, entropy, chain, gapped, strand;
    int32_t reserved[6];
} orc_params; /* same layout as mimeo_params (include/mimeo_hip.h) */

typedef struct { uint32_t tpos, qpos; } orc_hit;
typedef struct {
    uint32_t tstart, qstart, length, flags;
    int64_t score, raw_score;
} orc_hsp;
typedef struct {
    uint32_t tid, qid, tstart, tend, qstart, qend;
    int64_t score;
    uint32_t id_n, id_d, qstrand, reserved;
} orc_aln;

/* ---- encoding ------------------------------------------------------------------ */
/* codes: A0 C1 G2 T3, 4 = anything else (scored as N).  low[] marks lower case. */
static void encode(const uint8_t *s, uint64_t n, uint8_t *code, uint8_t *low) {
    for (uint64_t i = 0; i < n; i++) {
        uint8_t c = s[i], l = 0;
        if (c >= 'a' && c <= 'z') { c = (uint8_t)(c - 32); l = 1; }
        uint8_t v = 4;
        if (c == 'A') v = 0; else if (c == 'C') v = 1; else if (c == 'G') v = 2; else if (c == 'T') v = 3;
        code[i] = v;
        if (low) low[i] = l;
    }
}
static void revcomp(uint8_t *code, uint64_t n) {
    for (uint64_t i = 0, j = n; i < j; i++) {
        j--;
        uint8_t a = code[i], b = code[j];
        a = a < 4 ? (uint8_t)(3 - a) : 4;
        b = b < 4 ? (uint8_t)(3 - b) : 4;
        code[i] = b; code[j] = a;
        if (i == j) code[i] = a;
    }
}

/* HOXD70 (LASTZ default) + fill_score -100 for N [EXT] */
static const int SUB[5][5] = {
    {91, -114, -31, -123, -100},
    {-114, 100, -125, -31, -100},
    {-31, -125, 100, -114, -100},
    {-123, -31, -114, 91, -100},
    {-100, -100, -100, -100, -100}};

/* ---- seed words ------------------------------------------------------------------ */
/* word = 2 bits per care position, interleaved (LASTZ-style packing); valid only if
 * the whole 19-base window is ACGT (and, for the target, upper case) [EXT]. */
static int word_at(const uint8_t *code, const uint8_t *low, uint64_t n, uint64_t p, uint32_t *w) {
    if (p + SEED_LEN > n) return 0;
    for (int k = 0; k < SEED_LEN; k++) {
        if (code[p + k] > 3) return 0;
        if (low && low[p + k]) return 0;
    }
    uint32_t v = 0;
    for (int j = 0; j < SEED_WEIGHT; j++) v |= (uint32_t)code[p + CARE[j]] << (2 * j);
    *w = v;
    return 1;
}

typedef struct {
    uint32_t *off; /* 2^24 + 1 */
    uint32_t *pos;
} seed_table;

static int build_table(const uint8_t *code, const uint8_t *low, uint64_t n, seed_table *t) {
    const uint32_t NB = 1u << 24;
    t->off = (uint32_t *)calloc((size_t)NB + 1, 4);
    t->pos = (uint32_t *)malloc((n ? n : 1) * 4);
    if (!t->off || !t->pos) return -1;
    uint32_t w;
    for (uint64_t p = 0; p + SEED_LEN <= n; p++)
        if (word_at(code, low, n, p, &w)) t->off[w + 1]++;
    for (uint32_t b = 0; b < NB; b++) t->off[b + 1] += t->off[b];
    uint32_t *cur = (uint32_t *)malloc((size_t)NB * 4);
    if (!cur) return -1;
    memcpy(cur, t->off, (size_t)NB * 4);
    for (uint64_t p = 0; p + SEED_LEN <= n; p++)
        if (word_at(code, low, n, p, &w)) t->pos[cur[w]++] = (uint32_t)p;
    free(cur);
    return 0;
}
static void free_table(seed_table *t) { free(t->off); free(t->pos); }

/* growable arrays */
#define VEC(T) struct { T *v; uint64_t n, cap; }
#define VPUSH(vec, T, x) do { if ((vec).n == (vec).cap) { (vec).cap = (vec).cap ? (vec).cap * 2 : 1024; \
    (vec).v = (T *)realloc((vec).v, (vec).cap * sizeof(T)); } (vec).v[(vec).n++] = (x); } while (0)

/* ---- gap-free extension (A8) --------------------------------------------------- */
/* Seed window [t, t+19) x [q, q+19).  Left walk starts at the seed END and runs left
 * (through the seed), right walk starts at the seed end and runs right; each keeps the
 * best prefix and stops after the step that leaves the running score more than xdrop
 * below its best [EXT: LASTZ xdrop_extend_seed_hit]. */
typedef struct { uint32_t tstart, qstart, length; int64_t score; uint32_t rext; } ext_result;

static ext_result extend_hit(const uint8_t *T, uint64_t Lt, const uint8_t *Q, uint64_t Lq,
                             uint32_t t, uint32_t q, int xdrop) {
    uint64_t et = (uint64_t)t + SEED_LEN, eq = (uint64_t)q + SEED_LEN;
    int64_t run = 0, best = 0;
    uint64_t bl = 0, maxl = et < eq ? et : eq;
    for (uint64_t k = 1; k <= maxl; k++) {
        run += SUB[T[et - k]][Q[eq - k]];
        if (run > best) { best = run; bl = k; }
        if (run < best - xdrop) break;
    }
    int64_t runr = 0, bestr = 0;
    uint64_t br = 0, maxr = (Lt - et) < (Lq - eq) ? (Lt - et) : (Lq - eq);
    for (uint64_t k = 0; k < maxr; k++) {
        runr += SUB[T[et + k]][Q[eq + k]];
        if (runr > bestr) { bestr = runr; br = k + 1; }
        if (runr < bestr - xdrop) break;
    }
    ext_result r;
    r.tstart = (uint32_t)(et - bl);
    r.qstart = (uint32_t)(eq - bl);
    r.length = (uint32_t)(bl + br);
    r.score = best + bestr;
    r.rext = (uint32_t)br;
    return r;
}

/* --entropy [EXT]: the HSP score is scaled by the base-4 entropy of the bases at the
 * HSP's identical columns; q is quantised to 16 fractional bits so that the CPU and
 * GPU log implementations cannot disagree (DESIGN.md spec v1 §4). */
static int64_t entropy_adjust(const uint8_t *T, const uint8_t *Q, uint32_t ts, uint32_t qs,
                              uint32_t len, int64_t score) {
    uint64_t c[4] = {0, 0, 0, 0}, n = 0;
    for (uint32_t i = 0; i < len; i++) {
        uint8_t a = T[ts + i];
        if (a < 4 && a == Q[qs + i]) { c[a]++; n++; }
    }
    double h = 0.0;
    if (n) {
        for (int b = 0; b < 4; b++)
            if (c[b]) { double p = (double)c[b] / (double)n; h -= p * log(p); }
        h /= log(4.0);
    }
    int64_t q16 = (int64_t)floor(h * 65536.0 + 0.5);
    if (q16 > 65536) q16 = 65536;
    if (q16 < 0) q16 = 0;
    return (score * q16) >> 16;
}

/* variants of a seed word: itself + one transition (flip of the high bit of a 2-bit
 * code: A<->G, C<->T) at each care position [EXT lastz default --transition]. */
static int variants(uint32_t w, int transitions, uint32_t *out) {
    int n = 0;
    out[n++] = w;
    if (transitions)
        for (int j = 0; j < SEED_WEIGHT; j++) out[n++] = w ^ (2u << (2 * j));
    return n;
}

typedef VEC(orc_hit) hitvec;
typedef VEC(orc_hsp) hspvec;

/* The LASTZ-shaped scan: query positions ascending, 13 probes each, per-diagonal
 * extent array suppressing hits inside an already-extended region. */
static int scan_pair_strand(const uint8_t *T, const uint8_t *Tlow, uint64_t Lt, const uint8_t *Q,
                            uint64_t Lq, const orc_params *p, hitvec *hits, hspvec *hsps) {
    seed_table tab;
    if (build_table(T, Tlow, Lt, &tab)) return -1;
    uint32_t *reach = NULL;
    if (hsps) {
        reach = (uint32_t *)calloc(Lt + Lq + 1, 4);
        if (!reach) return -1;
    }
    uint32_t var[16], w;
    for (uint64_t q = 0; q + SEED_LEN <= Lq; q++) {
        if (!word_at(Q, NULL, Lq, q, &w)) continue;
        int nv = variants(w, p->transitions, var);
        for (int v = 0; v < nv; v++) {
            for (uint32_t k = tab.off[var[v]]; k < tab.off[var[v] + 1]; k++) {
                uint32_t t = tab.pos[k];
                if (hits) { orc_hit h = {t, (uint32_t)q}; VPUSH(*hits, orc_hit, h); }
                if (!hsps) continue;
                uint64_t d = (uint64_t)t + Lq - q;
                uint32_t et = t + SEED_LEN;
                if (et <= reach[d]) continue; /* seed lies inside an extended region */
                ext_result r = extend_hit(T, Lt, Q, Lq, t, (uint32_t)q, p->xdrop);
                reach[d] = et + r.rext;
                if (r.score < p->hspthresh) continue;
                int64_t adj = r.score;
                if (p->entropy) {
                    adj = entropy_adjust(T, Q, r.tstart, r.qstart, r.length, r.score);
                    if (adj < p->hspthresh) continue;
                }
                orc_hsp h = {r.tstart, r.qstart, r.length, 0, adj, r.score};
                VPUSH(*hsps, orc_hsp, h);
            }
        }
    }
    free(reach);
    free_table(&tab);
    return 0;
}

static int cmp_hit(const void *a, const void *b) {
    const orc_hit *x = (const orc_hit *)a, *y = (const orc_hit *)b;
    if (x->tpos != y->tpos) return x->tpos < y->tpos ? -1 : 1;
    if (x->qpos != y->qpos) return x->qpos < y->qpos ? -1 : 1;
    return 0;
}
/* canonical HSP order: (diagonal, tstart) */
static int cmp_hsp_diag(const void *a, const void *b) {
    const orc_hsp *x = (const orc_hsp *)a, *y = (const orc_hsp *)b;
    int64_t dx = (int64_t)x->tstart - x->qstart, dy = (int64_t)y->tstart - y->qstart;
    if (dx != dy) return dx < dy ? -1 : 1;
    if (x->tstart != y->tstart) return x->tstart < y->tstart ? -1 : 1;
    return 0;
}

/* ---- chain (A9) -------------------------------------------------------------------- */
/* --chain with zero penalties [EXT]: the maximum-score subset of HSPs in which each
 * HSP ends at or before the start of the next in both sequences.  HSPs are taken in
 * (tstart, qstart, length) order; ties in the DP go to the earliest predecessor and
 * the earliest end (DESIGN.md spec v1 §5). */
static int cmp_hsp_chain(const void *a, const void *b) {
    const orc_hsp *x = (const orc_hsp *)a, *y = (const orc_hsp *)b;
    if (x->tstart != y->tstart) return x->tstart < y->tstart ? -1 : 1;
    if (x->qstart != y->qstart) return x->qstart < y->qstart ? -1 : 1;
    if (x->length != y->length) return x->length < y->length ? -1 : 1;
    return 0;
}
static int chain_hsps(orc_hsp *h, uint64_t n) {
    if (!n) return 0;
    qsort(h, n, sizeof(orc_hsp), cmp_hsp_chain);
    int64_t *best = (int64_t *)malloc(n * 8);
    int64_t *pred = (int64_t *)malloc(n * 8);
    if (!best || !pred) return -1;
    for (uint64_t j = 0; j < n; j++) {
        int64_t b = 0, pi = -1;
        for (uint64_t i = 0; i < j; i++) {
            if (h[i].tstart + h[i].length <= h[j].tstart && h[i].qstart + h[i].length <= h[j].qstart &&
                best[i] > b) { b = best[i]; pi = (int64_t)i; }
        }
        best[j] = b + h[j].score;
        pred[j] = pi;
    }
    uint64_t e = 0;
    for (uint64_t j = 1; j < n; j++) if (best[j] > best[e]) e = j;
    for (uint64_t j = 0; j < n; j++) h[j].flags &= ~1u;
    for (int64_t k = (int64_t)e; k >= 0; k = pred[k]) h[k].flags |= 1u;
    free(best); free(pred);
    return 0;
}

/* ---- anchors + gapped extension (A10) --------------------------------------------- */
/* anchor = centre of the best 31-column window of the HSP [EXT lastz anchor peak] */
static uint32_t anchor_offset(const uint8_t *T, const uint8_t *Q, const orc_hsp *h) {
    const uint32_t W = 31;
    if (h->length <= W) return h->length / 2;
    int64_t sum = 0, bestsum;
    uint32_t bestw = 0;
    for (uint32_t i = 0; i < W; i++) sum += SUB[T[h->tstart + i]][Q[h->qstart + i]];
    bestsum = sum;
    for (uint32_t w = 1; w + W <= h->length; w++) {
        sum += SUB[T[h->tstart + w + W - 1]][Q[h->qstart + w + W - 1]] - SUB[T[h->tstart + w - 1]][Q[h->qstart + w - 1]];
        if (sum > bestsum) { bestsum = sum; bestw = w; }
    }
    return bestw + W / 2;
}

#define NEG ((int64_t)-4000000000000000LL)
typedef struct { int64_t s; uint32_t nm, nx; } cell;

typedef struct { int64_t score; uint64_t i, j; uint32_t nm, nx; } half_result;

/* One-sided y-drop affine DP (DESIGN.md spec v1 §7).  dir = +1: A = T[at..), B = Q[aq..);
 * dir = -1: A = T[at-1], T[at-2], ...; B likewise.  Rows consume A, columns consume B.
 * A cell whose C score is below (best score of earlier rows) - ydrop is dead. */
static half_result half_extend(const uint8_t *T, uint64_t Lt, const uint8_t *Q, uint64_t Lq,
                               uint64_t at, uint64_t aq, int dir, const orc_params *p) {
    const int64_t O = p->gap_open, E = p->gap_extend, Y = p->ydrop;
    uint64_t lenA = dir > 0 ? Lt - at : at, lenB = dir > 0 ? Lq - aq : aq;
    half_result best = {0, 0, 0, 0, 0};
    uint64_t cap = 1024;
    cell *C0 = (cell *)malloc(cap * sizeof(cell)), *D0 = (cell *)malloc(cap * sizeof(cell));
    cell *C1 = (cell *)malloc(cap * sizeof(cell)), *D1 = (cell *)malloc(cap * sizeof(cell));
    /* row 0 */
    uint64_t lo = 0, hi = 0;
    C0[0].s = 0; C0[0].nm = C0[0].nx = 0; D0[0].s = NEG; D0[0].nm = D0[0].nx = 0;
    for (uint64_t j = 1; j <= lenB; j++) {
        int64_t v = -O - (int64_t)j * E;
        if (v < -Y) break;
        if (j >= cap) { cap *= 2; C0 = realloc(C0, cap * sizeof(cell)); D0 = realloc(D0, cap * sizeof(cell));
                        C1 = realloc(C1, cap * sizeof(cell)); D1 = realloc(D1, cap * sizeof(cell)); }
        C0[j].s = v; C0[j].nm = C0[j].nx = 0; D0[j].s = NEG; D0[j].nm = D0[j].nx = 0;
        hi = j;
    }
    /* prev row stored at index (j - plo) */
    uint64_t plo = lo, phi = hi;
    for (uint64_t i = 1; i <= lenA; i++) {
        int64_t thr = best.score - Y;
        uint8_t a = dir > 0 ? T[at + i - 1] : T[at - i];
        uint64_t jlo = plo, jmax = phi + 1;
        if (jmax > lenB) jmax = lenB;
        /* the row may run past jmax through an insertion chain; bound: while alive */
        cell Icell = {NEG, 0, 0};
        uint64_t first = UINT64_MAX, last = 0;
        int64_t rowbest = NEG; uint64_t rowbestj = 0; cell rowbestc = {NEG, 0, 0};
        uint64_t j = jlo;
        for (;; j++) {
            if (j > lenB) break;
            uint64_t idx = j - jlo;
            if (idx + 2 >= cap) { cap *= 2; C0 = realloc(C0, cap * sizeof(cell)); D0 = realloc(D0, cap * sizeof(cell));
                                  C1 = realloc(C1, cap * sizeof(cell)); D1 = realloc(D1, cap * sizeof(cell)); }
            cell d = {NEG, 0, 0}, g = {NEG, 0, 0};
            /* D: gap consuming A, from (i-1, j) */
            if (j >= plo && j <= phi) {
                cell pc = C0[j - plo], pd = D0[j - plo];
                if (pd.s > NEG) { d = pd; d.s -= E; }
                if (pc.s > NEG && pc.s - O - E > d.s) { d = pc; d.s = pc.s - O - E; }
            }
            /* diagonal from (i-1, j-1) */
            if (j >= 1 && j - 1 >= plo && j - 1 <= phi && C0[j - 1 - plo].s > NEG) {
                uint8_t b = dir > 0 ? Q[aq + j - 1] : Q[aq - j];
                g = C0[j - 1 - plo];
                g.s += SUB[a][b];
                if (a < 4 && a == b) g.nm++; else g.nx++;
            }
            cell h = g;                       /* prefer the diagonal on ties */
            if (d.s > h.s) h = d;
            cell c = h;
            if (Icell.s > c.s) c = Icell;     /* prefer H over I on ties */
            if (c.s < thr || c.s <= NEG / 2) { c.s = NEG; d.s = NEG; }
            else {
                if (first == UINT64_MAX) first = j;
                last = j;
                if (c.s > rowbest) { rowbest = c.s; rowbestj = j; rowbestc = c; }
            }
            C1[idx] = c; D1[idx] = d;
            /* I for the next column: extend vs open from this column's H (spec: unpruned h) */
            cell ni = {NEG, 0, 0};
            if (Icell.s > NEG) { ni = Icell; ni.s -= E; }
            if (h.s > NEG / 2 && h.s - O - E > ni.s) { ni = h; ni.s = h.s - O - E; }
            Icell = ni;
            if (j >= jmax && Icell.s < thr) { j++; break; }
        }
        if (first == UINT64_MAX) break;
        if (rowbest > best.score) { best.score = rowbest; best.i = i; best.j = rowbestj; best.nm = rowbestc.nm; best.nx = rowbestc.nx; }
        /* next row's previous = [first, last], re-based */
        uint64_t w = last - first + 1;
        memmove(C1, C1 + (first - jlo), w * sizeof(cell));
        memmove(D1, D1 + (first - jlo), w * sizeof(cell));
        cell *tc = C0; C0 = C1; C1 = tc;
        cell *td = D0; D0 = D1; D1 = td;
        plo = first; phi = last;
    }
    free(C0); free(D0); free(C1); free(D1);
    return best;
}

typedef VEC(orc_aln) alnvec;

static int cmp_hsp_score_desc(const void *a, const void *b) {
    const orc_hsp *x = (const orc_hsp *)a, *y = (const orc_hsp *)b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    return cmp_hsp_chain(a, b);
}

/* all stages for one (target, query-strand) */
static int align_pair_strand(const uint8_t *T, const uint8_t *Tlow, uint64_t Lt, const uint8_t *Q,
                             uint64_t Lq, int minus, const orc_params *p, alnvec *out) {
    hspvec hsps = {0, 0, 0};
    if (scan_pair_strand(T, Tlow, Lt, Q, Lq, p, NULL, &hsps)) return -1;
    if (p->chain) {
        if (chain_hsps(hsps.v, hsps.n)) return -1;
        uint64_t m = 0;
        for (uint64_t i = 0; i < hsps.n; i++) if (hsps.v[i].flags & 1u) hsps.v[m++] = hsps.v[i];
        hsps.n = m;
    }
    if (hsps.n) qsort(hsps.v, hsps.n, sizeof(orc_hsp), cmp_hsp_score_desc);
    uint64_t first_out = out->n;
    for (uint64_t k = 0; k < hsps.n; k++) {
        orc_hsp *h = &hsps.v[k];
        orc_aln a;
        memset(&a, 0, sizeof a);
        a.qstrand = (uint32_t)minus;
        if (!p->gapped) {
            a.tstart = h->tstart; a.tend = h->tstart + h->length;
            a.qstart = h->qstart; a.qend = h->qstart + h->length;
            a.score = h->score;
            for (uint32_t i = 0; i < h->length; i++) {
                uint8_t x = T[h->tstart + i], y = Q[h->qstart + i];
                if (x < 4 && x == y) a.id_n++;
                a.id_d++;
            }
        } else {
            uint32_t off = anchor_offset(T, Q, h);
            uint64_t at = (uint64_t)h->tstart + off, aq = (uint64_t)h->qstart + off;
            int inside = 0; /* anchor inside the box of an earlier alignment of this strand */
            for (uint64_t e = first_out; e < out->n && !inside; e++) {
                orc_aln *o = &out->v[e];
                if (at >= o->tstart && at < o->tend && aq >= o->qstart && aq < o->qend) inside = 1;
            }
            if (inside) continue;
            half_result L = half_extend(T, Lt, Q, Lq, at, aq, -1, p);
            half_result R = half_extend(T, Lt, Q, Lq, at, aq, +1, p);
            a.tstart = (uint32_t)(at - L.i); a.tend = (uint32_t)(at + R.i);
            a.qstart = (uint32_t)(aq - L.j); a.qend = (uint32_t)(aq + R.j);
            a.score = L.score + R.score;
            a.id_n = L.nm + R.nm;
            a.id_d = L.nm + R.nm + L.nx + R.nx;
        }
        VPUSH(*out, orc_aln, a);
    }
    /* drop alignments below the gapped threshold (they still blocked later anchors) and
     * convert minus-strand query coordinates to the plus strand (start2+/end2+). */
    uint64_t m = first_out;
    for (uint64_t e = first_out; e < out->n; e++) {
        orc_aln a = out->v[e];
        if (a.score < p->hspthresh) continue;
        if (minus) { uint32_t s = (uint32_t)(Lq - a.qend), t2 = (uint32_t)(Lq - a.qstart); a.qstart = s; a.qend = t2; }
        out->v[m++] = a;
    }
    out->n = m;
    free(hsps.v);
    return 0;
}

/* ---- exported API (ctypes) ----------------------------------------------------- */

void orc_free(void *p) { free(p); }

void orc_params_default(orc_params *p) {
    memset(p, 0, sizeof *p);
    p->hspthresh = 3000; p->xdrop = 910; p->ydrop = 9400; p->gap_open = 400; p->gap_extend = 30;
    p->transitions = 1; p->entropy = 1; p->chain = 1; p->gapped = 1; p->strand = 3;
}

static int prep(const uint8_t *Ta, uint64_t Lt, const uint8_t *Qa, uint64_t Lq, int minus,
                uint8_t **T, uint8_t **Tlow, uint8_t **Q) {
    *T = (uint8_t *)malloc(Lt + 1); *Tlow = (uint8_t *)malloc(Lt + 1); *Q = (uint8_t *)malloc(Lq + 1);
    if (!*T || !*Tlow || !*Q) return -1;
    encode(Ta, Lt, *T, *Tlow);
    encode(Qa, Lq, *Q, NULL);
    if (minus) revcomp(*Q, Lq);
    return 0;
}

/* A6+A7: every seed hit of one (target, query-strand); sorted by (tpos, qpos). */
int orc_seed_hits(const uint8_t *Ta, uint64_t Lt, const uint8_t *Qa, uint64_t Lq, int minus,
                  const orc_params *p, orc_hit **out, uint64_t *nout) {
    uint8_t *T, *Tlow, *Q;
    if (prep(Ta, Lt, Qa, Lq, minus, &T, &Tlow, &Q)) return -1;
    hitvec hv = {0, 0, 0};
    int rc = scan_pair_strand(T, Tlow, Lt, Q, Lq, p, &hv, NULL);
    if (!rc && hv.n) qsort(hv.v, hv.n, sizeof(orc_hit), cmp_hit);
    *out = hv.v; *nout = hv.n;
    free(T); free(Tlow); free(Q);
    return rc;
}

/* A8: HSPs of one (target, query-strand); sorted by (diagonal, tstart). */
int orc_ungapped_hsps(const uint8_t *Ta, uint64_t Lt, const uint8_t *Qa, uint64_t Lq, int minus,
                      const orc_params *p, orc_hsp **out, uint64_t *nout) {
    uint8_t *T, *Tlow, *Q;
    if (prep(Ta, Lt, Qa, Lq, minus, &T, &Tlow, &Q)) return -1;
    hspvec hv = {0, 0, 0};
    int rc = scan_pair_strand(T, Tlow, Lt, Q, Lq, p, NULL, &hv);
    if (!rc && p->chain) rc = chain_hsps(hv.v, hv.n);
    if (!rc && hv.n) qsort(hv.v, hv.n, sizeof(orc_hsp), cmp_hsp_diag);
    *out = hv.v; *nout = hv.n;
    free(T); free(Tlow); free(Q);
    return rc;
}

/* A6-A10: one `lastz T Q` run; rows in (strand, discovery) order. */
/* the chain stage alone on caller-made HSPs of one (target, query, strand): sorted in place, bit 0 of flags = chained */
int orc_chain_hsps(orc_hsp *h, uint64_t n) { return chain_hsps(h, n); }

int orc_align_pair(const uint8_t *Ta, uint64_t Lt, const uint8_t *Qa, uint64_t Lq,
                   const orc_params *p, orc_aln **out, uint64_t *nout) {
    alnvec av = {0, 0, 0};
    int rc = 0;
    for (int minus = 0; minus < 2 && !rc; minus++) {
        if (!(p->strand & (minus ? 2 : 1))) continue;
        uint8_t *T, *Tlow, *Q;
        if (prep(Ta, Lt, Qa, Lq, minus, &T, &Tlow, &Q)) return -1;
        rc = align_pair_strand(T, Tlow, Lt, Q, Lq, minus, p, &av);
        free(T); free(Tlow); free(Q);
    }
    *out = av.v; *nout = av.n;
    return rc;
}

/*
 * box_vs_path.c — a STUDY on top of the CPU oracle, not part of the parity checker.
 *
 * TEST INFRASTRUCTURE ONLY (see the header of mimeo_oracle.c).  PARITY UNPINNED.
 *
 * Alignment specification v1 (DESIGN.md §2 rule 6, mimeo_oracle.c:428-435, k6_gapped.hip) skips an anchor that
 * lies inside the (t, q) BOX of an earlier alignment of the same (pair, strand).  LASTZ's documented rule for
 * `--gapped` (reference call site src/mimeo/wrappers.py:1031) is to skip an anchor that lies ON THE PATH of an
 * earlier alignment [EXT].  This file restates the gapped stage with a traceback so that both rules can be run on
 * the same chained anchors and the difference counted (scripts/box_vs_path.py; the numbers are in DESIGN.md §2).
 *
 * The DP below is half_extend() of mimeo_oracle.c cell for cell (same recurrences, pruning and tie-breaks — the
 * study asserts that scores and end cells agree with it) with one byte of traceback per live cell.
 */
#include "mimeo_oracle.c"

typedef struct { uint64_t *key; uint64_t n, cap; } pathset;   /* diagonal columns of an alignment: t << 32 | q */

typedef struct {
    uint8_t *tb;        /* traceback bytes, row after row */
    uint64_t *row_off;  /* offset of row i in tb */
    uint64_t *row_lo;   /* column of its first byte */
    uint64_t ntb, captb, nrows, caprows;
} tbstore;

enum { TB_HD = 1, TB_CI = 2, TB_DOPEN = 4, TB_IOPEN = 8 };

static void tb_row(tbstore *s, uint64_t i, uint64_t lo) {
    if (i >= s->caprows) {
        s->caprows = s->caprows ? s->caprows * 2 : 1024;
        while (i >= s->caprows) s->caprows *= 2;
        s->row_off = (uint64_t *)realloc(s->row_off, s->caprows * 8);
        s->row_lo = (uint64_t *)realloc(s->row_lo, s->caprows * 8);
    }
    s->row_off[i] = s->ntb; s->row_lo[i] = lo; s->nrows = i + 1;
}
static void tb_push(tbstore *s, uint8_t b) {
    if (s->ntb == s->captb) { s->captb = s->captb ? s->captb * 2 : 1 << 16; s->tb = (uint8_t *)realloc(s->tb, s->captb); }
    s->tb[s->ntb++] = b;
}
static uint8_t tb_get(const tbstore *s, uint64_t i, uint64_t j) { return s->tb[s->row_off[i] + (j - s->row_lo[i])]; }

/* half_extend with traceback: returns the same half_result and appends the diagonal columns of the optimal
 * path (first best cell in (row, column) order, walked back to the origin) to `path`. */
static half_result half_extend_tb(const uint8_t *T, uint64_t Lt, const uint8_t *Q, uint64_t Lq, uint64_t at, uint64_t aq,
                                  int dir, const orc_params *p, pathset *path) {
    const int64_t O = p->gap_open, E = p->gap_extend, Y = p->ydrop;
    uint64_t lenA = dir > 0 ? Lt - at : at, lenB = dir > 0 ? Lq - aq : aq;
    half_result best = {0, 0, 0, 0, 0};
    uint64_t cap = 1024;
    cell *C0 = (cell *)malloc(cap * sizeof(cell)), *D0 = (cell *)malloc(cap * sizeof(cell));
    cell *C1 = (cell *)malloc(cap * sizeof(cell)), *D1 = (cell *)malloc(cap * sizeof(cell));
    tbstore S = {0, 0, 0, 0, 0, 0, 0};
    uint64_t lo = 0, hi = 0;
    C0[0].s = 0; C0[0].nm = C0[0].nx = 0; D0[0].s = NEG; D0[0].nm = D0[0].nx = 0;
    tb_row(&S, 0, 0);
    tb_push(&S, 0);
    for (uint64_t j = 1; j <= lenB; j++) {
        int64_t v = -O - (int64_t)j * E;
        if (v < -Y) break;
        if (j >= cap) { cap *= 2; C0 = realloc(C0, cap * sizeof(cell)); D0 = realloc(D0, cap * sizeof(cell));
                        C1 = realloc(C1, cap * sizeof(cell)); D1 = realloc(D1, cap * sizeof(cell)); }
        C0[j].s = v; C0[j].nm = C0[j].nx = 0; D0[j].s = NEG; D0[j].nm = D0[j].nx = 0;
        tb_push(&S, TB_CI);   /* row 0 beyond the origin: an insertion chain */
        hi = j;
    }
    uint64_t plo = lo, phi = hi;
    for (uint64_t i = 1; i <= lenA; i++) {
        int64_t thr = best.score - Y;
        uint8_t a = dir > 0 ? T[at + i - 1] : T[at - i];
        uint64_t jlo = plo, jmax = phi + 1;
        if (jmax > lenB) jmax = lenB;
        cell Icell = {NEG, 0, 0};
        int iopen = 0;
        uint64_t first = UINT64_MAX, last = 0;
        int64_t rowbest = NEG; uint64_t rowbestj = 0; cell rowbestc = {NEG, 0, 0};
        uint64_t j = jlo;
        tb_row(&S, i, jlo);
        for (;; j++) {
            if (j > lenB) break;
            uint64_t idx = j - jlo;
            if (idx + 2 >= cap) { cap *= 2; C0 = realloc(C0, cap * sizeof(cell)); D0 = realloc(D0, cap * sizeof(cell));
                                  C1 = realloc(C1, cap * sizeof(cell)); D1 = realloc(D1, cap * sizeof(cell)); }
            cell d = {NEG, 0, 0}, g = {NEG, 0, 0};
            uint8_t tb = iopen ? TB_IOPEN : 0;
            if (j >= plo && j <= phi) {
                cell pc = C0[j - plo], pd = D0[j - plo];
                if (pd.s > NEG) { d = pd; d.s -= E; }
                if (pc.s > NEG && pc.s - O - E > d.s) { d = pc; d.s = pc.s - O - E; tb |= TB_DOPEN; }
            }
            if (j >= 1 && j - 1 >= plo && j - 1 <= phi && C0[j - 1 - plo].s > NEG) {
                uint8_t b = dir > 0 ? Q[aq + j - 1] : Q[aq - j];
                g = C0[j - 1 - plo];
                g.s += SUB[a][b];
                if (a < 4 && a == b) g.nm++; else g.nx++;
            }
            cell h = g;
            if (d.s > h.s) { h = d; tb |= TB_HD; }
            cell c = h;
            if (Icell.s > c.s) { c = Icell; tb |= TB_CI; }
            if (c.s < thr || c.s <= NEG / 2) { c.s = NEG; d.s = NEG; }
            else {
                if (first == UINT64_MAX) first = j;
                last = j;
                if (c.s > rowbest) { rowbest = c.s; rowbestj = j; rowbestc = c; }
            }
            C1[idx] = c; D1[idx] = d;
            tb_push(&S, tb);
            cell ni = {NEG, 0, 0};
            iopen = 0;
            if (Icell.s > NEG) { ni = Icell; ni.s -= E; }
            if (h.s > NEG / 2 && h.s - O - E > ni.s) { ni = h; ni.s = h.s - O - E; iopen = 1; }
            Icell = ni;
            if (j >= jmax && Icell.s < thr) { j++; break; }
        }
        if (first == UINT64_MAX) break;
        if (rowbest > best.score) { best.score = rowbest; best.i = i; best.j = rowbestj; best.nm = rowbestc.nm; best.nx = rowbestc.nx; }
        uint64_t w = last - first + 1;
        memmove(C1, C1 + (first - jlo), w * sizeof(cell));
        memmove(D1, D1 + (first - jlo), w * sizeof(cell));
        cell *tc = C0; C0 = C1; C1 = tc;
        cell *td = D0; D0 = D1; D1 = td;
        plo = first; phi = last;
    }
    /* walk back from the best cell; states: 0 = C, 1 = H (max of diagonal and D), 2 = D, 3 = I */
    {
        uint64_t i = best.i, j = best.j;
        int st = 0;
        uint32_t nm = 0, nx = 0;
        while (i || j) {
            if (i == 0) { j--; continue; }   /* row 0: insertion chain back to the origin */
            const uint8_t tb = tb_get(&S, i, j);
            if (st == 0) st = (tb & TB_CI) ? 3 : 1;
            else if (st == 1) {
                if (tb & TB_HD) st = 2;
                else {
                    const uint64_t t = dir > 0 ? at + i - 1 : at - i, q = dir > 0 ? aq + j - 1 : aq - j;
                    if (path->n == path->cap) { path->cap = path->cap ? path->cap * 2 : 4096; path->key = (uint64_t *)realloc(path->key, path->cap * 8); }
                    path->key[path->n++] = (t << 32) | q;
                    if (T[t] < 4 && T[t] == Q[q]) nm++; else nx++;
                    i--; j--; st = 0;
                }
            } else if (st == 2) { st = (tb & TB_DOPEN) ? 0 : 2; i--; }
            else { st = (tb & TB_IOPEN) ? 1 : 3; j--; }
        }
        if (nm != best.nm || nx != best.nx) { fprintf(stderr, "box_vs_path: traceback disagrees with the carried counts (%u/%u vs %u/%u)\n", nm, nx, best.nm, best.nx); abort(); }
    }
    free(S.tb); free(S.row_off); free(S.row_lo);
    free(C0); free(D0); free(C1); free(D1);
    return best;
}

static int cmp_u64(const void *a, const void *b) {
    const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* counts[0] anchors (chained HSPs), [1] skipped by the rule in force, [2] anchors inside the box of an earlier alignment
 * but off every earlier path (only counted under the path rule: the anchors the two rules treat differently at that
 * point of the sequence), [3] alignments kept (>= threshold). */
static int align_pair_strand_rule(const uint8_t *T, const uint8_t *Tlow, uint64_t Lt, const uint8_t *Q, uint64_t Lq, int minus,
                                  const orc_params *p, int path_rule, alnvec *out, uint64_t *counts) {
    hspvec hsps = {0, 0, 0};
    if (scan_pair_strand(T, Tlow, Lt, Q, Lq, p, NULL, &hsps)) return -1;
    if (p->chain) {
        if (chain_hsps(hsps.v, hsps.n)) return -1;
        uint64_t m = 0;
        for (uint64_t i = 0; i < hsps.n; i++) if (hsps.v[i].flags & 1u) hsps.v[m++] = hsps.v[i];
        hsps.n = m;
    }
    if (hsps.n) qsort(hsps.v, hsps.n, sizeof(orc_hsp), cmp_hsp_score_desc);
    const uint64_t first_out = out->n;
    pathset all = {0, 0, 0};   /* diagonal columns of every alignment so far, kept sorted */
    for (uint64_t k = 0; k < hsps.n; k++) {
        orc_hsp *h = &hsps.v[k];
        orc_aln a;
        memset(&a, 0, sizeof a);
        a.qstrand = (uint32_t)minus;
        const uint32_t off = anchor_offset(T, Q, h);
        const uint64_t at = (uint64_t)h->tstart + off, aq = (uint64_t)h->qstart + off;
        counts[0]++;
        int inbox = 0, onpath = 0;
        for (uint64_t e = first_out; e < out->n && !inbox; e++) {
            orc_aln *o = &out->v[e];
            if (at >= o->tstart && at < o->tend && aq >= o->qstart && aq < o->qend) inbox = 1;
        }
        if (all.n) { const uint64_t key = (at << 32) | aq; onpath = bsearch(&key, all.key, all.n, 8, cmp_u64) != NULL; }
        if (path_rule && inbox && !onpath) counts[2]++;
        if (path_rule ? onpath : inbox) { counts[1]++; continue; }
        pathset mine = {0, 0, 0};
        half_result L = half_extend_tb(T, Lt, Q, Lq, at, aq, -1, p, &mine);
        half_result R = half_extend_tb(T, Lt, Q, Lq, at, aq, +1, p, &mine);
        {   /* the restated DP is the oracle's DP */
            half_result L0 = half_extend(T, Lt, Q, Lq, at, aq, -1, p), R0 = half_extend(T, Lt, Q, Lq, at, aq, +1, p);
            if (L0.score != L.score || L0.i != L.i || L0.j != L.j || R0.score != R.score || R0.i != R.i || R0.j != R.j) {
                fprintf(stderr, "box_vs_path: half_extend_tb disagrees with half_extend\n"); abort();
            }
        }
        a.tstart = (uint32_t)(at - L.i); a.tend = (uint32_t)(at + R.i);
        a.qstart = (uint32_t)(aq - L.j); a.qend = (uint32_t)(aq + R.j);
        a.score = L.score + R.score;
        a.id_n = L.nm + R.nm;
        a.id_d = L.nm + R.nm + L.nx + R.nx;
        VPUSH(*out, orc_aln, a);
        if (mine.n) {
            if (all.n + mine.n > all.cap) { all.cap = (all.n + mine.n) * 2; all.key = (uint64_t *)realloc(all.key, all.cap * 8); }
            memcpy(all.key + all.n, mine.key, mine.n * 8);
            all.n += mine.n;
            qsort(all.key, all.n, 8, cmp_u64);
        }
        free(mine.key);
    }
    uint64_t m = first_out;
    for (uint64_t e = first_out; e < out->n; e++) {
        orc_aln a = out->v[e];
        if (a.score < p->hspthresh) continue;
        if (minus) { uint32_t s = (uint32_t)(Lq - a.qend), t2 = (uint32_t)(Lq - a.qstart); a.qstart = s; a.qend = t2; }
        out->v[m++] = a;
        counts[3]++;
    }
    out->n = m;
    free(all.key);
    free(hsps.v);
    return 0;
}

/* one `lastz T Q` run under the box rule (path_rule = 0: must reproduce orc_align_pair) or the path rule (1);
 * strands as in p->strand. */
int orc_align_pair_rule(const uint8_t *Ta, uint64_t Lt, const uint8_t *Qa, uint64_t Lq, const orc_params *p, int path_rule,
                        orc_aln **out, uint64_t *nout, uint64_t *counts) {
    alnvec av = {0, 0, 0};
    int rc = 0;
    for (int k = 0; k < 4; k++) counts[k] = 0;
    for (int minus = 0; minus < 2 && !rc; minus++) {
        if (!(p->strand & (minus ? 2 : 1))) continue;
        uint8_t *T, *Tlow, *Q;
        if (prep(Ta, Lt, Qa, Lq, minus, &T, &Tlow, &Q)) return -1;
        rc = align_pair_strand_rule(T, Tlow, Lt, Q, Lq, minus, p, path_rule, &av, counts);
        free(T); free(Tlow); free(Q);
    }
    *out = av.v; *nout = av.n;
    return rc;
}

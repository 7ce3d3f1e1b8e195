"""Alignment specification v1 (rules 1-7) spelled out in plain Python, for small inputs only.

TEST INFRASTRUCTURE.  A second, independent restatement of the rules the C oracle (oracle/mimeo_oracle.c) and the HIP engine
implement: full matrices instead of bands, dictionaries instead of hash tables, no shared code with either.  It follows the
reference's call site src/mimeo/wrappers.py:1025-1037 (`lastz T Q --entropy --gfextend --chain --gapped --step=1 --strand=both
--hspthresh=H`) as far as LASTZ's documentation goes; PARITY UNPINNED like the oracle itself — the reference holds no fixture for
these stages — so what tests/test_oracle_rules.py establishes is that two restatements written apart agree, not that either is LASTZ.
"""
import math

SEED = '1110100110010101111'          # 12of19
CARE = [i for i, c in enumerate(SEED) if c == '1']
HOXD70 = {('A', 'A'): 91, ('A', 'C'): -114, ('A', 'G'): -31, ('A', 'T'): -123,
          ('C', 'A'): -114, ('C', 'C'): 100, ('C', 'G'): -125, ('C', 'T'): -31,
          ('G', 'A'): -31, ('G', 'C'): -125, ('G', 'G'): 100, ('G', 'T'): -114,
          ('T', 'A'): -123, ('T', 'C'): -31, ('T', 'G'): -114, ('T', 'T'): 91}
TRANSITION = {('A', 'G'), ('G', 'A'), ('C', 'T'), ('T', 'C')}
COMP = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A'}
NEG = -10 ** 15


def score(a, b):
    """rule 1: HOXD70; anything that is not ACGT scores -100 against everything"""
    return HOXD70.get((a, b), -100)


def upper(s):
    return ''.join(c.upper() for c in s)


def revcomp(s):
    return ''.join(COMP.get(c, 'N') for c in reversed(upper(s)))


def seed_hits(T, Q, transitions=True):
    """rule 2: every (t, q) whose 19-windows hold only ACGT (the target's in upper case) and agree at the twelve care
    positions, one transition at one care position allowed"""
    Tu, Qu = upper(T), upper(Q)
    okT = [all(c in 'ACGT' for c in T[t:t + 19]) for t in range(len(T) - 18)]     # lower case or N: no word
    okQ = [all(c in 'ACGT' for c in Qu[q:q + 19]) for q in range(len(Q) - 18)]
    words = {}
    for t in range(len(T) - 18):
        if okT[t]:
            words.setdefault(''.join(Tu[t + k] for k in CARE), []).append(t)
    hits = []
    for q in range(len(Q) - 18):
        if not okQ[q]:
            continue
        w = [Qu[q + k] for k in CARE]
        cands = [''.join(w)]
        if transitions:
            for j in range(12):
                v = list(w)
                v[j] = {'A': 'G', 'G': 'A', 'C': 'T', 'T': 'C'}[v[j]]
                cands.append(''.join(v))
        for c in cands:
            for t in words.get(c, ()):
                hits.append((t, q))
    return sorted(hits)


def extend(Tu, Qu, t, q, xdrop):
    """rule 3: both walks start at the seed's END; each keeps its first best prefix and stops after the step that leaves
    the running score more than xdrop below it"""
    et, eq = t + 19, q + 19
    run = best = 0
    left = 0
    for k in range(1, min(et, eq) + 1):
        run += score(Tu[et - k], Qu[eq - k])
        if run > best:
            best, left = run, k
        if run < best - xdrop:
            break
    runr = bestr = 0
    right = 0
    for k in range(min(len(Tu) - et, len(Qu) - eq)):
        runr += score(Tu[et + k], Qu[eq + k])
        if runr > bestr:
            bestr, right = runr, k + 1
        if runr < bestr - xdrop:
            break
    return et - left, eq - left, left + right, best + bestr, et + right


def entropy_q16(Tu, Qu, ts, qs, length):
    """rule 4: base-4 entropy of the bases at the identical columns, in 16 fractional bits"""
    counts = {}
    for i in range(length):
        a = Tu[ts + i]
        if a in 'ACGT' and a == Qu[qs + i]:
            counts[a] = counts.get(a, 0) + 1
    n = sum(counts.values())
    h = 0.0
    if n:
        for c in 'ACGT':
            if counts.get(c):
                p = counts[c] / n
                h -= p * math.log(p)
        h /= math.log(4.0)
    return max(0, min(65536, int(math.floor(h * 65536.0 + 0.5))))


def ungapped_hsps(T, Q, hspthresh=3000, xdrop=910, transitions=True, entropy=True):
    """rules 2-4 for one strand: (tstart, qstart, length, score, raw) of every HSP.  On a diagonal the hits are taken by
    increasing position; one whose seed ends at or before the right end of the last EXTENDED hit of the diagonal (kept or not)
    is skipped."""
    Tu, Qu = upper(T), upper(Q)
    reach = {}
    out = []
    for t, q in sorted(seed_hits(T, Q, transitions), key=lambda h: (h[0] - h[1], h[0])):
        d = t - q
        if t + 19 <= reach.get(d, 0):
            continue
        ts, qs, length, raw, rend = extend(Tu, Qu, t, q, xdrop)
        reach[d] = rend
        if raw < hspthresh:
            continue
        adj = raw
        if entropy:
            adj = (raw * entropy_q16(Tu, Qu, ts, qs, length)) >> 16
            if adj < hspthresh:
                continue
        out.append((ts, qs, length, adj, raw))
    return out


def chain(hsps):
    """rule 5: the chained subset; HSPs in (tstart, qstart, length) order"""
    s = sorted(hsps, key=lambda h: (h[0], h[1], h[2]))
    best, pred = [], []
    for j, h in enumerate(s):
        b, p = 0, -1
        for i in range(j):
            g = s[i]
            if g[0] + g[2] <= h[0] and g[1] + g[2] <= h[1] and best[i] > b:
                b, p = best[i], i
        best.append(b + h[3])
        pred.append(p)
    keep = []
    if s:
        k = max(range(len(s)), key=lambda i: (best[i], -i))   # first maximum
        while k >= 0:
            keep.append(s[k])
            k = pred[k]
    return keep


def anchor(Tu, Qu, h):
    """rule 6: centre of the first best 31-column window"""
    ts, qs, length = h[0], h[1], h[2]
    if length <= 31:
        return length // 2
    col = [score(Tu[ts + i], Qu[qs + i]) for i in range(length)]
    sums = [sum(col[w:w + 31]) for w in range(length - 30)]
    return sums.index(max(sums)) + 15


def half_extend(A, B, O, E, Y):
    """rule 7, one side: rows consume A, columns consume B; full matrices.  A cell is (score, matches, mismatches).
    Returns (score, rows, columns, matches, mismatches) of the first best cell in (row, column) order."""
    nb = len(B)
    dead = (NEG, 0, 0)
    C = [dead] * (nb + 1)
    D = [dead] * (nb + 1)
    C[0] = (0, 0, 0)
    for j in range(1, nb + 1):
        v = -O - j * E
        if v < -Y:
            break
        C[j] = (v, 0, 0)
    best = (0, 0, 0, 0, 0)
    for i in range(1, len(A) + 1):
        thr = best[0] - Y
        a = A[i - 1]
        C1, D1 = [dead] * (nb + 1), [dead] * (nb + 1)
        ins = dead          # the insertion state arriving at column j
        alive = False
        rowbest = None
        for j in range(nb + 1):
            d = dead
            if D[j][0] > NEG:
                d = (D[j][0] - E, D[j][1], D[j][2])
            if C[j][0] > NEG and C[j][0] - O - E > d[0]:          # extension preferred over opening on ties
                d = (C[j][0] - O - E, C[j][1], C[j][2])
            g = dead
            if j >= 1 and C[j - 1][0] > NEG:
                b = B[j - 1]
                m = a in 'ACGT' and a == b
                g = (C[j - 1][0] + score(a, b), C[j - 1][1] + (1 if m else 0), C[j - 1][2] + (0 if m else 1))
            h = d if d[0] > g[0] else g                               # diagonal preferred on ties
            c = ins if ins[0] > h[0] else h                           # ... then the column gap, then the row gap
            if c[0] < thr or c[0] <= NEG // 2:
                c, d = dead, dead
            else:
                alive = True
                if rowbest is None or c[0] > rowbest[0]:
                    rowbest = (c[0], i, j, c[1], c[2])
            C1[j], D1[j] = c, d
            ni = dead
            if ins[0] > NEG:
                ni = (ins[0] - E, ins[1], ins[2])
            if h[0] > NEG // 2 and h[0] - O - E > ni[0]:              # from the cell's H before pruning
                ni = (h[0] - O - E, h[1], h[2])
            ins = ni
        if not alive:
            break
        if rowbest[0] > best[0]:
            best = rowbest
        C, D = C1, D1
    return best


def align_strand(T, Q, minus, hspthresh=3000, xdrop=910, ydrop=9400, gap_open=400, gap_extend=30):
    """rules 2-7 for one strand of one pair: (tstart, tend, qstart, qend, score, matches, columns, strand), query coordinates
    on the plus strand"""
    Tu = upper(T)
    Qs = revcomp(Q) if minus else Q
    Qu = upper(Qs)
    hs = chain(ungapped_hsps(T, Qs, hspthresh, xdrop))
    hs.sort(key=lambda h: (-h[3], h[0], h[1], h[2]))
    made = []
    for h in hs:
        off = anchor(Tu, Qu, h)
        at, aq = h[0] + off, h[1] + off
        if any(a[0] <= at < a[1] and a[2] <= aq < a[3] for a in made):     # inside the box of an earlier alignment
            continue
        L = half_extend(Tu[:at][::-1], Qu[:aq][::-1], gap_open, gap_extend, ydrop)
        R = half_extend(Tu[at:], Qu[aq:], gap_open, gap_extend, ydrop)
        made.append((at - L[1], at + R[1], aq - L[2], aq + R[2], L[0] + R[0], L[3] + R[3], L[3] + R[3] + L[4] + R[4]))
    out = []
    for a in made:
        if a[4] < hspthresh:           # (it still blocked later anchors)
            continue
        qs, qe = (len(Q) - a[3], len(Q) - a[2]) if minus else (a[2], a[3])
        out.append((a[0], a[1], qs, qe, a[4], a[5], a[6], 1 if minus else 0))
    return out

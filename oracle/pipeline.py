"""Pure-Python restatement of the integer / text half of mimeo's hot path (stages A11-A16).

TEST INFRASTRUCTURE ONLY (see oracle/mimeo_oracle.c header).  Unlike the alignment half this
part IS pinned: tests/test_oracle_golden.py checks every function here against vectors made by
running the reference's own command text / pandas code (tests/golden/make_golden.py).

Each function cites the reference lines it restates (paths relative to /root/reference).
"""
import numpy as np


def general_rows(names_t, names_q, lens_q, alns):
    """lastz --format=general:name1,strand1,start1,end1,length1,name2,strand2,start2+,end2+,
    length2,score,identity rows (src/mimeo/wrappers.py:1031) from oracle/engine alignment
    records (0-based half-open, query on the plus strand).  start1/start2+ are origin-one,
    the identity field expands to 'n/d' and 'pct%' (SURVEY §8a A10)."""
    rows = []
    for a in alns:
        n, d = int(a['id_n']), int(a['id_d'])
        pct = '%.1f%%' % (100.0 * n / d) if d else '0.0%'
        rows.append('\t'.join(map(str, [
            names_t[int(a['tid'])], '+', int(a['tstart']) + 1, int(a['tend']),
            int(a['tend']) - int(a['tstart']),
            names_q[int(a['qid'])], '-' if int(a['qstrand']) else '+', int(a['qstart']) + 1, int(a['qend']),
            int(a['qend']) - int(a['qstart']), int(a['score']), '%d/%d' % (n, d), pct])))
    return rows


def _awk_num(s):
    """awk's ``0+$n``: longest numeric prefix, 0 if none."""
    import re
    m = re.match(r'\s*[-+]?(\d+\.?\d*([eE][-+]?\d+)?|\.\d+([eE][-+]?\d+)?)', s)
    return float(m.group(0)) if m else 0.0


def _sort_key_tab(line, numfield):
    f = line.split()
    return (f[0].encode(), _awk_num(f[numfield]), line.encode())


def filter_project_sort(general_text, min_len, min_idt):
    """wrappers.py:1040-1056: sed 's/%//g' ; awk '!/^#/' | awk '0+$5>=minLen' |
    awk -v OFS='\\t' '0+$13>=minIdt {print $1,$2,$3,$4,$6,$7,$8,$9,$11,$13}' | sed 's/ //g' |
    sort -k 1,1 -k 3n,4n   (C locale; last-resort whole-line byte order)."""
    out = []
    for line in general_text.replace('%', '').split('\n'):
        if not line or line.startswith('#'):
            continue
        f = line.split()
        if len(f) < 13:
            continue
        if _awk_num(f[4]) >= min_len and _awk_num(f[12]) >= min_idt:
            out.append('\t'.join([f[0], f[1], f[2], f[3], f[5], f[6], f[7], f[8], f[10], f[12]]).replace(' ', ''))
    out.sort(key=lambda l: _sort_key_tab(l, 2))
    return out


def bed_project_sort(tab_lines):
    """wrappers.py:1120-1128: awk '!/^#/ {print $1,$3,$4}' | sed 's/%//g' | sort -k 1,1 -k 2n,3n."""
    bed = []
    for line in tab_lines:
        if not line or line.startswith('#'):
            continue
        f = line.split()
        bed.append('\t'.join([f[0], f[2], f[3]]).replace('%', ''))
    bed.sort(key=lambda l: _sort_key_tab(l, 1))
    return bed


def coverage_collapse(intervals, chromlens, min_cov, min_len):
    """wrappers.py:1131-1150 + :1166-1167: bedtools genomecov -bg -g lens | awk '$4>=cov' | sort |
    bedtools merge | awk '$3-$2>=minLen', restated per base: depth[x] = number of half-open
    [start,end) intervals covering x (ends clipped to the chromosome length); keep maximal runs
    of depth >= min_cov (book-ended runs merge, bedtools merge -d 0) of length >= min_len.
    ``intervals`` = iterable of (chrom, start, end); ``chromlens`` = dict chrom -> length.
    Returns [(chrom, start, end)] sorted by (chrom bytes, start)."""
    per = {}
    for c, s, e in intervals:
        per.setdefault(c, []).append((int(s), int(e)))
    out = []
    for c in sorted(per, key=lambda x: x.encode() if isinstance(x, str) else x):
        L = int(chromlens[c])
        diff = np.zeros(L + 1, dtype=np.int64)
        for s, e in per[c]:
            e = min(e, L)
            if s >= e:
                continue
            diff[s] += 1
            diff[e] -= 1
        depth = np.cumsum(diff[:L])
        mask = depth >= max(1, min_cov)
        edges = np.flatnonzero(np.diff(np.concatenate(([0], mask.view(np.int8), [0]))))
        for s, e in zip(edges[0::2], edges[1::2]):
            if e - s >= min_len:
                out.append((c, int(s), int(e)))
    return out


def genomecov_bg(intervals, chromlens):
    """wrappers.py:1131-1140 `bedtools genomecov -bg -i sorted.bed -g lens` alone, restated per base: the maximal
    runs of equal depth > 0 (genomecov reports per-base depth, so a position where one interval ends and another
    begins does not split a run).  Returns [(chrom, start, end, depth)] in (chrom, start) order."""
    per = {}
    for c, s, e in intervals:
        per.setdefault(c, []).append((int(s), int(e)))
    out = []
    for c in sorted(per, key=lambda x: x.encode() if isinstance(x, str) else x):
        L = int(chromlens[c])
        diff = np.zeros(L + 1, dtype=np.int64)
        for s, e in per[c]:
            e = min(e, L)
            if s >= e:
                continue
            diff[s] += 1
            diff[e] -= 1
        depth = np.concatenate(([0], np.cumsum(diff[:L]), [0]))
        change = np.flatnonzero(np.diff(depth))      # depth[change + 1] holds from position `change` on
        for a, b in zip(change[:-1], change[1:]):
            d = int(depth[a + 1])
            if d > 0:
                out.append((c, int(a), int(b), d))
    return out


def gff_self_lines(regions, label, prefix, source='mimeo-self'):
    """wrappers.py:1153-1177 (x: :870-894, source 'mimeo'): header + one row per region, ID counter
    in output order, start written un-shifted."""
    lines = ['##gff-version 3', '#seqid\tsource\ttype\tstart\tend\tscore\tstrand\tphase\tattributes']
    for i, (c, s, e) in enumerate(regions, 1):
        lines.append('\t'.join([str(c), source, label, str(s), str(e), '.', '+', '.', 'ID=%s_%05d' % (prefix, i)]))
    return lines


def import_align(tab_text, prefix, min_len, min_idt):
    """wrappers.py:33-117 import_Align: re-filter int(end)-int(start) >= minLen and float(pID) >=
    minIdt, sort by the STRING columns (tName, tStart, tEnd, tStrand) — lexicographic, stable —
    then UID = prefix_<rank zero-padded to len(str(n))>."""
    hits = []
    for line in tab_text.split('\n'):
        li = line.strip()
        if not li or li.startswith('#'):
            continue
        f = li.split()
        if int(f[3]) - int(f[2]) >= min_len and float(f[9]) >= min_idt:
            hits.append(f[:10])
    order = sorted(range(len(hits)), key=lambda k: (hits[k][0], hits[k][2], hits[k][3], hits[k][1]))
    width = len(str(len(hits)))
    out = []
    for rank, k in enumerate(order, 1):
        out.append(hits[k] + ['%s_%s' % (prefix if prefix else 'BHit', str(rank).zfill(width))])
    return out


def gff_map_lines(rows, chromlens, ftype='BHit'):
    """wrappers.py:443-522 writeGFFlines."""
    lines = ['##gff-version 3']
    for name, ln in chromlens or []:
        lines.append(' '.join(['##sequence-region', str(name), '1', str(ln)]))
    lines.append('\t'.join(['##seqid', 'source', 'type', 'start', 'end', 'score', 'strand', 'phase', 'attributes']))
    for r in rows:
        attrs = ';'.join(['ID=' + r[10], 'identity=' + str(r[9]), 'B_locus=' + '_'.join([r[4], r[5], r[6], r[7]])])
        lines.append('\t'.join([r[0], 'mimeo-map', ftype, r[2], r[3], r[8], r[1], '.', attrs]))
    return lines


def tandem_masked(seq, start, end, match=2, mismatch=7, minscore=50, maxperiod=50, delta=7):
    """CPU restatement of the tandem scorer v2 that stands in for
    `trf F 2 7 7 80 10 50 50 -m -h -ngs` in wrappers.py:120-262.  PARITY UNPINNED: TRF itself is
    absent; this restates OUR specification, not TRF's heuristics.  For every period p the slice is
    aligned locally with itself on the diagonals p-b .. p+b (b = 0 for p = 1, 1 for p < 5, else 2;
    delta <= 0: b = 0, the gap-free scorer of round 1):
      H[j][d] = max(0, H[j-1][d] + s(S[j], S[j-d]), H[j-1][d-1] - delta, H[j][d+1] - delta)
    (ties: diagonal, insertion, deletion); a path that reaches a new best >= minscore masks everything
    from the first base of its earlier copy to the base it has reached.  seq: bytes; returns the
    number of masked bases of seq[start:end]."""
    s = seq[start:end].upper()
    L = len(s)
    masked = bytearray(L)
    ok = [c in b'ACGT' for c in s]
    for p in range(1, maxperiod + 1):
        b = (0 if p == 1 else (1 if p < 5 else 2)) if delta > 0 else 0
        nd, d0 = 2 * b + 1, p - b
        if L <= d0:
            continue
        prev = [(0, 0, 0, 0)] * nd          # (h, start, best, mend)
        for j in range((d0 // 32) * 32, L):
            cur = [(0, 0, 0, 0)] * nd
            for k in range(nd - 1, -1, -1):
                d = d0 + k
                c = (0, 0, 0, 0)
                if j >= d:
                    ph, pst, pbest, pmend = prev[k]
                    sc = match if (ok[j] and ok[j - d] and s[j] == s[j - d]) else -mismatch
                    c = (ph + sc, pst, pbest, pmend) if ph > 0 else (sc, j - d, 0, 0)
                    if k > 0 and prev[k - 1][0] - delta > c[0]:
                        q = prev[k - 1]
                        c = (q[0] - delta, q[1], q[2], q[3])
                    if k + 1 < nd and cur[k + 1][0] - delta > c[0]:
                        q = cur[k + 1]
                        c = (q[0] - delta, q[1], q[2], q[3])
                    if c[0] <= 0:
                        c = (0, 0, 0, 0)
                    elif c[0] > c[2]:
                        h, st, _, mend = c
                        if h >= minscore:
                            for x in range(max(mend, st), j + 1):
                                masked[x] = 1
                            mend = j + 1
                        c = (h, st, h, mend)
                cur[k] = c
            prev = cur
    return sum(masked)


def trf_filter(rows, seq_of, prefix=None, tmatch=2, tmismatch=7, tminscore=50, tmaxperiod=50, maxtandem=40, masked_fn=None, tdelta=7):
    """wrappers.py:120-262 trfFilter with `tandem_masked` in TRF's place: slice seq[int(tStart):int(tEnd)]
    of the origin-one start (wrappers.py:190), keep while masked / len * 100 < maxtandem (:237-240), then the
    string sort and renumbering of :243-259.  rows: import_align rows; seq_of: name -> bytes."""
    fn = masked_fn or tandem_masked
    keep = []
    for r in rows:
        s, e = int(r[2]), min(int(r[3]), len(seq_of[r[0]]))
        if e - s > 0 and fn(seq_of[r[0]], s, e, tmatch, tmismatch, tminscore, tmaxperiod, tdelta) / (e - s) * 100 < float(maxtandem):
            keep.append(r[:10])
    keep.sort(key=lambda f: (f[0], f[2], f[3], f[1]))
    width = len(str(len(keep)))
    pre = str(prefix) if prefix else 'BHit'
    return [f + ['%s_%s' % (pre, str(i).zfill(width))] for i, f in enumerate(keep, 1)]

"""Host-side text stages of the mimeo hot path: what run_jobs.sh does with sed/awk/sort around the
lastz and bedtools calls.  Every function cites the reference command text it reproduces (paths
relative to /root/reference/src/mimeo); the engine (HIP) supplies the numbers, these functions
supply the bytes.
"""
import os
import re

import numpy as np

TAB_HEADER = '#name1\tstrand1\tstart1\tend1\tname2\tstrand2\tstart2+\tend2+\tscore\tidentity'  # wrappers.py:996
GFF_HEADER = '##gff-version 3\n#seqid\tsource\ttype\tstart\tend\tscore\tstrand\tphase\tattributes'  # wrappers.py:1153


def read_fasta(path, headers=None):
    """FASTA -> (ids, [uint8 arrays]).  id = first word of the header, like Biopython's rec.id
    used by the reference (utils.py:307, :549).  `headers` (a list) receives the full header lines
    (Biopython's rec.description: what SeqIO.write puts behind '>')."""
    names, seqs, cur = [], [], None
    with open(path, 'rb') as f:
        data = f.read()
    for block in data.split(b'>')[1:]:
        nl = block.find(b'\n')
        header = block if nl < 0 else block[:nl]
        body = b'' if nl < 0 else block[nl + 1:]
        names.append(header.split()[0].decode() if header.split() else '')
        if headers is not None:
            headers.append(header.rstrip(b'\r').decode())
        seqs.append(np.frombuffer(body.translate(None, b'\n\r \t'), dtype=np.uint8))
    return names, seqs


def read_fasta_dir(dirname):
    """All records of all files in a directory, like chromlens/get_all_pairs glob the split
    directory (utils.py:92-102, :529-531); files are visited in sorted order."""
    names, seqs = [], []
    for fn in sorted(os.listdir(dirname)):
        p = os.path.join(dirname, fn)
        if os.path.isfile(p):
            n, s = read_fasta(p)
            names += n
            seqs += s
    return names, seqs


def check_unique(names):
    """utils.py:300-306 / :472-499: duplicate sequence ids are fatal."""
    seen = set()
    for n in names:
        if n in seen:
            raise SystemExit('Non-unique name in genome: %s. Quitting.' % n)
        seen.add(n)


def chromlens(names, seqs, outfile=None):
    """utils.py:502-557: (id, str(len)) sorted by id; optional `id\\tlen` file (bedtools -g).
    `seqs`: sequences or plain lengths."""
    lens = sorted(((n, str(s if isinstance(s, int) else len(s))) for n, s in zip(names, seqs)), key=lambda x: x[0])
    if outfile:
        with open(outfile, 'w') as f:
            for n, l in lens:
                f.write(n + '\t' + l + '\n')
    return lens


def identity_pct(n, d):
    """lastz prints identity as `n/d` and `%.1f%%`; the reference strips the % (wrappers.py:1040)
    and compares the printed one-decimal value with minIdt (wrappers.py:1052)."""
    return '%.1f' % (100.0 * n / d) if d else '0.0'


def printed_tenths(id_n, id_d):
    """The digits of `'%.1f' % (100 n / d)` as an integer (80.0 -> 800), element-wise: lastz prints identity `%.1f%%`, the
    reference strips the % (wrappers.py:1040) and awk compares the PRINTED one-decimal value with minIdt (wrappers.py:1052):
    format first, compare the formatted number.  '%.1f' rounds the double's exact value half-even: floor(10 v + 1/2) away
    from a tie, Python's own formatting within 1e-6 of one (10 v carries a rounding error of its own there)."""
    idn, idd = np.asarray(id_n, dtype=np.float64), np.asarray(id_d, dtype=np.float64)
    with np.errstate(divide='ignore', invalid='ignore'):
        val = np.where(idd > 0, 100.0 * idn / idd, 0.0)
    x10 = val * 10.0
    fl = np.floor(x10)
    frac = x10 - fl
    tenths = (fl + (frac > 0.5)).astype(np.int64)
    for i in np.flatnonzero(np.abs(frac - 0.5) < 1e-6).tolist():
        tenths[i] = int(('%.1f' % val[i]).replace('.', ''))
    return tenths


def tab_blocks(alns, tnames, qnames, min_len, min_idt):
    """Every pair's block of the 10-column TAB at once (wrappers.py:1043-1056, per pair:
    awk '0+$5 >= minLen' | awk '0+$13 >= minIdt {print $1,$2,$3,$4,$6,$7,$8,$9,$11,$13}' | sort -k 1,1 -k 3n,4n).
    `alns`: engine records of any number of (target, query) pairs; start1 and start2+ are origin-one (lastz general
    format), ends inclusive == half-open end.  Returns ({(tid, qid): [lines]}, kept) where `kept` is an (n, 4)
    int64 array of (tid, qid, start1, end1) of the rows written, in block order — what the BED projection of the
    file would read back (wrappers.py:1120-1128).  One pass over numpy columns instead of a Python loop per record:
    a C4 job has 6e5 alignments."""
    if alns.size == 0:
        return {}, np.zeros((0, 4), dtype=np.int64)
    ts, te = alns['tstart'].astype(np.int64), alns['tend'].astype(np.int64)
    tenths = printed_tenths(alns['id_n'], alns['id_d'])
    keep = (te - ts >= min_len) & (tenths >= int(min_idt) * 10 if float(min_idt) == int(min_idt) else tenths / 10.0 >= min_idt)   # length1 = end1 - start1 + 1 = te - ts
    idx = np.flatnonzero(keep)
    if idx.size == 0:
        return {}, np.zeros((0, 4), dtype=np.int64)
    tid, qid = alns['tid'].astype(np.int64)[idx], alns['qid'].astype(np.int64)[idx]
    pair = tid << 32 | qid
    order = np.lexsort((ts[idx], pair))          # name1 is constant inside a block: start1 numeric, then the whole line
    idx, tid, qid, pair = idx[order], tid[order], qid[order], pair[order]
    s1, e1 = ts[idx] + 1, te[idx]
    tn, qn = np.array(list(tnames), dtype=object)[tid].tolist(), np.array(list(qnames), dtype=object)[qid].tolist()
    sign = np.array(['+', '-'], dtype=object)[(alns['qstrand'][idx] != 0).astype(np.int64)].tolist()
    t10 = tenths[idx]
    lines = ['%s\t+\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t%d.%d' % r
             for r in zip(tn, s1.tolist(), e1.tolist(), qn, sign, (alns['qstart'].astype(np.int64)[idx] + 1).tolist(),
                          alns['qend'][idx].tolist(), alns['score'][idx].tolist(), (t10 // 10).tolist(), (t10 % 10).tolist())]
    # ties on (pair, start1) fall to sort's last-resort comparison of the whole line, byte by byte
    same = np.flatnonzero((pair[1:] == pair[:-1]) & (s1[1:] == s1[:-1]))
    if same.size:
        perm = np.arange(idx.size)
        run_start = same[np.r_[True, np.diff(same) > 1]]
        run_end = same[np.r_[np.diff(same) > 1, True]] + 2
        for a, b in zip(run_start.tolist(), run_end.tolist()):
            o = sorted(range(a, b), key=lambda k: lines[k].encode())
            lines[a:b] = [lines[k] for k in o]
            perm[a:b] = o
        idx, tid, qid, s1 = idx[perm], tid[perm], qid[perm], s1[perm]
    cuts = np.flatnonzero(np.diff(pair)) + 1
    blocks = {}
    for a, b in zip(np.r_[0, cuts].tolist(), np.r_[cuts, pair.size].tolist()):
        blocks[(int(tid[a]), int(qid[a]))] = lines[a:b]
    return blocks, np.stack([tid, qid, s1, te[idx]], axis=1)


def tab_block(alns, tname, qname, min_len, min_idt):
    """One pair's block (records of ONE (target, query) pair): see tab_blocks."""
    if alns.size == 0:
        return []
    one = alns.copy()
    one['tid'], one['qid'] = 0, 0
    blocks, _ = tab_blocks(one, [tname], [qname], min_len, min_idt)
    return blocks.get((0, 0), [])


def parse_tab(path):
    """Read a 10-column TAB (ours, or imported from another aligner: README.md:329-347)."""
    rows = []
    with open(path) as f:
        for line in f:
            if not line.strip() or line.startswith('#'):
                continue
            rows.append(line.split())
    return rows


def _awk_num(s):
    m = re.match(r'\s*[-+]?(\d+\.?\d*([eE][-+]?\d+)?|\.\d+([eE][-+]?\d+)?)', s)
    return float(m.group(0)) if m else 0.0


def bed_intervals(tab_rows, chrom_ids):
    """wrappers.py:1120-1128: awk '{print $1,$3,$4}' — BED start is the origin-one start1,
    un-shifted.  Returns an (n,3) uint32 array of (chrom id, start, end);
    rows on unknown chromosomes are dropped (bedtools genomecov would reject them)."""
    out = []
    for f in tab_rows:
        c = chrom_ids.get(f[0])
        if c is None:
            continue
        s = int(f[2]) if f[2].isdigit() else int(_awk_num(f[2]))   # awk's numeric reading of the field; plain digits need no regex
        e = int(f[3]) if f[3].isdigit() else int(_awk_num(f[3]))
        if s < 0 or e < 0:
            continue
        out.append((c, s, e))
    return np.array(out, dtype=np.uint32).reshape(-1, 3)


def gff_repeat_lines(regions, names_sorted, source, label, prefix):
    """wrappers.py:1166-1177 (self, source 'mimeo-self') / :883-894 (x, source 'mimeo'): one
    row per region in (chrom, start) order, ID = prefix_%05d counting from 1."""
    lines = []
    for i, r in enumerate(regions, 1):
        lines.append('\t'.join([names_sorted[int(r['chrom'])], source, label, str(int(r['start'])), str(int(r['end'])),
                                '.', '+', '.', 'ID=%s_%05d' % (prefix, i)]))
    return lines


def import_align(tab_rows, prefix=None, min_len=100, min_idt=95):
    """wrappers.py:33-117 import_Align: re-filter on int(end)-int(start) and float(pID); sort on
    the STRING columns tName, tStart, tEnd, tStrand (lexicographic, stable); UID = prefix_<rank>
    zero-padded to the width of the hit count.  Exits 1 when nothing passes (wrappers.py:94-96)."""
    hits = [f[:10] for f in tab_rows if int(f[3]) - int(f[2]) >= min_len and float(f[9]) >= min_idt]
    if not hits:
        raise SystemExit(1)
    hits.sort(key=lambda f: (f[0], f[2], f[3], f[1]))
    width = len(str(len(hits)))
    pre = str(prefix) if prefix else 'BHit'
    return [f + ['%s_%s' % (pre, str(i).zfill(width))] for i, f in enumerate(hits, 1)]


def gff_map_lines(rows, chrlens=None, ftype='BHit'):
    """wrappers.py:443-522 writeGFFlines (source mimeo-map, ##sequence-region per chromosome)."""
    yield '##gff-version 3\n'
    for name, maxlen in chrlens or []:
        yield ' '.join(['##sequence-region', str(name), '1', str(maxlen) + '\n'])
    yield '\t'.join(['##seqid', 'source', 'type', 'start', 'end', 'score', 'strand', 'phase', 'attributes' + '\n'])
    for r in rows:
        attributes = ';'.join(['ID=' + r[10], 'identity=' + str(r[9]), 'B_locus=' + r[4] + '_' + r[5] + '_' + str(r[6]) + '_' + str(r[7])])
        yield '\t'.join([r[0], 'mimeo-map', ftype, str(r[2]), str(r[3]), str(r[8]), r[1], '.', attributes + '\n'])


def renumber(rows, prefix=None):
    """UIDs after a filter (wrappers.py:243-259): same string sort as import_Align, then rank."""
    rows = sorted((r[:10] for r in rows), key=lambda f: (f[0], f[2], f[3], f[1]))
    width = len(str(len(rows)))
    pre = str(prefix) if prefix else 'BHit'
    return [f + ['%s_%s' % (pre, str(i).zfill(width))] for i, f in enumerate(rows, 1)]


def write_trf_tab(rows, outtab):
    """wrappers.py:380-440 writetrf: the filtered hits in the 10-column TAB layout, `<outtab>.trf`."""
    outfile = outtab + '.trf'
    with open(outfile, 'w') as f:
        f.write('\t'.join(['#name1', 'strand1', 'start1', 'end1', 'name2', 'strand2', 'start2+', 'end2+', 'score',
                           'identity']) + '\n')
        for r in rows:
            f.write('\t'.join(r[:10]) + '\n')
    return outfile

#!/usr/bin/env python3
"""`bedtools`-compatible command line over the HIP engine, beside `lastz_shim`: with both, the
unmodified reference runs end to end on the GPU through its own `--lzpath` / `--bedtools` options
(src/mimeo/run_self.py:117-128).  Only the two invocations the reference emits are supported
(src/mimeo/wrappers.py:1131-1150):

    bedtools genomecov -bg -i SORTED.bed -g CHROMLENS     depth runs (chrom, start, end, depth), depth > 0
    bedtools merge -i SORTED.bed                          overlapping and book-ended intervals joined

Both run on the device (K7: events, radix sort, scan — mimeo_coverage_bedgraph / mimeo_coverage_collapse); the
output goes to stdout, chromosomes in the order the input names them (the input is sorted: that is bedtools' order
too).  Anything else is refused rather than ignored.
"""
import sys

import numpy as np

from . import engine


def read_bed(path):
    """-> chromosome names in order of first appearance, (n, 3) array of (chrom index, start, end)"""
    names, index, rows = [], {}, []
    with open(path) as f:
        for line in f:
            if not line.strip() or line.startswith(('#', 'track', 'browser')):
                continue
            p = line.rstrip('\n').split('\t')
            if len(p) < 3:
                raise SystemExit('bedtools_shim: malformed BED line: %r' % line)
            if p[0] not in index:
                index[p[0]] = len(names)
                names.append(p[0])
            rows.append((index[p[0]], int(p[1]), int(p[2])))
    return names, np.asarray(rows, dtype=np.int64).reshape(-1, 3)


def genomecov(bed, genome, out):
    names, iv = read_bed(bed)
    lens = {}
    with open(genome) as f:
        for line in f:
            p = line.split()
            if len(p) >= 2:
                lens[p[0]] = int(p[1])
    missing = [n for n in names if n not in lens]
    if missing:
        raise SystemExit('bedtools_shim: chromosome %s of %s is not in the genome file %s' % (missing[0], bed, genome))
    if not iv.size:
        return
    engine.init(0)
    runs = engine.coverage_bedgraph(iv, [lens[n] for n in names])
    for r in runs:
        out.write('%s\t%d\t%d\t%d\n' % (names[int(r['chrom'])], int(r['start']), int(r['end']), int(r['depth'])))


def merge(bed, out):
    names, iv = read_bed(bed)
    if not iv.size:
        return
    engine.init(0)
    ends = np.zeros(len(names), dtype=np.int64)
    np.maximum.at(ends, iv[:, 0], iv[:, 2])
    regions = engine.coverage_collapse(iv, ends, 1, 0)   # depth >= 1, no length filter: the union, book-ended runs joined
    for r in regions:
        out.write('%s\t%d\t%d\n' % (names[int(r['chrom'])], int(r['start']), int(r['end'])))


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit('bedtools_shim: expected a sub-command (genomecov or merge)')
    cmd, rest = argv[0], argv[1:]
    opts, flags, k = {}, set(), 0
    while k < len(rest):
        if rest[k] in ('-i', '-g'):
            if k + 1 >= len(rest):
                raise SystemExit('bedtools_shim: %s needs a file' % rest[k])
            opts[rest[k]] = rest[k + 1]
            k += 2
        elif rest[k] == '-bg':
            flags.add('-bg')
            k += 1
        else:
            raise SystemExit('bedtools_shim: unsupported argument %s (only the invocations mimeo emits are implemented)' % rest[k])
    if cmd == 'genomecov':
        if '-bg' not in flags or '-i' not in opts or '-g' not in opts:
            raise SystemExit('bedtools_shim: only `genomecov -bg -i BED -g GENOME` is supported')
        genomecov(opts['-i'], opts['-g'], sys.stdout)
    elif cmd == 'merge':
        if '-i' not in opts or flags or '-g' in opts:
            raise SystemExit('bedtools_shim: only `merge -i BED` is supported')
        merge(opts['-i'], sys.stdout)
    else:
        raise SystemExit('bedtools_shim: unsupported sub-command %s' % cmd)
    return 0


if __name__ == '__main__':
    sys.exit(main())

#!/usr/bin/env python3
"""`lastz`-compatible command line over the HIP engine.

The unmodified reference can drive the GPU engine through its own `--lzpath` option
(src/mimeo/run_self.py:117-122): point it at a wrapper that runs `python -m mimeo_amd.lastz_shim`.
Only the invocation the reference emits is supported (src/mimeo/wrappers.py:1025-1037, 786-798, 645-653):

    lastz TARGET QUERY --entropy --format=general:name1,strand1,start1,end1,length1,name2,strand2,
          start2+,end2+,length2,score,identity --markend --gfextend --chain --gapped --step=1
          --strand=both --hspthresh=H --output=FILE --verbosity=V

Flags map onto mimeo_params; anything else is refused rather than ignored.  Output rows are
lastz's `general` format: 12 requested fields, `identity` expanding to `n/d` and `pct%`.
"""
import sys

from . import engine, formats

GENERAL = 'general:name1,strand1,start1,end1,length1,name2,strand2,start2+,end2+,length2,score,identity'


def parse(argv):
    files, opt = [], {'entropy': 0, 'chain': 0, 'gapped': 1, 'gfextend': 1, 'strand': 'both', 'hspthresh': 3000,
                      'step': 1, 'output': None, 'markend': False, 'format': None}
    for a in argv:
        if not a.startswith('--'):
            files.append(a)
            continue
        k, _, v = a[2:].partition('=')
        if k in ('entropy', 'chain', 'gapped', 'gfextend', 'markend'):
            opt[k] = True if k == 'markend' else 1
        elif k in ('noentropy', 'nochain', 'nogapped'):
            opt[k[2:]] = 0
        elif k == 'strand':
            if v not in ('both', 'plus', 'minus'):
                raise SystemExit('lastz_shim: bad --strand=%s' % v)
            opt['strand'] = v
        elif k in ('hspthresh', 'step', 'verbosity'):
            opt[k] = int(v)
        elif k in ('output', 'format'):
            opt[k] = v
        else:
            raise SystemExit('lastz_shim: unsupported option --%s (only the flags mimeo passes are implemented)' % k)
    if len(files) != 2:
        raise SystemExit('lastz_shim: expected TARGET and QUERY files')
    if opt['format'] != GENERAL:
        raise SystemExit('lastz_shim: only --format=%s is supported' % GENERAL)
    if opt['step'] != 1 or not opt['gfextend']:
        raise SystemExit('lastz_shim: only --step=1 --gfextend is supported')
    return files, opt


def general_rows(alns, tnames, qnames):
    rows = []
    for a in alns:
        n, d = int(a['id_n']), int(a['id_d'])
        ts, te, qs, qe = int(a['tstart']), int(a['tend']), int(a['qstart']), int(a['qend'])
        rows.append('\t'.join(map(str, [tnames[int(a['tid'])], '+', ts + 1, te, te - ts, qnames[int(a['qid'])],
                                        '-' if int(a['qstrand']) else '+', qs + 1, qe, qe - qs, int(a['score']),
                                        '%d/%d' % (n, d), formats.identity_pct(n, d) + '%'])))
    return rows


def main(argv=None):
    files, opt = parse(sys.argv[1:] if argv is None else argv)
    engine.init(0)
    T, Q = engine.Genome.from_fasta(files[0]), engine.Genome.from_fasta(files[1])
    tn, qn = T.names, Q.names
    p = engine.default_params(hspthresh=opt['hspthresh'], entropy=opt['entropy'], chain=opt['chain'], gapped=opt['gapped'],
                              strand={'both': 3, 'plus': 1, 'minus': 2}[opt['strand']])
    alns = engine.align_pairs(T, Q, [(t, q) for t in range(len(tn)) for q in range(len(qn))], p)
    out = open(opt['output'], 'w') if opt['output'] else sys.stdout
    out.write('#' + GENERAL.split(':', 1)[1].replace(',', '\t').replace('identity', 'identity\tidPct') + '\n')
    for r in general_rows(alns, tn, qn):
        out.write(r + '\n')
    if opt['markend']:
        out.write('# lastz end-of-file\n')
    if out is not sys.stdout:
        out.close()
    T.close()
    Q.close()


if __name__ == '__main__':
    main()

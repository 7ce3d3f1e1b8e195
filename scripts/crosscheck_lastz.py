#!/usr/bin/env python3
"""A/B cross-check against a REAL lastz binary, if one is on PATH.

The alignment stages A6-A10 of this repo restate LASTZ's documented behaviour and are PARITY UNPINNED:
the reference (Adamtaranto/mimeo) holds no LASTZ output and the build image has no lastz.  This script is
what turns a lastz that IS present into parity evidence: it writes small synthetic FASTA pairs, runs the
literal command line the reference builds (src/mimeo/wrappers.py:1025-1037) with the real binary and with
this repo's drop-in (`python -m mimeo_amd.lastz_shim`, same argv), and diffs the 13-column `general` rows.

    python scripts/crosscheck_lastz.py [--json] [--cases N] [--lzpath lastz]

Without a lastz binary it prints "lastz absent: parity unpinned" and exits 0 (nothing is run, no GPU is
touched).  bench.py calls `probe()` and records which of the two happened in its JSON line.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FORMAT = 'general:name1,strand1,start1,end1,length1,name2,strand2,start2+,end2+,length2,score,identity'


def reference_argv(target_fa, query_fa, out, hspthresh=3000):
    """The flags of wrappers.py:1025-1037, in the reference's order."""
    return [target_fa, query_fa, '--entropy', '--format=' + FORMAT, '--markend', '--gfextend', '--chain', '--gapped',
            '--step=1', '--strand=both', '--hspthresh=%d' % hspthresh, '--output=' + out, '--verbosity=0']


def _is_our_shim(path):
    """scripts/lastz and scripts/bedtools of THIS repo are drop-ins that call the engine: a binary found under the repo, or a
    script that starts mimeo_amd's shims, is not the reference's tool — the A/B check would compare the engine with itself
    and report false parity evidence."""
    if not path:
        return False
    real = os.path.realpath(path)
    if real == ROOT or real.startswith(ROOT + os.sep):
        return True
    try:
        with open(real, 'rb') as f:
            head = f.read(4096)
    except OSError:
        return False
    return head.startswith(b'#!') and (b'mimeo_amd.lastz_shim' in head or b'mimeo_amd.bedtools_shim' in head or b'mimeo_amd' in head)


def probe(lzpath='lastz'):
    """{'lastz': path or None, 'bedtools': ..., 'trf': ..., 'status': ...} — no process is started.  This repo's own
    drop-ins (scripts/lastz, scripts/bedtools) do not count as the reference's tools."""
    found = {tool: shutil.which(path) for tool, path in (('lastz', lzpath), ('bedtools', 'bedtools'), ('trf', 'trf'))}
    for tool in ('lastz', 'bedtools'):
        if _is_our_shim(found[tool]):
            found[tool] = None
    found['status'] = 'lastz present: run scripts/crosscheck_lastz.py for the A/B diff' if found['lastz'] else 'lastz absent: parity unpinned'
    return found


def rows_of(path):
    with open(path) as f:
        return [l.rstrip('\n') for l in f if l.strip() and not l.startswith('#')]


def compare(real, ours):
    """Row-level diff of two `general` outputs: identical rows, rows whose coordinates agree but whose score or
    identity differ, and rows only one side has."""
    key = lambda r: tuple(r.split('\t')[i] for i in (0, 2, 3, 5, 6, 7, 8))
    a, b = {key(r): r for r in real}, {key(r): r for r in ours}
    same = sum(1 for k in a if k in b and a[k] == b[k])
    coords_only = sum(1 for k in a if k in b and a[k] != b[k])
    return {'rows_lastz': len(real), 'rows_ours': len(ours), 'identical': same, 'same_coordinates_other_score': coords_only,
            'only_lastz': len([k for k in a if k not in b]), 'only_ours': len([k for k in b if k not in a])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lzpath', default='lastz')
    ap.add_argument('--cases', type=int, default=3)
    ap.add_argument('--json', action='store_true')
    args = ap.parse_args()
    found = probe(args.lzpath)
    if not found['lastz']:
        print(json.dumps(found) if args.json else found['status'])
        return 0
    from mimeo_amd.synth import synth_genome, write_fasta
    report = {'lastz': found['lastz'], 'cases': []}
    with tempfile.TemporaryDirectory() as tmp:
        for c in range(args.cases):
            names, seqs = synth_genome(900 + c, 200_000, 2, repeat_frac=0.15, families=4, cons_len=(300, 2000))
            ta, qa = os.path.join(tmp, 't%d.fa' % c), os.path.join(tmp, 'q%d.fa' % c)
            write_fasta(ta, names[:1], seqs[:1])
            write_fasta(qa, names[1:], seqs[1:])
            o_real, o_ours = os.path.join(tmp, 'real%d.tab' % c), os.path.join(tmp, 'ours%d.tab' % c)
            subprocess.check_call([found['lastz']] + reference_argv(ta, qa, o_real))
            subprocess.check_call([sys.executable, '-m', 'mimeo_amd.lastz_shim'] + reference_argv(ta, qa, o_ours), cwd=ROOT)
            report['cases'].append(compare(rows_of(o_real), rows_of(o_ours)))
    tot = {k: sum(c[k] for c in report['cases']) for k in report['cases'][0]}
    report['total'] = tot
    report['status'] = ('lastz present: %d of %d lastz rows reproduced exactly, %d with the same coordinates and another score, '
                        '%d only in lastz, %d only here' % (tot['identical'], tot['rows_lastz'], tot['same_coordinates_other_score'],
                                                            tot['only_lastz'], tot['only_ours']))
    print(json.dumps(report) if args.json else report['status'])
    return 0


if __name__ == '__main__':
    sys.exit(main())

#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference itself.

Run in the build container only (needs /root/reference and real awk/sed/sort/bash):

    python tests/golden/make_golden.py

What is pinned, and how:

* The reference's command builders ``mimeo.wrappers.self_LZ_cmds / xspecies_LZ_cmds /
  map_LZ_cmds`` are imported from /root/reference/src and *executed*; their command text
  is not stored.  ``mimeo.wrappers`` does ``from Bio import SeqIO`` at import time and
  Biopython is not installed; none of the functions used here touches it, so an empty
  placeholder module named ``Bio`` is put in ``sys.modules`` for the import line only.
* ``filter_*.json`` — stage A11/A12 (sed|awk|awk|awk|sed|sort, BED projection + sort):
  the reference's own command list is executed by the reference's own
  ``mimeo.utils.run_cmd`` with ``lzpath`` pointing at a stand-in *script written here*
  that only copies a prepared synthetic ``--format=general`` file to ``--output=``
  (LASTZ itself is absent: alignment arithmetic stays "parity unpinned").  The command
  list is cut before the first ``bedtools`` line (bedtools is absent too).
* ``gff_self.json`` / ``gff_x.json`` — stage A14: the reference's last two commands (GFF
  header echo + minLen/awk formatter) run on a prepared merged ``temp.bed``.
* ``map_import.json`` — stage A16: the reference's ``import_Align`` + ``writeGFFlines``
  (pandas) run on a synthetic 10-column TAB.
* ``collapse_kat.json`` is NOT produced here: it is hand-derived from the documented
  semantics of ``bedtools genomecov -bg`` / ``bedtools merge``.

Only inputs and outputs are stored; no reference source text is copied.
"""
import json
import os
import random
import shutil
import stat
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = '/root/reference/src'


def import_reference():
    if 'Bio' not in sys.modules:
        try:
            import Bio  # noqa: F401
        except ImportError:
            placeholder = types.ModuleType('Bio')
            placeholder.SeqIO = None  # never dereferenced by the functions used below
            sys.modules['Bio'] = placeholder
    sys.path.insert(0, REF_SRC)
    import mimeo.utils as U
    import mimeo.wrappers as W
    return U, W


def synth_general_rows(rng, tname, qname, n):
    """Synthetic 13-field lastz general rows (text), with edge cases at the thresholds."""
    rows = []
    specials = [
        (100, 1600, 2000),  # 80.0% exactly, length1 == 100
        (99, 1999, 2000),   # length1 below minLen
        (250, 1599, 2000),  # 79.95 -> printed 80.0 or 79.9 depending on the double
        (250, 799, 1000),   # 79.9
        (400, 1, 1),        # 100.0
    ]
    for k in range(n):
        if k < len(specials):
            length1, idn, idd = specials[k]
        else:
            length1 = rng.choice([50, 99, 100, 101, 150, 1000, 12345])
            idd = rng.randint(max(10, length1 - 20), length1 + 5)
            idn = rng.randint(int(idd * 0.7), idd)
        start1 = rng.choice([1, 5, 5, 77, 1000, 1000, 99999, rng.randint(1, 200000)])
        end1 = start1 + length1 - 1
        strand2 = rng.choice('+-')
        start2 = rng.randint(1, 300000)
        length2 = length1 + rng.randint(-3, 3)
        end2 = start2 + max(1, length2) - 1
        score = rng.randint(3000, 900000)
        pct = '%.1f%%' % (100.0 * idn / idd)
        rows.append('\t'.join(map(str, [
            tname, '+', start1, end1, length1, qname, strand2, start2, end2, max(1, length2),
            score, '%d/%d' % (idn, idd), pct])))
    return rows


FAKE_LASTZ = r'''#!/bin/bash
# Stand-in used only by tests/golden/make_golden.py: emits a prepared synthetic
# --format=general file for the (target, query) pair named on the command line.
t=$(basename "$1" .fa); q=$(basename "$2" .fa); out=""
for a in "$@"; do case "$a" in --output=*) out="${a#--output=}";; esac; done
cp "$GOLDEN_ROWS_DIR/${q}_onto_${t}.general" "$out"
'''


def run_reference_cmds(U, cmds, workdir):
    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        U.run_cmd(cmds, keeptemp=True)
    finally:
        os.chdir(cwd)
    tmpdirs = [d for d in os.listdir(workdir) if d.startswith('tmp.')]
    assert len(tmpdirs) == 1
    return os.path.join(workdir, tmpdirs[0])


def main():
    os.environ['LC_ALL'] = 'C'
    U, W = import_reference()
    rng = random.Random(20250523)
    work = tempfile.mkdtemp(prefix='golden_')
    rows_dir = os.path.join(work, 'rows')
    os.makedirs(rows_dir)
    os.environ['GOLDEN_ROWS_DIR'] = rows_dir
    fake = os.path.join(work, 'fake_lastz.sh')
    with open(fake, 'w') as f:
        f.write(FAKE_LASTZ)
    os.chmod(fake, os.stat(fake).st_mode | stat.S_IEXEC)

    names = ['s1', 's10', 's2']
    gdir = os.path.join(work, 'g')
    os.makedirs(gdir)
    pairs = [(os.path.join(gdir, a + '.fa'), os.path.join(gdir, b + '.fa')) for a in names for b in names]
    general = {}
    for a in names:
        for b in names:
            rows = synth_general_rows(rng, a, b, 14)
            # lastz general output carries a '#' header line; keep one to test the !/^#/ filter
            text = '#name1\tstrand1\tstart1\tend1\tlength1\tname2\tstrand2\tstart2+\tend2+\tlength2\tscore\tidentity\tidPct\n'
            text += '\n'.join(rows) + '\n# lastz end-of-file\n'
            general['%s_onto_%s' % (b, a)] = text
            with open(os.path.join(rows_dir, '%s_onto_%s.general' % (b, a)), 'w') as f:
                f.write(text)

    lens_path = os.path.join(work, 'A_gen_lens.txt')
    chromlens = [('s1', 250000), ('s10', 250000), ('s2', 250000)]
    with open(lens_path, 'w') as f:
        for n, l in chromlens:
            f.write('%s\t%d\n' % (n, l))

    # ---- A11 (+A12) through the reference's own pipeline ---------------------------------
    cases = []
    for mode, minIdt, minLen, strict in [('self', 80, 100, False), ('self', 60, 100, True),
                                         ('x', 80, 150, False), ('map', 95, 100, False)]:
        wd = tempfile.mkdtemp(prefix='run_', dir=work)
        outtab = os.path.join(wd, 'out.tab')
        outgff = os.path.join(wd, 'out.gff3')
        if mode == 'self':
            cmds = W.self_LZ_cmds(lzpath=fake, pairs=pairs, splitSelf=strict, outtab=outtab, outgff=outgff,
                                  minIdt=minIdt, minLen=minLen, minCov=3, AchrmLens=lens_path,
                                  label='Self_Repeat', prefix='Self_Repeat')
        elif mode == 'x':
            cmds = W.xspecies_LZ_cmds(lzpath=fake, pairs=pairs, outtab=outtab, outgff=outgff, minIdt=minIdt,
                                      minLen=minLen, minCov=5, AchrmLens=lens_path, label='B_Repeat',
                                      prefix='B_Repeat')
        else:
            cmds = W.map_LZ_cmds(lzpath=fake, pairs=pairs, minIdt=minIdt, minLen=minLen, outfile=outtab)
        cut = next((i for i, c in enumerate(cmds) if 'genomecov' in c), len(cmds))
        tmpd = run_reference_cmds(U, cmds[:cut], wd)
        case = {'mode': mode, 'minIdt': minIdt, 'minLen': minLen, 'strictSelf': strict,
                'pairs': [[os.path.basename(a)[:-3], os.path.basename(b)[:-3]] for a, b in pairs],
                'outtab': open(outtab).read()}
        if strict:
            case['outtab_intra'] = open(outtab + '_intra.tab').read()
        bed = os.path.join(tmpd, 'temp_sorted.bed')
        if os.path.exists(bed):
            case['sorted_bed'] = open(bed).read()
        cases.append(case)
    with open(os.path.join(HERE, 'filter_stage.json'), 'w') as f:
        json.dump({'general': general, 'cases': cases}, f, indent=1)

    # ---- A14: GFF formatter on a prepared merged BED -----------------------------------
    merged = 's1\t20\t60\ns1\t65\t70\ns1\t100\t200\ns10\t5\t1005\ns2\t0\t99\ns2\t300\t400\n'
    gff_cases = []
    for mode in ('self', 'x'):
        wd = tempfile.mkdtemp(prefix='gff_', dir=work)
        outgff = os.path.join(wd, 'out.gff3')
        if mode == 'self':
            cmds = W.self_LZ_cmds(lzpath=fake, pairs=pairs, outtab=os.path.join(wd, 'o.tab'), outgff=outgff,
                                  minIdt=80, minLen=100, minCov=3, AchrmLens=lens_path, label='Self_Repeat',
                                  prefix='SR')
        else:
            cmds = W.xspecies_LZ_cmds(lzpath=fake, pairs=pairs, outtab=os.path.join(wd, 'o.tab'), outgff=outgff,
                                      minIdt=80, minLen=100, minCov=5, AchrmLens=lens_path, label='B_Repeat',
                                      prefix='BR')
        tail = cmds[-2:]
        assert 'gff-version' in tail[0] and 'sprintf' in tail[1]
        cwd = os.getcwd()
        os.chdir(wd)
        with open('temp.bed', 'w') as f:
            f.write(merged)
        with open('tail.sh', 'w') as f:
            f.write('\n'.join(tail) + '\n')
        U.syscall('bash tail.sh')
        os.chdir(cwd)
        gff_cases.append({'mode': mode, 'minLen': 100, 'label': 'Self_Repeat' if mode == 'self' else 'B_Repeat',
                          'prefix': 'SR' if mode == 'self' else 'BR', 'merged_bed': merged,
                          'gff': open(outgff).read()})
    with open(os.path.join(HERE, 'gff_stage.json'), 'w') as f:
        json.dump(gff_cases, f, indent=1)

    # ---- A16: import_Align + writeGFFlines ------------------------------------------------
    tab_rows = ['#name1\tstrand1\tstart1\tend1\tname2\tstrand2\tstart2+\tend2+\tscore\tidentity']
    for k in range(40):
        t = rng.choice(names)
        s = rng.choice([5, 50, 200, 1000, 1000, 20000, rng.randint(1, 100000)])
        ln = rng.choice([99, 100, 101, 500, 5000])
        tab_rows.append('\t'.join(map(str, [t, '+', s, s + ln, rng.choice(names), rng.choice('+-'),
                                            rng.randint(1, 100000), rng.randint(100001, 200000),
                                            rng.randint(3000, 99999), rng.choice(['94.9', '95.0', '98.7', '100.0'])])))
    tab_text = '\n'.join(tab_rows) + '\n'
    wd = tempfile.mkdtemp(prefix='map_', dir=work)
    tab_path = os.path.join(wd, 'in.tab')
    with open(tab_path, 'w') as f:
        f.write(tab_text)
    df = W.import_Align(infile=tab_path, prefix='HGT', minLen=100, minIdt=95)
    gff = ''.join(W.writeGFFlines(alnDF=df, chrlens=[(n, str(l)) for n, l in chromlens], ftype='BHit'))
    with open(os.path.join(HERE, 'map_import.json'), 'w') as f:
        json.dump({'tab': tab_text, 'prefix': 'HGT', 'minLen': 100, 'minIdt': 95, 'label': 'BHit',
                   'chromlens': chromlens, 'uids': list(df['UID']), 'gff': gff}, f, indent=1)

    shutil.rmtree(work)
    print('golden vectors written to', HERE)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Soak (build container only: needs /root/reference, bash, awk, sed, sort): the host text stages A11 / A12 / A16 of
mimeo_amd.formats (A11 / A12 / A14 / A16) against the REFERENCE'S OWN pipeline on random inputs.

    python scripts/soak_text_stages.py [cases] [first seed]

Per case the reference's own command builders (mimeo.wrappers.self_LZ_cmds / xspecies_LZ_cmds / map_LZ_cmds, imported from
/root/reference/src as tests/golden/make_golden.py does: an empty placeholder module named Bio satisfies the import line, nothing
used touches it) make the command list, and the reference's own run_cmd runs it — `lastz` being a stand-in script that copies a
prepared synthetic --format=general file to --output= (LASTZ is absent; the alignment arithmetic is not what is tested here) — up to
the first bedtools line (bedtools is absent).  The TAB (and the _intra TAB of --strictSelf) and the sorted BED must equal what
formats.tab_block / formats.bed_intervals make of the same records; import_Align + writeGFFlines (pandas) must equal
formats.import_align + gff_map_lines; the reference's GFF3 formatter (its last two commands: header echo + the minLen / awk line, A14) run on a
random merged BED must equal formats.gff_repeat_lines.  Random thresholds, names that sort differently in C and in Python order, identities on
and next to the printed-tenth boundaries."""
import os
import random
import shutil
import stat
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_golden as MG                      # noqa: E402
from mimeo_amd import _ffi, formats           # noqa: E402


def alns_from_general(text):
    recs = []
    for line in text.split('\n'):
        if not line or line.startswith('#'):
            continue
        f = line.split('\t')
        n, d = f[11].split('/')
        recs.append((0, 0, int(f[2]) - 1, int(f[3]), int(f[7]) - 1, int(f[8]), int(f[10]), int(n), int(d), 1 if f[6] == '-' else 0, 0))
    return np.array(recs, dtype=_ffi.ALIGNMENT)


def rows(rng, tname, qname, n):
    out = MG.synth_general_rows(rng, tname, qname, n)
    # identities next to the printed-tenth boundaries of random thresholds, and ties of every sort key
    for _ in range(n // 3):
        idd = rng.randint(50, 3000)
        pct = rng.choice([60, 75, 79.95, 80, 80.05, 94.95, 95, 99.95])
        idn = max(1, min(idd, int(round(idd * pct / 100.0)) + rng.choice([-1, 0, 0, 1])))
        length1 = rng.choice([99, 100, 101, 149, 150, 151, 400])
        start1 = rng.choice([1, 1, 5, 5, 1000])
        f = [tname, '+', start1, start1 + length1 - 1, length1, qname, rng.choice('+-'), rng.choice([7, 7, 300]), 0, length1,
             rng.choice([3000, 3000, 5000]), '%d/%d' % (idn, idd), '%.1f%%' % (100.0 * idn / idd)]
        f[8] = f[7] + length1 - 1
        out.append('\t'.join(map(str, f)))
    rng.shuffle(out)
    return out


def one(U, W, seed, work, fake):
    rng = random.Random(seed)
    names = rng.sample(['s1', 's10', 's2', 'Chr_B', 'chr_a', 'Z', 'a.1', 'a-1', 'A1', 'scaffold_100', 'scaffold_99'], rng.randint(2, 4))
    rows_dir = os.path.join(work, 'rows')
    shutil.rmtree(rows_dir, ignore_errors=True)
    os.makedirs(rows_dir)
    os.environ['GOLDEN_ROWS_DIR'] = rows_dir
    gdir = os.path.join(work, 'g')
    os.makedirs(gdir, exist_ok=True)
    pairs = [(os.path.join(gdir, a + '.fa'), os.path.join(gdir, b + '.fa')) for a in names for b in names]
    general = {}
    for a in names:
        for b in names:
            text = '#name1\tstrand1\tstart1\tend1\tlength1\tname2\tstrand2\tstart2+\tend2+\tlength2\tscore\tidentity\tidPct\n'
            r = rows(rng, a, b, rng.randint(0, 20))
            text += ('\n'.join(r) + '\n' if r else '') + '# lastz end-of-file\n'
            general[(b, a)] = text
            with open(os.path.join(rows_dir, '%s_onto_%s.general' % (b, a)), 'w') as f:
                f.write(text)
    lens_path = os.path.join(work, 'A_gen_lens.txt')
    with open(lens_path, 'w') as f:
        for n in names:
            f.write('%s\t%d\n' % (n, 400000))
    mode = rng.choice(['self', 'self', 'x', 'map'])
    minIdt, minLen = rng.choice([60, 75, 80, 95]), rng.choice([100, 100, 150])
    strict = mode == 'self' and rng.random() < 0.5
    wd = tempfile.mkdtemp(prefix='run_', dir=work)
    outtab, outgff = os.path.join(wd, 'out.tab'), os.path.join(wd, 'out.gff3')
    if mode == 'self':
        cmds = W.self_LZ_cmds(lzpath=fake, pairs=pairs, splitSelf=strict, outtab=outtab, outgff=outgff, minIdt=minIdt, minLen=minLen,
                              minCov=3, AchrmLens=lens_path, label='Self_Repeat', prefix='Self_Repeat')
    elif mode == 'x':
        cmds = W.xspecies_LZ_cmds(lzpath=fake, pairs=pairs, outtab=outtab, outgff=outgff, minIdt=minIdt, minLen=minLen, minCov=5,
                                  AchrmLens=lens_path, label='B_Repeat', prefix='B_Repeat')
    else:
        cmds = W.map_LZ_cmds(lzpath=fake, pairs=pairs, minIdt=minIdt, minLen=minLen, outfile=outtab)
    cut = next((i for i, c in enumerate(cmds) if 'genomecov' in c), len(cmds))
    tmpd = MG.run_reference_cmds(U, cmds[:cut], wd)
    bad = []
    stats = {'tab_rows': 0, 'bed_rows': 0, 'map_gff_lines': 0, 'gff_lines': 0}
    main, intra = [formats.TAB_HEADER], [formats.TAB_HEADER]
    for a in names:
        for b in names:
            block = formats.tab_block(alns_from_general(general[(b, a)]), a, b, minLen, minIdt)
            (intra if strict and a == b else main).extend(block)
    stats['tab_rows'] = len(main) + len(intra) - 2
    if '\n'.join(main) + '\n' != open(outtab).read():
        bad.append('tab')
    if strict and '\n'.join(intra) + '\n' != open(outtab + '_intra.tab').read():
        bad.append('intra_tab')
    bed = os.path.join(tmpd, 'temp_sorted.bed')
    if os.path.exists(bed):
        text = open(bed).read().strip()
        exp_lines = [l.split('\t') for l in text.split('\n')] if text else []
        cnames = sorted(names, key=lambda s: s.encode())
        cid = {n: i for i, n in enumerate(cnames)}
        iv = formats.bed_intervals(formats.parse_tab(outtab), cid)
        # the projection is compared as a multiset: the engine sorts the intervals itself (K7, by C-locale rank of the name, start,
        # end — the reference's `sort -k 1,1 -k 2n,2 -k 3n,3` under LC_ALL=C), the file order is not an output
        got = sorted((cnames[c], int(s), int(e)) for c, s, e in iv.tolist())
        stats['bed_rows'] = len(got)
        if got != sorted((c, int(s), int(e)) for c, s, e in exp_lines):
            bad.append('sorted_bed')
        # (the reference's order — `sort -k 1,1 -k 2n,3n`: name, start, then the whole line as a string — is (C-locale rank, start)
        # with ties in text order: all genomecov needs; checked here on the two keys that matter)
        keys = [(cid[c], int(s)) for c, s, e in exp_lines]
        if keys != sorted(keys):
            bad.append('reference_bed_not_sorted_by_rank_and_start')
    if mode == 'map' and os.path.getsize(outtab) > 200:
        try:
            df = W.import_Align(infile=outtab, prefix='HGT', minLen=minLen, minIdt=minIdt)
            chrl = [(n, '400000') for n in names]
            exp = ''.join(W.writeGFFlines(alnDF=df, chrlens=chrl, ftype='BHit'))
            mine = formats.import_align(formats.parse_tab(outtab), 'HGT', minLen, minIdt)
            got = ''.join(formats.gff_map_lines(mine, [(n, 400000) for n in names], 'BHit'))
            stats['map_gff_lines'] = exp.count('\n')
            if got != exp:
                bad.append('map_gff')
        except SystemExit:
            pass   # the reference exits on an empty table; formats.import_align does too (tests/test_host_formats.py)
    if mode in ('self', 'x') and not strict:   # (--strictSelf writes its header earlier and two sections: workflow tests)
        # A14: the reference's last two commands (GFF header + the minLen / awk formatter) on a random merged BED
        tail = cmds[-2:]
        assert 'gff-version' in tail[0] and 'sprintf' in tail[1]
        cnames = sorted(names, key=lambda s: s.encode())
        regs = []
        for c in cnames:
            pos = 0
            for _ in range(rng.randint(0, 12)):
                pos += rng.randint(1, 5000)
                ln = rng.choice([1, 50, minLen - 1, minLen, minLen + 1, 1000, rng.randint(1, 30000)])
                regs.append((c, pos, pos + ln))
                pos += ln
        cwd = os.getcwd()
        os.chdir(wd)
        try:
            with open('temp.bed', 'w') as f:
                f.write(''.join('%s\t%d\t%d\n' % r for r in regs))
            with open('tail.sh', 'w') as f:
                f.write('\n'.join(tail) + '\n')
            U.syscall('bash tail.sh')
        finally:
            os.chdir(cwd)
        exp = open(outgff).read()
        kept = np.array([(cnames.index(c), a, b) for c, a, b in regs if b - a >= minLen], dtype=_ffi.INTERVAL)
        src, label, prefix = ('mimeo-self', 'Self_Repeat', 'Self_Repeat') if mode == 'self' else ('mimeo', 'B_Repeat', 'B_Repeat')
        got = formats.GFF_HEADER + '\n' + ''.join(l + '\n' for l in formats.gff_repeat_lines(kept, cnames, src, label, prefix))
        stats['gff_lines'] = exp.count('\n')
        if got != exp:
            bad.append('gff')
    shutil.rmtree(wd, ignore_errors=True)
    return bad, mode, strict, stats


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    os.environ['LC_ALL'] = 'C'
    U, W = MG.import_reference()
    work = tempfile.mkdtemp(prefix='soak_text_')
    fake = os.path.join(work, 'fake_lastz.sh')
    with open(fake, 'w') as f:
        f.write(MG.FAKE_LASTZ)
    os.chmod(fake, os.stat(fake).st_mode | stat.S_IEXEC)
    fails, modes, tot = 0, {}, {}
    for k in range(n):
        bad, mode, strict, st = one(U, W, seed + k, work, fake)
        for kk, v in st.items():
            tot[kk] = tot.get(kk, 0) + v
        key = mode + ('+strict' if strict else '')
        modes[key] = modes.get(key, 0) + 1
        if bad:
            fails += 1
            print('MISMATCH seed %d (%s): %s' % (seed + k, key, ' '.join(bad)), flush=True)
    shutil.rmtree(work, ignore_errors=True)
    print('soak_text_stages: %d cases (seeds %d..%d; %s), compared %s, %d mismatching cases' % (n, seed, seed + n - 1, modes, tot, fails), flush=True)
    return 1 if fails else 0


if __name__ == '__main__':
    sys.exit(main())

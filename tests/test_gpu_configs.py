"""The BASELINE.json configurations with THEIR OWN flags, at test size, through the CLI entry points and
against the oracle pipeline (C alignment oracle -> Python restatement of the reference's text stages):
  C3  `mimeo x  --minCov 5`                        src/mimeo/run_interspecies.py:173-258, wrappers.py:847-894
  C5  `mimeo map --minIdt 98 --maxtandem 40 --writeTRF`   src/mimeo/run_map.py:190-328
C2 / C4 at full unit size are in test_gpu_fullsize.py (size-independent properties)."""
import numpy as np
import pytest

from mimeo_amd.synth import make_families, synth_genome, write_fasta

pytestmark = pytest.mark.gpu

_ACGT = np.frombuffer(b'ACGT', dtype=np.uint8)


def _oracle_tab(an, aseq, bn, bseq, pairs, min_len, min_idt):
    from oracle import oracle as O, pipeline as P
    lines = ['#name1\tstrand1\tstart1\tend1\tname2\tstrand2\tstart2+\tend2+\tscore\tidentity']
    for t, q in pairs:
        al = O.align_pair(aseq[t].tobytes(), bseq[q].tobytes())
        al['tid'], al['qid'] = t, q
        text = '\n'.join(P.general_rows(an, bn, [len(s) for s in bseq], al)) + '\n'
        lines += P.filter_project_sort(text, min_len, min_idt)
    return lines


def test_c3_mimeo_x_mincov5_cli_matches_oracle(tmp_path):
    """C3's flags (--minCov 5 --minIdt 80 --minLen 100) end to end through run_interspecies.main: a region of
    A is reported when at least five B alignments (different B scaffolds / strands: --chain keeps one chain
    per unit) cover it."""
    from mimeo_amd import run_interspecies
    from oracle import pipeline as P
    fams = make_families(203, 4, (300, 1200))
    an, aseq = synth_genome(201, 240_000, 2, repeat_frac=0.25, shared_families=fams, prefix='a', max_div=0.1)
    bn, bseq = synth_genome(202, 640_000, 8, repeat_frac=0.25, shared_families=fams, prefix='b', max_div=0.1)
    fa, fb = str(tmp_path / 'A.fa'), str(tmp_path / 'B.fa')
    write_fasta(fa, an, aseq)
    write_fasta(fb, bn, bseq)
    out = tmp_path / 'out'
    run_interspecies.main(['--afasta', fa, '--bfasta', fb, '-d', str(out), '--minIdt', '80', '--minLen', '100', '--minCov', '5'])
    pairs = [(a, b) for a in range(2) for b in range(8)]
    exp_tab = _oracle_tab(an, aseq, bn, bseq, pairs, 100, 80)
    assert (out / 'mimeo_alignment.tab').read_text() == '\n'.join(exp_tab) + '\n'
    bed = P.bed_project_sort(exp_tab)
    iv = [(l.split('\t')[0], int(l.split('\t')[1]), int(l.split('\t')[2])) for l in bed]
    lens = {n: len(s) for n, s in zip(an, aseq)}
    regs5 = P.coverage_collapse(iv, lens, 5, 100)
    exp_gff = P.gff_self_lines(regs5, 'B_Repeat', 'B_Repeat', source='mimeo')
    assert len(regs5) >= 3, 'the case must exercise depth >= 5'
    assert P.coverage_collapse(iv, lens, 3, 100) != regs5, '--minCov 5 must change the answer on this case'
    assert (out / 'mimeo_B_in_A.gff3').read_text() == '\n'.join(exp_gff) + '\n'
    assert (out / 'A_gen_lens.txt').read_text() == ''.join('%s\t%d\n' % (n, len(s)) for n, s in zip(an, aseq))


def _plant(seq, pos, unit, n):
    s = np.frombuffer((unit * n), dtype=np.uint8)
    seq[pos:pos + s.size] = s


def test_c5_mimeo_map_minidt98_tandem_filter_cli_matches_oracle(tmp_path):
    """C5's flags (--minIdt 98 --maxtandem 40 --writeTRF) end to end through run_map.main on a pair of genomes
    that share near-identical repeat copies AND microsatellite loci: the SSR-to-SSR alignments pass the identity
    filter and are the rows the tandem filter has to remove.  Expected files: oracle TAB -> import_Align ->
    trfFilter (CPU restatement of the same tandem scorer) -> writetrf / writeGFFlines."""
    from mimeo_amd import formats, run_map
    from oracle import pipeline as P
    fams = make_families(1003, 3, (400, 1500))
    an, aseq = synth_genome(1001, 300_000, 2, repeat_frac=0.1, shared_families=fams, prefix='a', max_div=0.015, indel_rate=0.002)
    bn, bseq = synth_genome(1002, 300_000, 2, repeat_frac=0.1, shared_families=fams, prefix='b', max_div=0.015, indel_rate=0.002)
    aseq = [s.copy() for s in aseq]
    bseq = [s.copy() for s in bseq]
    # shared microsatellites (period 1-6), collinear on scaffold 1 of both genomes, far from each other
    ssr = [(b'A', 260), (b'CA', 180), (b'AAG', 120), (b'GATA', 100), (b'CCTGA', 70), (b'TTAGGG', 60)]
    for k, (unit, n) in enumerate(ssr):
        _plant(aseq[1], 20_000 + 18_000 * k, unit, n)
        _plant(bseq[1], 11_000 + 21_000 * k, unit, n + 7)
    fa, fb = str(tmp_path / 'A.fa'), str(tmp_path / 'B.fa')
    write_fasta(fa, an, aseq)
    write_fasta(fb, bn, bseq)
    out = tmp_path / 'out'
    run_map.main(['--afasta', fa, '--bfasta', fb, '-d', str(out), '--minIdt', '98', '--minLen', '100', '--maxtandem', '40',
                  '--writeTRF', '--gffout', 'map.gff3', '--prefix', 'HGT'])
    pairs = [(a, b) for a in range(2) for b in range(2)]
    exp_tab = _oracle_tab(an, aseq, bn, bseq, pairs, 100, 98)
    assert (out / 'mimeo_alignment.tab').read_text() == '\n'.join(exp_tab) + '\n'
    rows = P.import_align('\n'.join(exp_tab), 'HGT', 100, 98)
    seq_of = {n: s.tobytes() for n, s in zip(an, aseq)}
    kept = P.trf_filter(rows, seq_of, prefix='HGT', maxtandem=40)
    assert 0 < len(kept) < len(rows), 'the case must have rows on both sides of the tandem filter (%d of %d kept)' % (len(kept), len(rows))
    dropped = [r for r in rows if r[:10] not in [k[:10] for k in kept]]
    assert any(r[0] == an[1] for r in dropped)  # the SSR loci of scaffold 1
    trf_lines = ['\t'.join(['#name1', 'strand1', 'start1', 'end1', 'name2', 'strand2', 'start2+', 'end2+', 'score', 'identity'])]
    trf_lines += ['\t'.join(r[:10]) for r in kept]
    assert (out / 'mimeo_alignment.tab.trf').read_text() == '\n'.join(trf_lines) + '\n'
    exp_gff = P.gff_map_lines(kept, formats.chromlens(an, aseq), 'BHit')
    assert (out / 'map.gff3').read_text() == '\n'.join(exp_gff) + '\n'

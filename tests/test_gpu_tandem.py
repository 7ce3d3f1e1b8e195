"""K8 tandem scorer (stands in for TRF in `mimeo map --maxtandem`, wrappers.py:120-262): exact
agreement with the CPU restatement of the same specification, and the behaviour the filter
needs — microsatellites are masked, random sequence is not.  Agreement with TRF itself cannot be
measured here (TRF is absent): PARITY UNPINNED."""
import numpy as np
import pytest

from mimeo_amd.synth import synth_genome

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from mimeo_amd import engine
    engine.init(0)
    return engine


def _with_ssrs(seed):
    rng = np.random.default_rng(seed)
    names, seqs = synth_genome(seed, 60_000, 2, repeat_frac=0.0)
    out, spans = [], []
    for s in seqs:
        s = s.copy()
        sp = []
        for unit in (b'A', b'AC', b'AAG', b'ACGT', b'AAAAG', b'ACACGT', b'ACGTTGCAAC', b'ACGGTCATTGACCGTAAGCTTAGCAT'):
            p = int(rng.integers(100, s.size - 3000))
            ln = int(rng.integers(60, 900))
            rep = np.frombuffer((unit * (ln // len(unit) + 1))[:ln], dtype=np.uint8).copy()
            mut = rng.random(ln) < 0.03
            rep[mut] = np.frombuffer(b'ACGT', np.uint8)[rng.integers(0, 4, int(mut.sum()))]
            s[p:p + ln] = rep
            sp.append((p, p + ln))
        s[500:520] = ord('N')
        s[3000:3300] |= 0x20
        out.append(s)
        spans.append(sp)
    return names, out, spans


def test_tandem_masked_equals_cpu_restatement(eng):
    from oracle import pipeline as P
    names, seqs, spans = _with_ssrs(3)
    g = eng.Genome(names, seqs)
    rng = np.random.default_rng(4)
    iv = []
    for c in range(2):
        for a, b in spans[c]:
            iv.append((c, max(0, a - int(rng.integers(0, 80))), min(len(seqs[c]), b + int(rng.integers(0, 80)))))
        for _ in range(12):
            a = int(rng.integers(0, len(seqs[c]) - 2500))
            iv.append((c, a, a + int(rng.integers(1, 2400))))
    iv += [(0, 10, 10), (0, 5, 6), (1, 29990, 40000), (0, 495, 530)]
    # (match, mismatch, minscore, maxperiod, delta): TRF's defaults, variations, and delta 0 = the gap-free scorer
    for params in ((2, 7, 50, 50, 7), (2, 5, 30, 12, 3), (3, 7, 80, 64, 9), (2, 7, 50, 50, 0)):
        got = eng.tandem_masked(g, np.array(iv, dtype=np.uint32), *params)
        exp = [P.tandem_masked(seqs[c].tobytes(), s, min(e, len(seqs[c])), *params) for c, s, e in iv]
        assert got.tolist() == exp, params
    g.close()


def test_ssrs_with_indels_are_masked(eng):
    """What v2 is for: microsatellites carry indels (TRF's default model expects 10 %).  Arrays of period 2-6 and
    10 with 2 % single-base insertions / deletions plus 2 % substitutions must still be masked almost entirely with the
    indel moves on (tdelta 7), never less than by the gap-free comparison (tdelta 0: an indel costs it a period's worth
    of mismatches, which splits long periods into segments); random sequence stays unmasked either way."""
    rng = np.random.default_rng(12)
    acgt = np.frombuffer(b'ACGT', np.uint8)
    seq = acgt[rng.integers(0, 4, 40_000)].copy()
    spans = []
    pos = 1000
    for unit in (b'AC', b'AAG', b'ACGT', b'AAAAG', b'ACACGT', b'ACGTTGCAAC'):
        ln = 600
        rep = bytearray((unit * (ln // len(unit) + 2))[:ln])
        out = bytearray()
        for c in rep:
            r = rng.random()
            if r < 0.01:
                continue                                  # deletion
            if r < 0.02:
                out.append(int(acgt[rng.integers(0, 4)]))  # insertion in front
            if rng.random() < 0.02:
                c = int(acgt[rng.integers(0, 4)])
            out.append(c)
        seq[pos:pos + len(out)] = np.frombuffer(bytes(out), np.uint8)
        spans.append((pos, pos + len(out)))
        pos += 3000
    g = eng.Genome(['s'], [seq])
    iv = np.array([(0, a, b) for a, b in spans], dtype=np.uint32)
    with_indels = eng.tandem_masked(g, iv, 2, 7, 50, 50, 7) / np.array([b - a for a, b in spans])
    gap_free = eng.tandem_masked(g, iv, 2, 7, 50, 50, 0) / np.array([b - a for a, b in spans])
    assert (with_indels > 0.85).all(), with_indels
    assert (with_indels >= gap_free - 1e-9).all(), (with_indels, gap_free)
    rnd = np.array([(0, a, a + 1000) for a in range(20_000, 38_000, 1000)], dtype=np.uint32)
    assert (eng.tandem_masked(g, rnd, 2, 7, 50, 50, 7) / 1000.0 < 0.08).all()
    g.close()


def test_ssr_masked_random_not(eng):
    names, seqs, spans = _with_ssrs(5)
    g = eng.Genome(names, seqs)
    ssr = [(0, a, b) for a, b in spans[0]]
    m = eng.tandem_masked(g, np.array(ssr, dtype=np.uint32))
    frac = m / np.array([b - a for _, a, b in ssr])
    assert (frac[:6] > 0.85).all(), frac  # periods 1..6 are masked almost entirely
    rnd = [(1, a, a + 1000) for a in range(5000, 25000, 1000) if not any(x < a + 1000 and a < y for x, y in spans[1])]
    m2 = eng.tandem_masked(g, np.array(rnd, dtype=np.uint32))
    assert (m2 / 1000.0 < 0.05).all()
    g.close()


def test_trf_filter_and_writetrf(eng, tmp_path):
    from mimeo_amd import formats, workflow
    names, seqs, spans = _with_ssrs(6)
    g = eng.Genome(names, seqs)
    a, b = spans[0][2]
    rows = [[names[0], '+', str(a), str(b), names[1], '+', '1', '500', '9999', '99.0'],
            [names[0], '+', '20000', '21000', names[1], '-', '7', '1007', '8888', '98.5'],
            [names[1], '+', '20000', '21500', names[0], '+', '9', '1509', '7777', '98.1']]
    kept = workflow.trf_filter(rows, g, prefix='HGT', maxtandem=40)
    assert [r[2] for r in kept] == ['20000', '20000'] and [r[10] for r in kept] == ['HGT_1', 'HGT_2']
    assert len(workflow.trf_filter(rows, g, prefix='HGT', maxtandem=101)) == 3
    out = formats.write_trf_tab(kept, str(tmp_path / 'o.tab'))
    lines = open(out).read().split('\n')
    assert lines[0].startswith('#name1\tstrand1') and lines[1].split('\t')[2] == '20000' and out.endswith('.tab.trf')
    g.close()


def test_filter_command_drops_ssr_rich_records(eng, tmp_path):
    """`mimeo filter` (run_filter.py:127-210 / wrappers.py:265-377): a library record stays when less than
    maxtandem percent of it is tandem (or N); survivors are written in input order."""
    from mimeo_amd import formats, run_filter
    rng = np.random.default_rng(5)
    rnd = lambda n: bytes(rng.choice(np.frombuffer(b'ACGT', dtype=np.uint8), n))
    recs = [('te1', rnd(900)),
            ('ssr_ca', b'CA' * 300),
            ('half', rnd(600) + b'AAG' * 190),           # ~49 % tandem
            ('mostlyN', rnd(300) + b'N' * 400),
            ('te2', rnd(2000) + b'AT' * 30)]            # 3 % tandem
    lib = tmp_path / 'lib.fa'
    lib.write_bytes(b''.join(b'>' + n.encode() + b' desc\n' + s + b'\n' for n, s in recs))
    run_filter.main(['--infile', str(lib), '-d', str(tmp_path / 'out')])
    names, seqs = formats.read_fasta(str(tmp_path / 'out' / 'lib_filtered.fa'))
    assert names == ['te1', 'te2']
    assert seqs[0].tobytes() == recs[0][1] and seqs[1].tobytes() == recs[4][1]
    keep = run_filter.filter_fasta(str(lib), str(tmp_path / 'o2.fa'), maxtandem=60)
    assert keep == ['te1', 'half', 'mostlyN', 'te2']

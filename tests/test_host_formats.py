"""Host-side text stages (mimeo_amd.formats) against the vectors produced by the reference's
own pipeline (tests/golden/make_golden.py).  No GPU, no oracle needed."""
import json
import os

import numpy as np
import pytest

from mimeo_amd import _ffi, formats

G = os.path.join(os.path.dirname(__file__), 'golden')


def _load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def _alns_from_general(text):
    """Turn synthetic lastz general rows back into engine records (what K6 would deliver)."""
    recs = []
    for line in text.split('\n'):
        if not line or line.startswith('#'):
            continue
        f = line.split('\t')
        n, d = f[11].split('/')
        recs.append((0, 0, int(f[2]) - 1, int(f[3]), int(f[7]) - 1, int(f[8]), int(f[10]), int(n), int(d),
                     1 if f[6] == '-' else 0, 0))
    return np.array(recs, dtype=_ffi.ALIGNMENT)


@pytest.mark.parametrize('k', range(4))
def test_tab_block_matches_reference_pipeline(k):
    d = _load('filter_stage.json')
    case = d['cases'][k]
    main, intra = [formats.TAB_HEADER], [formats.TAB_HEADER]
    for t, q in case['pairs']:
        alns = _alns_from_general(d['general']['%s_onto_%s' % (q, t)])
        block = formats.tab_block(alns, t, q, case['minLen'], case['minIdt'])
        (intra if case['strictSelf'] and t == q else main).extend(block)
    assert '\n'.join(main) + '\n' == case['outtab']
    if case['strictSelf']:
        assert '\n'.join(intra) + '\n' == case['outtab_intra']


def test_bed_projection_matches_reference(tmp_path):
    d = _load('filter_stage.json')
    case = d['cases'][0]
    p = tmp_path / 'o.tab'
    p.write_text(case['outtab'])
    names = sorted({l.split('\t')[0] for l in case['sorted_bed'].strip().split('\n')}, key=lambda s: s.encode())
    cid = {n: i for i, n in enumerate(names)}
    iv = formats.bed_intervals(formats.parse_tab(str(p)), cid)
    exp = sorted((cid[c], int(s), int(e)) for c, s, e in (l.split('\t') for l in case['sorted_bed'].strip().split('\n')))
    assert sorted(map(tuple, iv.tolist())) == exp


def test_gff_repeat_lines_match_reference_awk():
    for case in _load('gff_stage.json'):
        names = sorted({l.split('\t')[0] for l in case['merged_bed'].strip().split('\n')}, key=lambda s: s.encode())
        regs = [(names.index(c), int(s), int(e)) for c, s, e in (l.split('\t') for l in case['merged_bed'].strip().split('\n'))
                if int(e) - int(s) >= case['minLen']]
        regions = np.array(regs, dtype=_ffi.INTERVAL)
        src = 'mimeo-self' if case['mode'] == 'self' else 'mimeo'
        got = formats.GFF_HEADER + '\n' + ''.join(l + '\n' for l in formats.gff_repeat_lines(regions, names, src, case['label'], case['prefix']))
        assert got == case['gff']


def test_import_align_and_map_gff_match_reference_pandas(tmp_path):
    d = _load('map_import.json')
    p = tmp_path / 'in.tab'
    p.write_text(d['tab'])
    rows = formats.import_align(formats.parse_tab(str(p)), d['prefix'], d['minLen'], d['minIdt'])
    assert [r[10] for r in rows] == d['uids']
    assert ''.join(formats.gff_map_lines(rows, [(n, str(l)) for n, l in d['chromlens']], d['label'])) == d['gff']


def test_import_align_exits_when_empty():
    with pytest.raises(SystemExit):
        formats.import_align([], 'x', 100, 95)


def test_read_fasta_and_chromlens(tmp_path):
    p = tmp_path / 'g.fa'
    p.write_text('>s2 desc here\nACGT\nacgtn\n>s10\nAAAA\n>s1\n\n')
    names, seqs = formats.read_fasta(str(p))
    assert names == ['s2', 's10', 's1'] and [len(s) for s in seqs] == [9, 4, 0]
    lens = formats.chromlens(names, seqs, str(tmp_path / 'lens.txt'))
    assert lens == [('s1', '0'), ('s10', '4'), ('s2', '9')]
    assert (tmp_path / 'lens.txt').read_text() == 's1\t0\ns10\t4\ns2\t9\n'
    with pytest.raises(SystemExit):
        formats.check_unique(['a', 'b', 'a'])


def test_identity_rounding_is_printf():
    assert formats.identity_pct(1599, 2000) == '%.1f' % (100.0 * 1599 / 2000)
    assert formats.identity_pct(1, 1) == '100.0'


def test_printed_tenths_is_what_printf_prints():
    """The vectorised digits of the printed identity (wrappers.py:1040, :1052: awk compares the PRINTED one-decimal value with
    minIdt) against Python's own '%.1f' on 3 * 10^5 (n, d) pairs: random ones, every n for a few denominators that make exact
    x.x5 ties (2000, 400, 8000), d = 0, and the A11 filter of bench.py built on it against the formatted strings."""
    rng = np.random.default_rng(5)
    d = rng.integers(1, 200_000, 250_000)
    n = (d * rng.uniform(0.5, 1.0, d.size)).astype(np.int64)
    for den in (2000, 400, 8000, 3, 7):
        k = np.arange(0, den + 1)
        d, n = np.concatenate([d, np.full(k.size, den)]), np.concatenate([n, k])
    d, n = np.concatenate([d, np.zeros(5, np.int64)]), np.concatenate([n, np.arange(5)])
    got = formats.printed_tenths(n, d)
    exp = np.array([int(('%.1f' % (100.0 * a / b)).replace('.', '')) if b else 0 for a, b in zip(n.tolist(), d.tolist())])
    assert np.array_equal(got, exp)
    import bench
    a = np.zeros(n.size, dtype=_ffi.ALIGNMENT)
    a['tend'] = 500
    a['id_n'], a['id_d'] = n, d
    for min_idt in (80, 95, 98, 87.5):
        keep = np.array([float('%.1f' % (100.0 * x / y)) if y else 0.0 for x, y in zip(n.tolist(), d.tolist())]) >= min_idt
        assert np.array_equal(bench.a11_filter(a, 100, min_idt)['id_n'], a['id_n'][keep])

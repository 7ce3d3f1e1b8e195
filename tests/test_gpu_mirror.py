"""Shared plus strand, per-pair strands and per-pair failure (mimeo_align_units, mimeo_get_failed_pairs).

The reference enumerates BOTH orders of every scaffold pair of a self job (src/mimeo/utils.py:97-102) and runs lastz on
each.  The gap-free stage is symmetric in target and query, so the engine computes the plus-strand unit of an unordered
pair once and emits its HSPs for both orders (k4_mirror_hsps) — unless a scaffold has soft-masked bases (lastz keeps them
out of TARGET seeding only).  Chain and gapped extension still run per ordered pair."""
import hashlib

import numpy as np
import pytest

from mimeo_amd.synth import synth_genome

pytestmark = pytest.mark.gpu

ACOLS = ['tstart', 'tend', 'qstart', 'qend', 'score', 'id_n', 'id_d', 'qstrand']


@pytest.fixture(scope='module')
def eng():
    from mimeo_amd import engine
    engine.init(0)
    return engine


def _oracle_pair(O, T, Q, **kw):
    return O.align_pair(T.tobytes(), Q.tobytes(), O.default_params(**kw))


def test_transposed_hsps_equal_the_oracles_hsps_of_the_swapped_pair(eng, monkeypatch):
    """--chain and --gapped off: the rows are the HSPs.  (1, 0) receives the HSPs of (0, 1) transposed; they must be the
    oracle's HSPs of lastz(target 1, query 0) — both strands (the minus strand is never shared)."""
    from oracle import oracle as O
    names, seqs = synth_genome(91, 1_200_000, 2, repeat_frac=0.1, families=5, cons_len=(200, 3000))
    g = eng.Genome(names, seqs)
    for k in ('MIMEO_MIRROR', 'MIMEO_PACK'):
        monkeypatch.delenv(k, raising=False)
    got = eng.align_pairs(g, None, [(0, 1), (1, 0)], eng.default_params(chain=0, gapped=0))
    st = eng.stats()
    assert st['scan_launches'] == 3 and st['pair_strands'] == 4
    for t, q in ((0, 1), (1, 0)):
        exp = _oracle_pair(O, seqs[t], seqs[q], chain=0, gapped=0)
        a = np.sort(got[(got['tid'] == t) & (got['qid'] == q)][ACOLS], order=ACOLS)
        b = np.sort(exp[ACOLS], order=ACOLS)
        assert a.size == b.size > 20 and (a == b).all(), (t, q, a.size, b.size)
    g.close()


def test_soft_masked_scaffold_falls_back_to_two_scans(eng):
    """lower-case target bases take no part in seeding (target role only): (t, q, +) and (q, t, +) are no transposes any more
    and the engine must run both — checked against the oracle, which reads the case."""
    from oracle import oracle as O
    names, seqs = synth_genome(92, 900_000, 3, repeat_frac=0.12, families=4, cons_len=(300, 2500))
    seqs[1] = seqs[1].copy()
    seqs[1][100_000:180_000] |= 0x20
    g = eng.Genome(names, seqs)
    pairs = [(t, q) for t in range(3) for q in range(3)]
    got = eng.align_pairs(g, None, pairs)
    st = eng.stats()
    assert st['scan_launches'] == 17   # only {0, 2} is shared; the pairs with scaffold 1 are scanned in both orders
    for t, q in pairs:
        exp = _oracle_pair(O, seqs[t], seqs[q])
        a = np.sort(got[(got['tid'] == t) & (got['qid'] == q)][ACOLS], order=ACOLS)
        b = np.sort(exp[ACOLS], order=ACOLS)
        assert a.size == b.size and (a == b).all(), (t, q)
    g.close()


def test_units_api_rows_cover_the_job_and_give_the_same_records(eng, monkeypatch):
    """dist.units_of_row: the rows of a self job name every unit exactly once; aligned row by row through
    mimeo_align_units (plus-strand pairs in both orders in the row that owns them) the records are those of one
    mimeo_align_pairs call over the whole matrix, and of the same call without sharing."""
    from mimeo_amd import dist
    S = 5
    names, seqs = synth_genome(93, 2_000_000, S, repeat_frac=0.1, families=6, cons_len=(300, 2500))
    g = eng.Genome(names, seqs)
    seen = {}
    for t in range(S):
        for a, b, m in dist.units_of_row(t, S):
            for bit in (1, 2):
                if m & bit:
                    assert (a, b, bit) not in seen
                    seen[(a, b, bit)] = t
    assert len(seen) == 2 * S * S
    rows = [len([1 for (a, b, bit), r in seen.items() if r == t and bit == 1 and a != b]) for t in range(S)]
    assert max(rows) - min(rows) <= 2 and sum(rows) == S * S - S
    monkeypatch.delenv('MIMEO_MIRROR', raising=False)
    whole = eng.align_pairs(g, None, [(t, q) for t in range(S) for q in range(S)])
    launches_whole = eng.stats()['scan_launches']
    assert launches_whole == 2 * S * S - (S * S - S) // 2
    parts, launches = [], 0
    for t in range(S):
        parts.append(eng.align_units(g, None, dist.units_of_row(t, S)))
        launches += eng.stats()['scan_launches']
    assert launches == launches_whole
    rows_all = np.concatenate(parts)
    cols = ['tid', 'qid'] + ACOLS
    assert np.array_equal(np.sort(rows_all[cols], order=cols), np.sort(whole[cols], order=cols))
    monkeypatch.setenv('MIMEO_MIRROR', '0')
    plain = eng.align_pairs(g, None, [(t, q) for t in range(S) for q in range(S)])
    assert eng.stats()['scan_launches'] == 2 * S * S
    monkeypatch.delenv('MIMEO_MIRROR')
    assert plain.tobytes() == whole.tobytes()
    # a strand mask is honoured: plus only, minus only
    only_plus = eng.align_units(g, None, [(0, 1, 1)])
    only_minus = eng.align_units(g, None, [(0, 1, 2)])
    both = eng.align_pairs(g, None, [(0, 1)])
    assert (only_plus['qstrand'] == 0).all() and (only_minus['qstrand'] == 1).all()
    assert np.concatenate([only_plus, only_minus]).tobytes() == both.tobytes()
    g.close()


def test_c2_job_with_and_without_the_shared_plus_strand(eng, monkeypatch):
    """The whole C2 job on both paths (super-scaffolds; one unit per pair): same md5 with MIMEO_MIRROR=0."""
    names, seqs = synth_genome(50, 50_000_000, 10)
    A = eng.Genome(names, seqs)
    pairs = [(t, q) for t in range(10) for q in range(10)]
    md5 = {}
    for tag, env in (('packed', {}), ('packed_no_mirror', {'MIMEO_MIRROR': '0'}), ('unit_per_pair', {'MIMEO_PACK': '0'}),
                     ('unit_per_pair_no_mirror', {'MIMEO_PACK': '0', 'MIMEO_MIRROR': '0'})):
        for k in ('MIMEO_MIRROR', 'MIMEO_PACK'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        al = eng.align_pairs(A, None, pairs)
        st = eng.stats()
        md5[tag] = (hashlib.md5(al.tobytes()).hexdigest(), int(al.size), st['scan_launches'], round(st['ms_scan']))
    for k in ('MIMEO_MIRROR', 'MIMEO_PACK'):
        monkeypatch.delenv(k, raising=False)
    assert len({v[:2] for v in md5.values()}) == 1 and md5['packed'][1] > 1000, md5
    assert md5['unit_per_pair'][2] == 200 - 45 and md5['unit_per_pair_no_mirror'][2] == 200, md5
    assert md5['packed'][2] < md5['packed_no_mirror'][2], md5
    A.close()


def test_one_pair_beyond_a_limit_is_left_out_and_the_others_are_returned(eng):
    """A gapped extension whose DP band outgrows 65 536 columns is a documented limit (MIMEO_ERR_LIMIT).  With a tiny gap
    extension penalty and a huge y-drop the band is as wide as the query is long: pairs of the two 100 kbp scaffolds overflow,
    pairs with a 20 kbp query cannot.  The call returns 0, names the failed pairs, and every other pair equals the oracle."""
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b'ACGT', np.uint8)
    rep = rng.integers(0, 4, size=1500, dtype=np.uint8)
    lens = [20_000] * 6 + [100_000] * 2
    seqs = []
    for n in lens:
        s = rng.integers(0, 4, size=n, dtype=np.uint8)
        for _ in range(2):
            c = rep.copy()
            m = rng.random(c.size) < 0.04
            c[m] = (c[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
            p = int(rng.integers(0, n - c.size))
            s[p:p + c.size] = c
        seqs.append(acgt[s])
    names = ['s%d' % i for i in range(len(lens))]
    g = eng.Genome(names, seqs)
    kw = dict(gap_extend=1, ydrop=100_000)
    pairs = [(t, q) for t in range(8) for q in range(8)]
    got = eng.align_pairs(g, None, pairs, eng.default_params(**kw))
    failed = {pairs[i] for i, code in eng.failed_pairs()}
    assert all(code == -5 for _, code in eng.failed_pairs())
    assert failed and all(q >= 6 for _, q in failed), failed      # only a 100 kbp query can carry a band beyond 65 536 columns
    assert 'left out' in eng.last_error()
    for t, q in pairs:
        rows = got[(got['tid'] == t) & (got['qid'] == q)]
        if (t, q) in failed:
            assert rows.size == 0
            continue
        if q >= 6 or (t + q) % 3:   # the oracle pays for every cell of these wide bands: a third of the small pairs will do
            continue
        exp = _oracle_pair(O, seqs[t], seqs[q], **kw)
        a, b = np.sort(rows[ACOLS], order=ACOLS), np.sort(exp[ACOLS], order=ACOLS)
        assert a.size == b.size and (a == b).all(), (t, q)
    # the same call with default parameters fails nowhere
    eng.align_pairs(g, None, pairs)
    assert eng.failed_pairs() == []
    g.close()


def test_cli_self_two_ranks_deal_units_and_write_the_files_of_one_process(tmp_path):
    """`mimeo self` under torchrun (two gloo ranks sharing GPU 0) on five large-ish scaffolds: the ranks are dealt target rows
    with the plus-strand pairs in both orders (dist.deal_units -> mimeo_align_units, shared plus strand inside each rank), and
    rank 0 writes the TAB / GFF3 of a single process — also with --strictSelf (the intra-scaffold rows in their own file)."""
    import os
    import socket
    import subprocess
    import sys
    from mimeo_amd.synth import write_fasta
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names, seqs = synth_genome(47, 5 * 300_000, 5, repeat_frac=0.15, families=5, cons_len=(200, 2000), max_div=0.12)
    fa = str(tmp_path / 'five.fa')
    write_fasta(fa, names, seqs)
    for extra, files in (([], ('mimeo_alignment.tab', 'mimeo-self_repeats.gff3')),
                         (['--strictSelf'], ('mimeo_alignment.tab', 'mimeo_alignment.tab_intra.tab', 'mimeo-self_repeats.gff3'))):
        args = ['self', '--afasta', fa, '--minIdt', '80', '--minLen', '100', '--minCov', '3'] + extra
        tag = 's' if extra else 'p'
        one = subprocess.run([sys.executable, '-m', 'mimeo_amd'] + args + ['-d', str(tmp_path / ('one' + tag))], cwd=root, capture_output=True, text=True,
                             timeout=600)
        assert one.returncode == 0, one.stdout + one.stderr
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        env2 = dict(os.environ, MIMEO_DIST_BACKEND='gloo', MIMEO_FORCE_DEVICE='0')
        two = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                              '--master-port', str(port), '-m', 'mimeo_amd'] + args + ['-d', str(tmp_path / ('two' + tag))], cwd=root,
                             capture_output=True, text=True, timeout=900, env=env2)
        assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-3000:]
        for f in files:
            a, b = (tmp_path / ('one' + tag) / f).read_text(), (tmp_path / ('two' + tag) / f).read_text()
            assert a == b, (extra, f)
        assert (tmp_path / ('one' + tag) / 'mimeo_alignment.tab').read_text().count('\n') > 30

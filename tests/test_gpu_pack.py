"""Super-scaffolds for fragmented assemblies (pack.hip): the per-pair loop of the reference (src/mimeo/wrappers.py:1015-1059)
runs lastz once per ordered scaffold pair; this library runs the seed index and the gap-free stage on small scaffolds
concatenated behind spacers of N and hands every HSP back to its scaffold pair.  The alignments must be byte for byte
those of the unit-per-pair path, and those of the oracle."""
import hashlib

import numpy as np
import pytest

from mimeo_amd.synth import synth_genome, make_families

pytestmark = pytest.mark.gpu

PACK_KNOBS = ('MIMEO_PACK', 'MIMEO_PACK_MIN', 'MIMEO_PACK_SUPER', 'MIMEO_PACK_MEMBER', 'MIMEO_BATCH_UNITS', 'MIMEO_INDEX_BUDGET_MB')


@pytest.fixture(scope='module')
def eng():
    from mimeo_amd import engine
    engine.init(0)
    return engine


def _clear(monkeypatch):
    for k in PACK_KNOBS:
        monkeypatch.delenv(k, raising=False)


def _digest(a):
    return a.size, hashlib.md5(a.tobytes()).hexdigest()


def _fragmented(seed, nscaf, total, **kw):
    names, seqs = synth_genome(seed, total, nscaf, repeat_frac=0.15, families=8, cons_len=(200, 1500), max_div=0.12, **kw)
    rng = np.random.default_rng(seed)
    # unequal lengths (members must not sit on a regular grid), N runs, a soft-masked stretch, one scaffold too short to seed
    for i in range(len(seqs)):
        seqs[i] = seqs[i][:len(seqs[i]) - int(rng.integers(0, len(seqs[i]) // 3))].copy()
    seqs[1][500:530] = ord('N')
    seqs[3][-40:] = ord('N')
    seqs[4][0:25] = ord('N')
    seqs[2][2000:2600] = np.frombuffer(bytes(seqs[2][2000:2600]).lower(), dtype=np.uint8)
    seqs[5] = seqs[5][:12].copy()
    return names, seqs


def test_packed_self_alignments_equal_the_unit_per_pair_path(eng, monkeypatch):
    names, seqs = _fragmented(5, 24, 24 * 30_000)
    g = eng.Genome(names, seqs)
    pairs = [(t, q) for t in range(24) for q in range(24)]
    _clear(monkeypatch)
    monkeypatch.setenv('MIMEO_PACK', '0')
    ref = eng.align_pairs(g, None, pairs)
    st0 = eng.stats()
    assert st0['super_units'] == 0 and st0['pair_strands'] == 2 * 24 * 24
    outs = {}
    for tag, env in (('one_super', {}), ('three_supers', {'MIMEO_PACK_SUPER': '250000'}), ('supers_of_two', {'MIMEO_PACK_SUPER': '50000'}),
                     ('big_ones_alone', {'MIMEO_PACK_MEMBER': '24000', 'MIMEO_PACK_MIN': '2'}), ('one_unit_batches', {'MIMEO_BATCH_UNITS': '1', 'MIMEO_PACK_SUPER': '250000'}),
                     # the supers' seed indexes (77 MB each, three roles) do not fit 400 MB: index blocks over the super x super matrix
                     ('index_blocks', {'MIMEO_PACK_SUPER': '250000', 'MIMEO_INDEX_BUDGET_MB': '400'})):
        _clear(monkeypatch)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        a = eng.align_pairs(g, None, pairs)
        st = eng.stats()
        assert st['super_units'] > 0, tag
        assert st['pair_strands'] == st0['pair_strands'] and st['hsps'] == st0['hsps'] and st['chained_hsps'] == st0['chained_hsps'], (tag, st, st0)
        outs[tag] = _digest(a)
        if tag == 'one_super':
            assert st['super_units'] == 2
        assert (st['index_blocks'] > 1) == (tag == 'index_blocks'), (tag, st['index_blocks'])
    _clear(monkeypatch)
    assert ref.size > 50
    assert set(outs.values()) == {_digest(ref)}, (outs, _digest(ref))
    g.close()


def test_packed_share_of_a_self_job_equals_the_unit_per_pair_path(eng, monkeypatch):
    """A rank's share of a self job (mimeo_amd/dist.py: some targets against every scaffold): target and query roles are
    packed separately, so a scaffold's main diagonal goes through the general follower rule instead of k4_diag0."""
    names, seqs = _fragmented(13, 20, 20 * 20_000)
    g = eng.Genome(names, seqs)
    pairs = [(t, q) for t in (1, 2, 3, 7, 8, 12, 19) for q in range(20)]
    _clear(monkeypatch)
    monkeypatch.setenv('MIMEO_PACK', '0')
    ref = eng.align_pairs(g, None, pairs)
    _clear(monkeypatch)
    a = eng.align_pairs(g, None, pairs)
    st = eng.stats()
    assert st['super_units'] == 2 and ref.size > 20
    assert _digest(a) == _digest(ref)
    g.close()


def test_packed_interspecies_alignments_equal_the_unit_per_pair_path(eng, monkeypatch):
    rng = np.random.Generator(np.random.PCG64(11))
    fams = make_families(rng, 6, (200, 1200))
    na, sa = synth_genome(21, 18 * 25_000, 18, repeat_frac=0.15, shared_families=fams, max_div=0.1, prefix='A')
    nb, sb = synth_genome(22, 5 * 40_000, 5, repeat_frac=0.15, shared_families=fams, max_div=0.1, prefix='B')
    A, B = eng.Genome(na, sa), eng.Genome(nb, sb)
    pairs = [(t, q) for t in range(18) for q in range(5)]
    _clear(monkeypatch)
    monkeypatch.setenv('MIMEO_PACK', '0')
    ref = eng.align_pairs(A, B, pairs)
    _clear(monkeypatch)
    a = eng.align_pairs(A, B, pairs)     # 18 small targets: packed by default; the five queries ride along in one super
    st = eng.stats()
    assert st['super_units'] == 2
    assert ref.size > 20 and _digest(a) == _digest(ref)
    # a pair list that is not a full cross product, and one with a duplicate: the first is left to the other path, the second answered twice
    a2 = eng.align_pairs(A, B, pairs[:-1])
    assert eng.stats()['super_units'] == 0
    r2 = ref[~((ref['tid'] == 17) & (ref['qid'] == 4))]
    assert _digest(a2) == _digest(r2)
    a3 = eng.align_pairs(A, B, pairs + [pairs[3]])
    assert eng.stats()['super_units'] == 2
    dup = ref[(ref['tid'] == pairs[3][0]) & (ref['qid'] == pairs[3][1])]
    assert a3.size == ref.size + dup.size and _digest(a3[:ref.size]) == _digest(ref) and _digest(a3[ref.size:]) == _digest(dup)
    A.close(); B.close()
    _clear(monkeypatch)


def test_packed_alignments_match_the_oracle(eng, monkeypatch):
    from oracle import oracle as O
    names, seqs = _fragmented(9, 16, 16 * 12_000)
    g = eng.Genome(names, seqs)
    pairs = [(t, q) for t in range(16) for q in range(16)]
    _clear(monkeypatch)
    got = eng.align_pairs(g, None, pairs)
    assert eng.stats()['super_units'] == 2
    exp_all = []
    for t, q in pairs:
        e = O.align_pair(seqs[t].tobytes(), seqs[q].tobytes())
        e['tid'], e['qid'] = t, q
        exp_all.append(e)
    exp = np.concatenate(exp_all)
    cols = ['tid', 'qid', 'tstart', 'tend', 'qstart', 'qend', 'score', 'id_n', 'id_d', 'qstrand']
    assert exp.size > 30 and got.size == exp.size
    bad = np.flatnonzero(got[cols] != exp[cols])
    assert bad.size == 0, (got[cols][bad[:5]], exp[cols][bad[:5]])
    g.close()


def test_fragmented_self_workflow_files_do_not_depend_on_packing(eng, tmp_path, monkeypatch):
    """`mimeo self` end to end on a fragmented genome (40 scaffolds of uneven size, one above the member limit): TAB and
    GFF3 written with super-scaffolds and one unit per pair are the same bytes."""
    from mimeo_amd import workflow
    names, seqs = synth_genome(31, 40 * 15_000, 40, repeat_frac=0.2, families=6, cons_len=(200, 1200), max_div=0.1)
    rng = np.random.default_rng(31)
    for i in range(40):
        seqs[i] = seqs[i][:len(seqs[i]) - int(rng.integers(0, 6000))].copy()
    names = ['ctg%d' % (97 * i % 40) for i in range(40)]   # C-locale order differs from FASTA order
    A = eng.Genome(names, seqs)
    pairs = workflow.all_pairs(40)
    outs = {}
    for tag, env in (('packed', {'MIMEO_PACK_MEMBER': '14500'}), ('unit_per_pair', {'MIMEO_PACK': '0'})):
        _clear(monkeypatch)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        outtab, outgff = str(tmp_path / (tag + '.tab')), str(tmp_path / (tag + '.gff3'))
        workflow.self_repeats(A, pairs, outtab, outgff, minIdt=80, minLen=100, minCov=3, label='Self_Repeat', prefix='Self_Repeat')
        st = eng.stats()
        assert (st['super_units'] > 0) == (tag == 'packed')
        outs[tag] = (open(outtab).read(), open(outgff).read())
    _clear(monkeypatch)
    assert outs['packed'] == outs['unit_per_pair']
    assert outs['packed'][0].count('\n') > 100 and outs['packed'][1].count('\n') > 3
    A.close()


def test_cli_self_two_ranks_on_a_fragmented_genome(tmp_path):
    """`mimeo self` under torchrun (two gloo ranks sharing GPU 0) on a fragmented genome: every rank takes some targets
    against all scaffolds (dist.shard_pairs_by_target) — a full cross product, packed — and rank 0 writes the same TAB /
    GFF3 as a single process."""
    import os
    import socket
    import subprocess
    import sys
    from mimeo_amd.synth import write_fasta
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names, seqs = synth_genome(41, 36 * 12_000, 36, repeat_frac=0.2, families=5, cons_len=(200, 1200), max_div=0.1)
    fa = str(tmp_path / 'frag.fa')
    write_fasta(fa, names, seqs)
    args = ['self', '--afasta', fa, '--minIdt', '80', '--minLen', '100', '--minCov', '3']
    env1 = {k: v for k, v in os.environ.items() if not k.startswith('MIMEO_PACK')}
    one = subprocess.run([sys.executable, '-m', 'mimeo_amd'] + args + ['-d', str(tmp_path / 'one')], cwd=root, capture_output=True, text=True,
                         timeout=600, env=env1)
    assert one.returncode == 0, one.stdout + one.stderr
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env2 = dict(env1, MIMEO_DIST_BACKEND='gloo', MIMEO_FORCE_DEVICE='0')
    two = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), '-m', 'mimeo_amd'] + args + ['-d', str(tmp_path / 'two')], cwd=root, capture_output=True,
                         text=True, timeout=900, env=env2)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-3000:]
    for f in ('mimeo_alignment.tab', 'mimeo-self_repeats.gff3', 'A_gen_lens.txt'):
        a, b = (tmp_path / 'one' / f).read_text(), (tmp_path / 'two' / f).read_text()
        assert a == b, f
    assert (tmp_path / 'one' / 'mimeo_alignment.tab').read_text().count('\n') > 50

"""End to end: `mimeo self | x | map` through the HIP engine must write byte-identical TAB / GFF3
to the oracle pipeline (C alignment oracle -> Python restatement of the reference's text stages)."""
import os

import numpy as np
import pytest

from mimeo_amd.synth import make_families, synth_genome

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from mimeo_amd import engine
    engine.init(0)
    return engine


def _oracle_tab(an, aseq, bn, bseq, pairs, min_len, min_idt, select=None):
    from oracle import oracle as O, pipeline as P
    lines = ['#name1\tstrand1\tstart1\tend1\tname2\tstrand2\tstart2+\tend2+\tscore\tidentity']
    for t, q in pairs:
        if select and not select((t, q)):
            continue
        al = O.align_pair(aseq[t].tobytes(), bseq[q].tobytes())
        al['tid'], al['qid'] = t, q
        text = '\n'.join(P.general_rows(an, bn, [len(s) for s in bseq], al)) + '\n'
        lines += P.filter_project_sort(text, min_len, min_idt)
    return lines


def _oracle_gff(tab_lines, names, seqs, cov, min_len, source, label, prefix):
    from oracle import pipeline as P
    bed = P.bed_project_sort(tab_lines)
    iv = [(l.split('\t')[0], int(l.split('\t')[1]), int(l.split('\t')[2])) for l in bed]
    regs = P.coverage_collapse(iv, {n: len(s) for n, s in zip(names, seqs)}, cov, min_len)
    return P.gff_self_lines(regs, label, prefix, source=source)


def test_self_workflow_files_match_oracle(eng, tmp_path):
    from mimeo_amd import workflow
    names, seqs = synth_genome(50, 300_000, 3, repeat_frac=0.2, families=3, cons_len=(300, 1500), max_div=0.1)
    names = ['s2', 's10', 's1']  # C-locale order differs from FASTA order
    A = eng.Genome(names, seqs)
    pairs = workflow.all_pairs(3)
    outtab, outgff = str(tmp_path / 'o.tab'), str(tmp_path / 'o.gff3')
    workflow.self_repeats(A, pairs, outtab, outgff, minIdt=80, minLen=100, minCov=3, label='Self_Repeat', prefix='Self_Repeat')
    exp_tab = _oracle_tab(names, seqs, names, seqs, pairs, 100, 80)
    assert open(outtab).read() == '\n'.join(exp_tab) + '\n'
    exp_gff = _oracle_gff(exp_tab, names, seqs, 3, 100, 'mimeo-self', 'Self_Repeat', 'Self_Repeat')
    assert len(exp_gff) > 2
    assert open(outgff).read() == '\n'.join(exp_gff) + '\n'
    # --recycle re-enters at the collapse with the same result
    os.remove(outgff)
    workflow.self_repeats(A, pairs, outtab, outgff, minIdt=80, minLen=100, minCov=3, reuseTab=True, label='Self_Repeat', prefix='Self_Repeat')
    assert open(outgff).read() == '\n'.join(exp_gff) + '\n'
    A.close()


def test_strict_self_workflow(eng, tmp_path):
    from mimeo_amd import workflow
    names, seqs = synth_genome(52, 200_000, 2, repeat_frac=0.25, families=2, cons_len=(300, 1200), max_div=0.08)
    A = eng.Genome(names, seqs)
    pairs = workflow.all_pairs(2)
    outtab, outgff = str(tmp_path / 'o.tab'), str(tmp_path / 'o.gff3')
    workflow.self_repeats(A, pairs, outtab, outgff, minIdt=60, minLen=100, minCov=2, intraCov=1, splitSelf=True,
                          label='SR', prefix='SR')
    inter = _oracle_tab(names, seqs, names, seqs, pairs, 100, 60, select=lambda p: p[0] != p[1])
    intra = _oracle_tab(names, seqs, names, seqs, pairs, 100, 60, select=lambda p: p[0] == p[1])
    assert open(outtab).read() == '\n'.join(inter) + '\n'
    assert open(outtab + '_intra.tab').read() == '\n'.join(intra) + '\n'
    g1 = _oracle_gff(inter, names, seqs, 2, 100, 'mimeo-self', 'SR', 'SR')
    g2 = _oracle_gff(intra, names, seqs, 1, 100, 'mimeo-self', 'SR_intra', 'SR')
    assert open(outgff).read() == '\n'.join(g1 + g2[2:]) + '\n'
    A.close()


def test_x_and_map_workflows(eng, tmp_path):
    from mimeo_amd import formats, workflow
    from oracle import pipeline as P
    fams = make_families(77, 3, (300, 1500))
    an, aseq = synth_genome(61, 200_000, 2, repeat_frac=0.2, shared_families=fams, prefix='a')
    bn, bseq = synth_genome(62, 200_000, 2, repeat_frac=0.2, shared_families=fams, prefix='b')
    A, B = eng.Genome(an, aseq), eng.Genome(bn, bseq)
    pairs = workflow.all_pairs(2, 2)
    outtab, outgff = str(tmp_path / 'x.tab'), str(tmp_path / 'x.gff3')
    workflow.self_repeats(A, pairs, outtab, outgff, minIdt=60, minLen=100, minCov=2, label='B_Repeat', prefix='B_Repeat',
                          source='mimeo', B=B)
    exp_tab = _oracle_tab(an, aseq, bn, bseq, pairs, 100, 60)
    assert open(outtab).read() == '\n'.join(exp_tab) + '\n'
    exp_gff = _oracle_gff(exp_tab, an, aseq, 2, 100, 'mimeo', 'B_Repeat', 'B_Repeat')
    assert open(outgff).read() == '\n'.join(exp_gff) + '\n'
    # map: TAB then import_Align + GFF
    maptab = str(tmp_path / 'm.tab')
    workflow.map_hits(A, B, pairs, maptab, minIdt=90, minLen=100)
    exp_map = _oracle_tab(an, aseq, bn, bseq, pairs, 100, 90)
    assert open(maptab).read() == '\n'.join(exp_map) + '\n'
    rows = formats.import_align(formats.parse_tab(maptab), 'HGT', 100, 90)
    got = ''.join(formats.gff_map_lines(rows, formats.chromlens(an, aseq), 'BHit'))
    orows = P.import_align('\n'.join(exp_map), 'HGT', 100, 90)
    assert got == '\n'.join(P.gff_map_lines(orows, formats.chromlens(an, aseq), 'BHit')) + '\n'
    A.close()
    B.close()


def test_cli_self_end_to_end(eng, tmp_path):
    """`python -m mimeo_amd self --afasta ...` writes the three reference outputs."""
    import subprocess
    import sys
    from mimeo_amd.synth import write_fasta
    names, seqs = synth_genome(53, 160_000, 2, repeat_frac=0.2, families=2, cons_len=(300, 1200))
    fa = str(tmp_path / 'g.fa')
    write_fasta(fa, names, seqs)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-m', 'mimeo_amd', 'self', '--afasta', fa, '-d', str(tmp_path / 'out'), '--minIdt', '80',
                        '--minLen', '100', '--minCov', '2'], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = tmp_path / 'out'
    assert (out / 'A_gen_lens.txt').read_text() == ''.join('%s\t%d\n' % (n, len(s)) for n, s in zip(names, seqs))
    exp_tab = _oracle_tab(names, seqs, names, seqs, [(a, b) for a in range(2) for b in range(2)], 100, 80)
    assert (out / 'mimeo_alignment.tab').read_text() == '\n'.join(exp_tab) + '\n'
    assert (out / 'mimeo-self_repeats.gff3').read_text().startswith('##gff-version 3\n#seqid\tsource')


def test_bench_two_ranks_on_one_gpu_match_single_rank(tmp_path):
    """The N>1 path end to end: two ranks (gloo, sharing GPU 0) must report the same alignments and
    regions as one rank.  On the 8-GPU node the same code runs over RCCL with one GPU per rank."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    one = subprocess.run([sys.executable, 'bench.py', '--workload', 'small', '--steps', '1', '--warmup', '0', '--no-cpu-baseline'],
                         cwd=root, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    a = json.loads(one.stdout.strip().split('\n')[-1])
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MIMEO_DIST_BACKEND='gloo', MIMEO_FORCE_DEVICE='0')
    two = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), 'bench.py', '--gpus', '2', '--workload', 'small', '--steps', '1', '--warmup', '0'],
                         cwd=root, capture_output=True, text=True, timeout=900, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    b = json.loads([l for l in two.stdout.strip().split('\n') if l.startswith('{')][-1])
    assert b['n_gpus'] == 2 and b['result'] == a['result'] and b['scaling'] == 'strong'
    assert b['config']['pair_strands_rank0'] < a['config']['pair_strands_rank0']


def test_bench_row_mode_two_ranks_cover_the_same_rows(tmp_path):
    """Row mode (the default C4 bench line, here at test size): 6 steps of one rank and 3 steps of two
    ranks (gloo, sharing GPU 0) are the same six target rows — same records kept, same regions — and the
    line says weak scaling with twice the per-step value basis."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    one = subprocess.run([sys.executable, 'bench.py', '--workload', 'c4small', '--steps', '6', '--warmup', '0', '--no-cpu-baseline'],
                         cwd=root, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    a = json.loads(one.stdout.strip().split('\n')[-1])
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MIMEO_DIST_BACKEND='gloo', MIMEO_FORCE_DEVICE='0')
    two = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), 'bench.py', '--gpus', '2', '--workload', 'c4small', '--steps', '3', '--warmup', '0'],
                         cwd=root, capture_output=True, text=True, timeout=900, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    b = json.loads([l for l in two.stdout.strip().split('\n') if l.startswith('{')][-1])
    assert a['scaling'] == b['scaling'] == 'weak' and b['n_gpus'] == 2
    assert a['result']['records_kept'] > 0 and a['result']['regions'] > 0
    assert b['result'] == a['result']
    # a row of six scaffolds: 6 minus-strand units, its own plus-strand unit, and 2 or 3 shared plus-strand pairs in both orders
    assert a['config']['pair_strands_rank0'] in (11, 13) and b['config']['pair_strands_rank0'] in (11, 13)
    # every step rebuilt the indexes of its target (two strands), nothing else
    assert 0 < a['stage_ms_per_step_rank0']['ms_index'] < 0.5 * a['stage_ms_per_step_rank0']['ms_total']


def test_bench_under_rccl_single_rank(tmp_path):
    """The collectives of the N>1 path through the real RCCL backend: one rank under torchrun with
    MIMEO_DIST_FORCE=1 (two ranks cannot share a GPU under RCCL).  Device tensors, all_gather of
    sizes and payload, all_reduce(MAX) of the time and the barrier all run as they do on 8 GPUs."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MIMEO_DIST_BACKEND='nccl', MIMEO_DIST_FORCE='1')
    run = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
                          '127.0.0.1', '--master-port', str(port), 'bench.py', '--gpus', '1', '--workload', 'c4small', '--steps', '2',
                          '--warmup', '1', '--no-cpu-baseline'], cwd=root, capture_output=True, text=True, timeout=900, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    b = json.loads([l for l in run.stdout.strip().split('\n') if l.startswith('{')][-1])
    assert b['n_gpus'] == 1 and b['scaling'] == 'weak' and b['result']['records_kept'] > 0 and b['result']['regions'] > 0


def test_lastz_shim_runs_the_reference_invocation(eng, tmp_path):
    """The literal argv the reference builds for lastz (wrappers.py:1025-1037) -> 13-field general
    rows that the reference's own awk filter turns into the same TAB block as the oracle."""
    import subprocess
    import sys
    from mimeo_amd.synth import write_fasta
    from oracle import oracle as O, pipeline as P
    names, seqs = synth_genome(54, 160_000, 2, repeat_frac=0.2, families=2, cons_len=(300, 1200))
    ta, qa = str(tmp_path / 'scaf0000.fa'), str(tmp_path / 'scaf0001.fa')
    write_fasta(ta, names[:1], seqs[:1])
    write_fasta(qa, names[1:], seqs[1:])
    out = str(tmp_path / 'temp_q_onto_t_.tab')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    argv = [ta, qa, '--entropy', '--format=general:name1,strand1,start1,end1,length1,name2,strand2,start2+,end2+,length2,score,identity',
            '--markend', '--gfextend', '--chain', '--gapped', '--step=1', '--strand=both', '--hspthresh=3000',
            '--output=' + out, '--verbosity=0']
    r = subprocess.run([sys.executable, '-m', 'mimeo_amd.lastz_shim'] + argv, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    text = open(out).read()
    assert text.endswith('# lastz end-of-file\n')
    al = O.align_pair(seqs[0].tobytes(), seqs[1].tobytes())
    al['tid'], al['qid'] = 0, 0
    exp = P.general_rows(names[:1], names[1:], [len(seqs[1])], al)
    assert [l for l in text.split('\n') if l and not l.startswith('#')] == exp
    assert P.filter_project_sort(text, 100, 80) == P.filter_project_sort('\n'.join(exp), 100, 80)
    bad = subprocess.run([sys.executable, '-m', 'mimeo_amd.lastz_shim', ta, qa, '--seed=match12'], cwd=root, capture_output=True, text=True)
    assert bad.returncode != 0


def test_pipeline_modes_give_identical_alignments(eng, monkeypatch):
    """How a call is cut into batches must not change a single record: one batch for everything, one unit per
    batch, five units per batch, batches cut by expected seed hits, queues so small that every batch is repeated
    with larger ones, and the round-1 heavy kernels in the place of the fused one."""
    import hashlib
    n, s = synth_genome(77, 1_500_000, 3, repeat_frac=0.08, families=6, cons_len=(300, 2500))
    g = eng.Genome(n, s)
    pairs = [(t, q) for t in range(3) for q in range(3)]
    outs = {}
    knobs = ('MIMEO_BATCH_UNITS', 'MIMEO_BATCH_HITS', 'MIMEO_QUEUE_SHRINK', 'MIMEO_HEAVY', 'MIMEO_MIRROR', 'MIMEO_QUEUE_BUDGET_MB')
    for tag, env in (('default', {}), ('one_unit', {'MIMEO_BATCH_UNITS': '1'}), ('five_units', {'MIMEO_BATCH_UNITS': '5'}),
                     ('by_hits', {'MIMEO_BATCH_HITS': '6e5'}), ('rerun', {'MIMEO_QUEUE_SHRINK': '3000'}), ('v1', {'MIMEO_HEAVY': 'v1'}),
                     ('no_mirror', {'MIMEO_MIRROR': '0'}), ('no_mirror_one_unit', {'MIMEO_MIRROR': '0', 'MIMEO_BATCH_UNITS': '1'}),
                     ('split_by_memory', {'MIMEO_QUEUE_BUDGET_MB': '150'})):
        for k in knobs:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        a = eng.align_pairs(g, None, pairs)
        st = eng.stats()
        # the plus-strand units (t, q) and (q, t) share one seed scan (3 of the 18 units launch nothing and count no hits)
        hits = st['seed_hits']
        if 'no_mirror' not in tag:
            assert st['scan_launches'] == 15
            hits_shared = hits
        else:
            assert st['scan_launches'] == 18 and hits > hits_shared
        outs[tag] = (a.size, hashlib.md5(a.tobytes()).hexdigest(), st['hsps'])
        if tag == 'default':
            assert st['batches'] == 1 and st['queue_reruns'] == 0 and st['pair_strands'] == 18
        if tag == 'one_unit':
            assert st['batches'] == 15   # a unit and its mirror stay together
        if tag == 'no_mirror_one_unit':
            assert st['batches'] == 18
        if tag == 'split_by_memory':
            assert st['batches'] > 1     # the queues of the one batch did not fit the (pretended) free memory: cut in two, again and again
        if tag == 'five_units':
            assert st['batches'] == 4
        if tag == 'by_hits':
            assert 2 < st['batches'] < 18
        if tag == 'rerun':
            assert st['queue_reruns'] >= 1
    for k in knobs:
        monkeypatch.delenv(k, raising=False)
    assert len(set(outs.values())) == 1, outs
    assert outs['default'][0] > 10
    g.close()


def test_kept_seed_indexes_are_reused_and_change_nothing(eng):
    """mimeo_genome_keep_indexes: a job issued row by row (one call per target scaffold) gives the
    records of the single call, later calls build nothing, dropped scaffolds are rebuilt."""
    from mimeo_amd import workflow
    names, seqs = synth_genome(77, 400_000, 4, repeat_frac=0.2, families=3, cons_len=(300, 1500), max_div=0.1)
    seqs[2][1000:3000] = np.frombuffer(bytes(seqs[2][1000:3000]).lower(), dtype=np.uint8)  # a soft-masked target (own sv plane)
    A = eng.Genome(names, seqs)
    pairs = workflow.all_pairs(4)
    whole = eng.align_pairs(A, None, pairs)
    built_whole = eng.stats()['ms_index']
    assert built_whole > 0
    A.keep_indexes(True)
    rows = []
    for t in range(4):
        rows.append(eng.align_pairs(A, None, [p for p in pairs if p[0] == t]))
        if t > 0:  # every strand was a query of row 0; only a soft-masked target adds an index of its own
            assert (eng.stats()['ms_index'] == 0) == (t != 2)
    got = np.concatenate(rows)
    assert got.tobytes() == whole.tobytes()
    A.drop_indexes([1])
    again = eng.align_pairs(A, None, [p for p in pairs if p[0] == 1])
    assert eng.stats()['ms_index'] > 0
    assert again.tobytes() == rows[1].tobytes()
    A.drop_indexes()
    A.keep_indexes(False)
    assert eng.align_pairs(A, None, pairs).tobytes() == whole.tobytes()
    B = eng.Genome(names[:2], seqs[:2])  # cross-genome: both handles keep their own
    A.keep_indexes(True); B.keep_indexes(True)
    x1 = eng.align_pairs(A, B, [(0, 0), (1, 1)])
    x2 = eng.align_pairs(A, B, [(0, 0), (1, 1)])
    assert eng.stats()['ms_index'] == 0 and x1.tobytes() == x2.tobytes()
    with pytest.raises(RuntimeError):
        A.drop_indexes([99])
    A.close(); B.close()


def test_index_blocks_when_the_seed_indexes_do_not_fit(eng, monkeypatch):
    """A genome of many scaffolds cannot keep every seed index (64 MiB + 52 B/base per strand) on the device: the
    pair matrix is then cut into blocks whose indexes fit, rebuilt block by block.  Same records as one block."""
    from mimeo_amd import workflow
    names, seqs = synth_genome(123, 600_000, 6, repeat_frac=0.25, families=3, cons_len=(300, 1500), max_div=0.1)
    A = eng.Genome(names, seqs)
    pairs = workflow.all_pairs(6)
    whole = eng.align_pairs(A, None, pairs)
    assert eng.stats()['index_blocks'] == 1 and whole.size > 50
    monkeypatch.setenv('MIMEO_INDEX_BUDGET_MB', '300')   # room for about two target and one query scaffold (both strands)
    blocked = eng.align_pairs(A, None, pairs)
    st = eng.stats()
    assert st['index_blocks'] >= 9, st['index_blocks']
    assert blocked.tobytes() == whole.tobytes()
    sub = [(5, 0), (0, 5), (2, 2), (3, 1)]                # an arbitrary pair list, not a full matrix
    monkeypatch.delenv('MIMEO_INDEX_BUDGET_MB')
    ref = eng.align_pairs(A, None, sub)
    monkeypatch.setenv('MIMEO_INDEX_BUDGET_MB', '300')
    assert eng.align_pairs(A, None, sub).tobytes() == ref.tobytes()
    A.close()


def test_long_repeat_copies_match_the_oracle(eng):
    """Long repeat copies (2-9 kbp, 5 % divergence: follower chains of hundreds of seed hits per diagonal, walks far
    beyond the frame) through the whole pipeline against the oracle, pair by pair."""
    from oracle import oracle as O
    names, seqs = synth_genome(321, 900_000, 3, repeat_frac=0.3, families=3, cons_len=(2000, 9000), max_div=0.05)
    A = eng.Genome(names, seqs)
    pairs = [(0, 1), (1, 1), (2, 0)]
    whole = eng.align_pairs(A, None, pairs)
    assert whole.size > 20
    cols = ['tstart', 'tend', 'qstart', 'qend', 'score', 'id_n', 'id_d', 'qstrand']
    for t, q in pairs:
        exp = O.align_pair(seqs[t].tobytes(), seqs[q].tobytes())
        sub = whole[(whole['tid'] == t) & (whole['qid'] == q)]
        assert np.array_equal(np.sort(sub[cols], order=cols), np.sort(exp[cols], order=cols)), (t, q)
    A.close()

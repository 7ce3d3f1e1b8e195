"""N>1 path on CPU: target sharding and the all-gatherv of alignment records over gloo with
world_size 2 (the GPU path uses the same code over RCCL)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_pairs_by_target_covers_everything_once():
    from mimeo_amd.dist import shard_pairs_by_target
    pairs = [(t, q) for t in range(7) for q in range(7)]
    cost = {t: (t + 1) * 10 for t in range(7)}
    for world in (1, 2, 3, 8):
        shards = [shard_pairs_by_target(pairs, cost, world, r) for r in range(world)]
        flat = sorted(p for s in shards for p in s)
        assert flat == sorted(pairs)
        for s in shards:  # a target never splits across ranks
            for other in shards:
                if s is not other:
                    assert not ({t for t, _ in s} & {t for t, _ in other})
    two = [shard_pairs_by_target(pairs, cost, 2, r) for r in range(2)]
    loads = [sum(cost[t] for t in {t for t, _ in s}) for s in two]
    assert abs(loads[0] - loads[1]) <= max(cost.values())


def test_bench_rows_are_dealt_round_robin_and_cover_the_job():
    """bench.py row mode: in one step the N ranks hold N different target rows, S/N steps cover every row
    once, and the dealing of one rank is what N ranks do in N times fewer steps."""
    sys.path.insert(0, ROOT)
    import bench
    S = 100
    for world in (1, 2, 4, 8):
        steps = -(-S // world)
        seen = [bench.row_of(k, world, r, S) for k in range(steps) for r in range(world)]
        assert sorted(seen[:S]) == list(range(S))
        for k in range(steps):
            assert len({bench.row_of(k, world, r, S) for r in range(world)}) == world
    assert [bench.row_of(k, 1, 0, 6) for k in range(6)] == [bench.row_of(k // 2, 2, k % 2, 6) for k in range(6)]


def test_bench_roofline_blocks_read_the_committed_counter_files(monkeypatch):
    """bench.py prices K34's launch with counters kept under profiles/ (SQ_INSTS_VALU, the ISA opcode mix, FETCH / WRITE_SIZE): the
    files the shipped form needs must be there and make sense — a missing one would turn `roofline.valu` into an error string on the
    GPU box.  0.905 ms per C4 unit is the measured launch time of the default form, 0.974 that of the lane-major form."""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.delenv('MIMEO_K34_FORM', raising=False)
    v = bench.valu_issue(0.905)
    assert 'error' not in v and 4.0e8 < v['insts_per_launch'] < 5.0e8 and 3.5 < v['cycles_per_inst_mix'] < 4.0
    assert 0.7 < v['frac_of_issue_peak'] < 0.85 and 'r03b_pmc_k34_sq.json' in v['source']
    monkeypatch.setenv('MIMEO_K34_FORM', 'lane')
    w = bench.valu_issue(0.974)
    assert 'error' not in w and w['insts_per_launch'] > v['insts_per_launch'] and 0.75 < w['frac_of_issue_peak'] < 0.9
    monkeypatch.setenv('MIMEO_K34_FORM', 'cut')
    assert 'error' in bench.valu_issue(0.9)      # no counters on file for that form: said, not guessed
    for wl in ('c4', 'c2'):
        t = bench.pmc_traffic(wl)
        assert isinstance(t, int) and 3e8 < t < 3e9
    assert bench.pmc_traffic('c3') is None


def test_bench_a11_filter_matches_the_printed_identity_rule():
    """The vectorised A11 filter of bench.py keeps exactly the records formats.tab_block keeps
    (length1 >= minLen and the PRINTED one-decimal identity >= minIdt, wrappers.py:1043-1052)."""
    sys.path.insert(0, ROOT)
    import bench
    from mimeo_amd import _ffi, formats
    rng = np.random.default_rng(5)
    a = np.zeros(4000, dtype=_ffi.ALIGNMENT)
    a['tstart'] = rng.integers(0, 10_000, a.size)
    a['tend'] = a['tstart'] + rng.integers(1, 400, a.size)
    a['id_d'] = rng.integers(0, 3000, a.size)
    a['id_n'] = (a['id_d'] * rng.uniform(0.7, 1.0, a.size)).astype(np.uint32)
    a['id_n'][:200] = (a['id_d'][:200] * 0.7995).astype(np.uint32)   # near the 79.95 / 80.0 rounding edge
    a['score'] = np.arange(a.size)
    got = bench.a11_filter(a, 100, 80)
    exp = [int(r['score']) for r in a if int(r['tend']) - int(r['tstart']) >= 100
           and float(formats.identity_pct(int(r['id_n']), int(r['id_d']))) >= 80]
    assert list(got['score']) == exp and 0 < len(exp) < a.size


WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    from mimeo_amd import _ffi
    from mimeo_amd.dist import Dist
    d = Dist().init('gloo')
    rng = np.random.default_rng(d.rank)
    n = [3, 0][d.rank] if os.environ.get('EMPTY_RANK1') else 5 + 4 * d.rank
    a = np.zeros(n, dtype=_ffi.ALIGNMENT)
    a['tid'] = d.rank
    a['tstart'] = np.arange(n)
    a['score'] = 1000 * (d.rank + 1) + np.arange(n)
    g = d.allgather_records(a)
    exp_n = 3 if os.environ.get('EMPTY_RANK1') else 5 + 9
    assert g.size == exp_n, g.size
    assert list(g['tid'][: (3 if os.environ.get('EMPTY_RANK1') else 5)]) == [0] * (3 if os.environ.get('EMPTY_RANK1') else 5)
    if not os.environ.get('EMPTY_RANK1'):
        assert list(g['score'][5:]) == [2000 + i for i in range(9)]
    assert d.max_float(float(d.rank)) == 1.0 and d.sum_int(d.rank + 1) == 3
    d.barrier()
    print('rank', d.rank, 'ok')
''')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_world2(extra_env=None):
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, '-c', WORKER % ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_allgather_records_gloo_world2():
    _run_world2()


def test_allgather_records_with_empty_rank():
    _run_world2({'EMPTY_RANK1': '1'})

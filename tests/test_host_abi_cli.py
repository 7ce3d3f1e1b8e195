"""The C-ABI library loads and exports every symbol include/mimeo_hip.h declares (no compute
without a GPU), fails loudly without a device, and the CLI mirrors the reference's flags."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from mimeo_amd import _ffi
    if not os.path.exists(_ffi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _ffi.load()


def test_header_symbols_are_exported():
    lib = _lib()
    hdr = open(os.path.join(ROOT, 'include', 'mimeo_hip.h')).read()
    declared = set(re.findall(r'\b(mimeo_[a-z_]+)\s*\(', hdr))
    from mimeo_amd import _ffi
    assert declared == set(_ffi.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.mimeo_abi_version() == _ffi.ABI_VERSION == 3


def test_struct_layouts_match_header():
    from mimeo_amd import _ffi
    assert C.sizeof(_ffi.Params) == 64
    assert _ffi.SEED_HIT.itemsize == 8 and _ffi.HSP.itemsize == 32
    assert _ffi.ALIGNMENT.itemsize == 48 and _ffi.INTERVAL.itemsize == 12
    assert C.sizeof(_ffi.Stats) == 8 * 8 + 7 * 8 + 8 + 8 + 7 * 8


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    from mimeo_amd import engine
    with pytest.raises(RuntimeError):
        engine.init(0)
    lib = _lib()
    p = C.c_void_p()
    n = C.c_uint64()
    from mimeo_amd import _ffi
    prm = _ffi.Params()
    assert lib.mimeo_params_default(C.byref(prm)) == 0 and prm.hspthresh == 3000 and prm.xdrop == 910
    assert lib.mimeo_coverage_collapse(None, 0, None, 0, 3, 100, C.byref(p), C.byref(n)) == -2  # MIMEO_ERR_NO_DEVICE
    assert b'no CPU fallback' in lib.mimeo_last_error() or b'mimeo_init' in lib.mimeo_last_error()


def test_cli_flags_and_defaults_mirror_reference():
    from mimeo_amd import run_interspecies, run_map, run_self
    a = run_self.mainArgs(['--afasta', 'g.fa'])
    assert (a.minIdt, a.minLen, a.minCov, a.hspthresh, a.intraCov) == (60, 100, 3, 3000, 5)
    assert (a.gffout, a.outfile, a.label, a.prefix) == ('mimeo-self_repeats.gff3', 'mimeo_alignment.tab', 'Self_Repeat', 'Self_Repeat')
    assert not a.strictSelf and not a.recycle and a.lzpath == 'lastz' and a.bedtools == 'bedtools'
    x = run_interspecies.mainArgs(['--afasta', 'a.fa', '--bfasta', 'b.fa'])
    assert (x.minCov, x.gffout, x.label, x.prefix) == (5, 'mimeo_B_in_A.gff3', 'B_Repeat', 'B_Repeat')
    m = run_map.mainArgs(['--afasta', 'a.fa', '--bfasta', 'b.fa', '--minIdt', '98'])
    assert (m.minIdt, m.gffout, m.label, m.prefix, m.maxtandem, m.tminscore) == (98, None, 'BHit', 'BHit', None, 50)
    with pytest.raises(SystemExit):
        run_self.mainArgs(['--minIdt', '80.5'])  # minIdt is an int flag in the reference


def test_all_pairs_order():
    from mimeo_amd import workflow
    assert workflow.all_pairs(2) == [(0, 0), (0, 1), (1, 0), (1, 1)]
    assert workflow.all_pairs(2, 3) == [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (1, 2)]


def test_crosscheck_lastz_probe_and_row_diff(tmp_path):
    """scripts/crosscheck_lastz.py: without a lastz binary it reports 'parity unpinned' and exits 0; its row
    diff tells identical rows from rows that only agree on coordinates; the argv is the reference's."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'scripts'))
    import crosscheck_lastz as X
    found = X.probe(str(tmp_path / 'no_such_lastz'))
    assert found['lastz'] is None and found['status'] == 'lastz absent: parity unpinned'
    r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'crosscheck_lastz.py'), '--lzpath', str(tmp_path / 'nope')],
                       capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == 'lastz absent: parity unpinned'
    argv = X.reference_argv('t.fa', 'q.fa', 'o.tab')
    assert argv[:3] == ['t.fa', 'q.fa', '--entropy'] and '--hspthresh=3000' in argv and argv[-2:] == ['--output=o.tab', '--verbosity=0']
    row = lambda s1, e1, s2, e2, sc, idt: '\t'.join(['t', '+', str(s1), str(e1), str(e1 - s1 + 1), 'q', '+', str(s2), str(e2), str(e2 - s2 + 1), str(sc), '90/100', idt])
    real = [row(1, 100, 5, 104, 9000, '90.0%'), row(200, 300, 7, 107, 8000, '90.0%'), row(400, 500, 1, 101, 7000, '90.0%')]
    ours = [row(1, 100, 5, 104, 9000, '90.0%'), row(200, 300, 7, 107, 8100, '90.0%'), row(600, 700, 1, 101, 7000, '90.0%')]
    d = X.compare(real, ours)
    assert d == {'rows_lastz': 3, 'rows_ours': 3, 'identical': 1, 'same_coordinates_other_score': 1, 'only_lastz': 1, 'only_ours': 1}


def test_lastz_probe_does_not_take_the_repos_own_shim_for_the_reference_tool(monkeypatch):
    """scripts/lastz is this repo's drop-in for the reference's --lzpath: with it on PATH (as INTEGRATION.md suggests) the
    A/B cross-check must still say that no real lastz is there — comparing the engine with itself is no parity evidence."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import crosscheck_lastz
    monkeypatch.setenv('PATH', os.path.join(ROOT, 'scripts') + os.pathsep + os.environ.get('PATH', ''))
    for arg in ('lastz', os.path.join(ROOT, 'scripts', 'lastz')):
        found = crosscheck_lastz.probe(arg)
        assert found['lastz'] is None and found['bedtools'] is None and 'parity unpinned' in found['status']


def test_unit_dealing_names_every_unit_once():
    """dist.units_of_row / deal_units: the rows of a self job name each of the 2 S^2 (target, query, strand) units exactly once,
    an unordered pair's plus-strand unit sits in one row in both orders, rows carry nearly equal numbers of them, and the
    ranks' shares are disjoint and complete."""
    from mimeo_amd import dist
    for S in (1, 2, 5, 6, 10, 100):
        seen = {}
        per_row = []
        for t in range(S):
            units = dist.units_of_row(t, S)
            owned = 0
            for a, b, m in units:
                for bit in (1, 2):
                    if m & bit:
                        assert (a, b, bit) not in seen
                        seen[(a, b, bit)] = t
                if a != t:   # a transposed plus-strand unit: its partner is in the same row
                    assert m == 1 and (t, a, 1) in seen and seen[(t, a, 1)] == t
                    owned += 1
            per_row.append(owned)
        assert len(seen) == 2 * S * S
        assert max(per_row) - min(per_row) <= 1 and sum(per_row) == (S * S - S) // 2
    S, W = 10, 4
    cost = {t: 1 + t % 3 for t in range(S)}
    shares = [dist.deal_units(S, cost, W, r) for r in range(W)]
    allu = [u for sh in shares for u in sh]
    named = {(a, b, bit) for a, b, m in allu for bit in (1, 2) if m & bit}
    assert len(named) == 2 * S * S and sum(bin(m).count('1') for _, _, m in allu) == 2 * S * S

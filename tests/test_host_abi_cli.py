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
    assert lib.mimeo_abi_version() == 1


def test_struct_layouts_match_header():
    from mimeo_amd import _ffi
    assert C.sizeof(_ffi.Params) == 64
    assert _ffi.SEED_HIT.itemsize == 8 and _ffi.HSP.itemsize == 32
    assert _ffi.ALIGNMENT.itemsize == 48 and _ffi.INTERVAL.itemsize == 12
    assert C.sizeof(_ffi.Stats) == 8 * 8 + 7 * 8 + 8 + 8 + 6 * 8


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

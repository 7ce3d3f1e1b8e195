"""ctypes loader for the C oracle (oracle/mimeo_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by mimeo_amd/.  PARITY UNPINNED for the alignment stages (see the
header of mimeo_oracle.c).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_build', 'libmimeo_oracle.so')


class Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('hspthresh', 'xdrop', 'ydrop', 'gap_open', 'gap_extend',
                                         'transitions', 'entropy', 'chain', 'gapped', 'strand')] + \
               [('reserved', C.c_int32 * 6)]


HIT = np.dtype([('tpos', '<u4'), ('qpos', '<u4')])
HSP = np.dtype([('tstart', '<u4'), ('qstart', '<u4'), ('length', '<u4'), ('flags', '<u4'),
                ('score', '<i8'), ('raw_score', '<i8')])
ALN = np.dtype([('tid', '<u4'), ('qid', '<u4'), ('tstart', '<u4'), ('tend', '<u4'), ('qstart', '<u4'),
                ('qend', '<u4'), ('score', '<i8'), ('id_n', '<u4'), ('id_d', '<u4'), ('qstrand', '<u4'),
                ('reserved', '<u4')])

_lib = None


def build():
    subprocess.check_call(['make', '-s', '-C', HERE])


def use_native_build():
    """bench.py's cpu_baseline leg: rebuild with -O3 -march=native on the box that runs it and load
    that library from now on; False (and the portable -O3 build stays) when no compiler is there."""
    global _lib
    native = os.path.join(HERE, '_build', 'libmimeo_oracle_native.so')
    try:
        subprocess.check_call(['make', '-s', '-C', HERE, 'native'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        return False
    _lib = None
    lib(native)
    return True


def lib(path=None):
    global _lib
    if _lib is None:
        if path is None and not os.path.exists(LIB):
            build()
        _lib = C.CDLL(path or LIB)
        _lib.orc_free.argtypes = [C.c_void_p]
        for fn in (_lib.orc_seed_hits, _lib.orc_ungapped_hsps):
            fn.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_int, C.POINTER(Params),
                           C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
            fn.restype = C.c_int
        _lib.orc_align_pair.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.POINTER(Params),
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        _lib.orc_align_pair.restype = C.c_int
    return _lib


def default_params(**kw):
    p = Params()
    lib().orc_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _take(ptr, n, dtype):
    n = int(n.value)
    if n == 0 or not ptr.value:
        if ptr.value:
            lib().orc_free(ptr)
        return np.zeros(0, dtype)
    buf = (C.c_char * (n * dtype.itemsize)).from_address(ptr.value)
    out = np.frombuffer(buf, dtype=dtype, count=n).copy()
    lib().orc_free(ptr)
    return out


def _b(s):
    return s if isinstance(s, (bytes, bytearray)) else bytes(s)


def seed_hits(T, Q, minus=0, params=None):
    p = params or default_params()
    ptr, n = C.c_void_p(), C.c_uint64()
    T, Q = _b(T), _b(Q)
    rc = lib().orc_seed_hits(T, len(T), Q, len(Q), int(minus), C.byref(p), C.byref(ptr), C.byref(n))
    assert rc == 0
    return _take(ptr, n, HIT)


def ungapped_hsps(T, Q, minus=0, params=None):
    p = params or default_params()
    ptr, n = C.c_void_p(), C.c_uint64()
    T, Q = _b(T), _b(Q)
    rc = lib().orc_ungapped_hsps(T, len(T), Q, len(Q), int(minus), C.byref(p), C.byref(ptr), C.byref(n))
    assert rc == 0
    return _take(ptr, n, HSP)


def chain_hsps(hsps):
    """the chain stage alone (mimeo_oracle.c: chain_hsps, O(n^2)): the HSPs sorted by (tstart, qstart, length), flags bit 0 = chained"""
    h = np.ascontiguousarray(hsps, dtype=HSP).copy()
    rc = lib().orc_chain_hsps(h.ctypes.data_as(C.c_void_p), C.c_uint64(h.size))
    assert rc == 0
    return h


def align_pair(T, Q, params=None):
    p = params or default_params()
    ptr, n = C.c_void_p(), C.c_uint64()
    T, Q = _b(T), _b(Q)
    rc = lib().orc_align_pair(T, len(T), Q, len(Q), C.byref(p), C.byref(ptr), C.byref(n))
    assert rc == 0
    return _take(ptr, n, ALN)

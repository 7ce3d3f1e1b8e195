"""Host-side handle on the HIP engine: what the reference's `*_LZ_cmds` + `run_cmd`
(src/mimeo/wrappers.py:899-1271, src/mimeo/utils.py:213-254) become once the shell pipeline
is replaced by calls through the C-ABI."""
import ctypes as C
import os

import numpy as np

from . import _ffi

_initialised_device = None


def init(device=0):
    """Bind this process to one GPU (one process per GPU)."""
    global _initialised_device
    lib = _ffi.load()
    _ffi.check(lib.mimeo_init(int(device)))
    _initialised_device = int(device)


def default_params(**kw):
    p = _ffi.Params()
    _ffi.check(_ffi.load().mimeo_params_default(C.byref(p)))
    for k, v in kw.items():
        setattr(p, k, int(v))
    return p


def stats():
    s = _ffi.Stats()
    _ffi.check(_ffi.load().mimeo_get_stats(C.byref(s)))
    return s.asdict()


class Genome:
    """Device-resident scaffolds (both strands, bit-plane packed)."""

    def __init__(self, names, seqs):
        """names: list[str]; seqs: list of bytes / uint8 arrays of ASCII bases."""
        lib = _ffi.load()
        self.names = list(names)
        arrs = [np.frombuffer(s, dtype=np.uint8) if isinstance(s, (bytes, bytearray)) else np.ascontiguousarray(s, dtype=np.uint8)
                for s in seqs]
        self.lengths = [int(a.size) for a in arrs]
        offsets = np.zeros(len(arrs) + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum(self.lengths, dtype=np.uint64)
        bases = np.concatenate(arrs) if arrs else np.zeros(0, dtype=np.uint8)
        if bases.size == 0:
            bases = np.zeros(1, dtype=np.uint8)
        h = C.c_void_p()
        _ffi.check(lib.mimeo_genome_create(len(arrs), bases.ctypes.data, offsets.ctypes.data, C.byref(h)))
        self._h = h

    @classmethod
    def from_fasta(cls, paths, split_dir=None):
        """Stream FASTA file(s) from disk straight into device memory (native parser thread + K1;
        replaces utils.py:274-309 splitFasta and the Biopython parses).  `split_dir`: also leave one
        `<id>.fa` per record there, like the reference's --adir/--bdir."""
        lib = _ffi.load()
        paths = [paths] if isinstance(paths, (str, bytes)) else list(paths)
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        h = C.c_void_p()
        _ffi.check(lib.mimeo_genome_load_fasta(arr, len(paths), os.fsencode(split_dir) if split_dir else None, C.byref(h)))
        self = cls.__new__(cls)
        self._h = h
        n = C.c_uint32()
        _ffi.check(lib.mimeo_genome_nscaf(h, C.byref(n)))
        self.names, self.lengths = [], []
        for i in range(n.value):
            nm, ln = C.c_char_p(), C.c_uint64()
            _ffi.check(lib.mimeo_genome_name(h, i, C.byref(nm)))
            _ffi.check(lib.mimeo_genome_length(h, i, C.byref(ln)))
            self.names.append(nm.value.decode())
            self.lengths.append(int(ln.value))
        return self

    def keep_indexes(self, keep=True):
        """Seed indexes built by later align_pairs calls stay attached to this genome and are reused
        (lastz rebuilds the table in every run, wrappers.py:1028-1031); results do not change."""
        _ffi.check(_ffi.load().mimeo_genome_keep_indexes(self._h, 1 if keep else 0))

    def build_indexes(self, scaffolds=None):
        """Build and keep the seed indexes of the listed scaffolds now (None = all)."""
        ids = np.ascontiguousarray([] if scaffolds is None else list(scaffolds), dtype=np.uint32)
        if scaffolds is not None and ids.size == 0:
            return
        _ffi.check(_ffi.load().mimeo_genome_build_indexes(self._h, ids.ctypes.data if ids.size else None, ids.size))

    def drop_indexes(self, scaffolds=None):
        """Release the kept seed indexes (both strands) of the listed scaffolds; None = all."""
        ids = np.ascontiguousarray([] if scaffolds is None else list(scaffolds), dtype=np.uint32)
        if scaffolds is not None and ids.size == 0:
            return
        _ffi.check(_ffi.load().mimeo_genome_drop_indexes(self._h, ids.ctypes.data if ids.size else None, ids.size))

    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            _ffi.load().mimeo_genome_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return len(self.names)


def seed_hits(T, tid, Q, qid, qstrand=0, params=None):
    p = params or default_params()
    ptr, n = C.c_void_p(), C.c_uint64()
    _ffi.check(_ffi.load().mimeo_seed_hits(T._h, tid, Q._h, qid, int(qstrand), C.byref(p), C.byref(ptr), C.byref(n)))
    return _ffi.take(ptr, n, _ffi.SEED_HIT)


def ungapped_hsps(T, tid, Q, qid, qstrand=0, params=None):
    p = params or default_params()
    ptr, n = C.c_void_p(), C.c_uint64()
    _ffi.check(_ffi.load().mimeo_ungapped_hsps(T._h, tid, Q._h, qid, int(qstrand), C.byref(p), C.byref(ptr), C.byref(n)))
    return _ffi.take(ptr, n, _ffi.HSP)


def chain_hsps(hsps):
    """lastz --chain on the HSPs of one (target, query, strand): the HSPs in (tstart, qstart, length) order, bit 0 of
    `flags` set on the members of the chain (include/mimeo_hip.h: mimeo_chain_hsps; a test entry like ungapped_hsps)."""
    h = np.ascontiguousarray(hsps, dtype=_ffi.HSP)
    out = np.zeros(h.size, dtype=_ffi.HSP)
    _ffi.check(_ffi.load().mimeo_chain_hsps(h.ctypes.data, h.size, out.ctypes.data))
    return out


def align_pair(T, tid, Q, qid, params=None):
    p = params or default_params()
    ptr, n = C.c_void_p(), C.c_uint64()
    _ffi.check(_ffi.load().mimeo_align_pair(T._h, tid, Q._h, qid, C.byref(p), C.byref(ptr), C.byref(n)))
    return _ffi.take(ptr, n, _ffi.ALIGNMENT)


def align_pairs(A, B, pairs, params=None):
    """pairs: iterable of (target index in A, query index in B or A)."""
    p = params or default_params()
    pr = np.asarray(list(pairs), dtype=np.uint32).reshape(-1, 2)
    pt = np.ascontiguousarray(pr[:, 0])
    pq = np.ascontiguousarray(pr[:, 1])
    ptr, n = C.c_void_p(), C.c_uint64()
    _ffi.check(_ffi.load().mimeo_align_pairs(A._h, B._h if B is not None else None, pt.ctypes.data, pq.ctypes.data,
                                             len(pt), C.byref(p), C.byref(ptr), C.byref(n)))
    return _ffi.take(ptr, n, _ffi.ALIGNMENT)


def align_units(A, B, units, params=None):
    """units: iterable of (target index in A, query index in B or A, strands) with strands = 1 plus, 2 minus, 3 both
    (masked with params.strand): mimeo_align_units — a rank's share of a sharded self job (dist.deal_units)."""
    p = params or default_params()
    un = np.asarray(list(units), dtype=np.uint32).reshape(-1, 3)
    pt, pq = np.ascontiguousarray(un[:, 0]), np.ascontiguousarray(un[:, 1])
    ps = np.ascontiguousarray(un[:, 2].astype(np.uint8))
    ptr, n = C.c_void_p(), C.c_uint64()
    _ffi.check(_ffi.load().mimeo_align_units(A._h, B._h if B is not None else None, pt.ctypes.data, pq.ctypes.data, ps.ctypes.data,
                                             len(pt), C.byref(p), C.byref(ptr), C.byref(n)))
    return _ffi.take(ptr, n, _ffi.ALIGNMENT)


def failed_pairs():
    """Pairs of the last align_pairs / align_units call that hit a documented limit and were left out (the reference's script
    loses only the failing lastz run's rows: utils.py:125-128): [(index into the call's pair list, error code)]."""
    lib = _ffi.load()
    n = C.c_uint64()
    _ffi.check(lib.mimeo_get_failed_pairs(None, None, 0, C.byref(n)))
    if not n.value:
        return []
    idx = np.zeros(n.value, dtype=np.uint64)
    code = np.zeros(n.value, dtype=np.int32)
    _ffi.check(lib.mimeo_get_failed_pairs(idx.ctypes.data, code.ctypes.data, n.value, C.byref(n)))
    return list(zip(idx.tolist(), code.tolist()))


def last_error():
    return _ffi.load().mimeo_last_error().decode()


def coverage_collapse(intervals, chrom_len, min_cov, min_len):
    """intervals: structured array (_ffi.INTERVAL) or (n,3) ints of (chrom id, start, end)."""
    iv = np.asarray(intervals)
    if iv.dtype != _ffi.INTERVAL:
        a = np.asarray(iv, dtype=np.uint32).reshape(-1, 3)
        iv = np.zeros(a.shape[0], dtype=_ffi.INTERVAL)
        iv['chrom'], iv['start'], iv['end'] = a[:, 0], a[:, 1], a[:, 2]
    iv = np.ascontiguousarray(iv)
    cl = np.ascontiguousarray(chrom_len, dtype=np.uint32)
    ptr, n = C.c_void_p(), C.c_uint64()
    _ffi.check(_ffi.load().mimeo_coverage_collapse(iv.ctypes.data if iv.size else None, iv.size, cl.ctypes.data,
                                                   cl.size, int(min_cov), int(min_len), C.byref(ptr), C.byref(n)))
    return _ffi.take(ptr, n, _ffi.INTERVAL)


def coverage_bedgraph(intervals, chrom_len):
    """bedtools genomecov -bg: maximal runs of equal depth > 0 (records of _ffi.DEPTH_RUN, in chrom / start order)."""
    iv = np.asarray(intervals)
    if iv.dtype != _ffi.INTERVAL:
        a = np.asarray(iv, dtype=np.uint32).reshape(-1, 3)
        iv = np.zeros(a.shape[0], dtype=_ffi.INTERVAL)
        iv['chrom'], iv['start'], iv['end'] = a[:, 0], a[:, 1], a[:, 2]
    iv = np.ascontiguousarray(iv)
    cl = np.ascontiguousarray(chrom_len, dtype=np.uint32)
    ptr, n = C.c_void_p(), C.c_uint64()
    _ffi.check(_ffi.load().mimeo_coverage_bedgraph(iv.ctypes.data if iv.size else None, iv.size, cl.ctypes.data, cl.size,
                                                   C.byref(ptr), C.byref(n)))
    return _ffi.take(ptr, n, _ffi.DEPTH_RUN)


def tandem_masked(A, intervals, match=2, mismatch=7, minscore=50, maxperiod=50, delta=7):
    """Bases of each (scaffold id, start, end) slice of genome A marked by the tandem scorer (K8)."""
    iv = np.asarray(intervals)
    if iv.dtype != _ffi.INTERVAL:
        a = np.asarray(iv, dtype=np.uint32).reshape(-1, 3)
        iv = np.zeros(a.shape[0], dtype=_ffi.INTERVAL)
        iv['chrom'], iv['start'], iv['end'] = a[:, 0], a[:, 1], a[:, 2]
    iv = np.ascontiguousarray(iv)
    out = np.zeros(iv.size, dtype=np.uint32)
    if iv.size:
        _ffi.check(_ffi.load().mimeo_tandem_masked(A._h, iv.ctypes.data, iv.size, int(match), int(mismatch), int(delta),
                                                   int(minscore), int(maxperiod), out.ctypes.data))
    return out

"""ctypes binding of libmimeo_hip.so (include/mimeo_hip.h).

The product path has no CPU fallback: importing this module without the built library, or
calling any compute entry point without a gfx950 device, raises RuntimeError.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libmimeo_hip.so')
ABI_VERSION = 3

SYMBOLS = [
    'mimeo_abi_version', 'mimeo_init', 'mimeo_shutdown', 'mimeo_last_error', 'mimeo_params_default',
    'mimeo_get_stats', 'mimeo_free', 'mimeo_genome_create', 'mimeo_genome_destroy', 'mimeo_genome_nscaf',
    'mimeo_genome_length', 'mimeo_seed_hits', 'mimeo_ungapped_hsps', 'mimeo_align_pair', 'mimeo_align_pairs',
    'mimeo_coverage_collapse', 'mimeo_tandem_masked', 'mimeo_genome_load_fasta', 'mimeo_genome_name',
    'mimeo_genome_keep_indexes', 'mimeo_genome_drop_indexes', 'mimeo_genome_build_indexes', 'mimeo_coverage_bedgraph',
    'mimeo_align_units', 'mimeo_get_failed_pairs', 'mimeo_chain_hsps',
]


class Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('hspthresh', 'xdrop', 'ydrop', 'gap_open', 'gap_extend', 'transitions',
                                         'entropy', 'chain', 'gapped', 'strand')] + [('reserved', C.c_int32 * 6)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ('pair_strands', 'seed_hits', 'hsps', 'chained_hsps', 'alignments',
                                          'query_bases_scanned', 'scan_bytes_algorithmic', 'scan_bytes_kernel')] + \
               [(n, C.c_double) for n in ('ms_index', 'ms_scan', 'ms_extend', 'ms_chain', 'ms_gapped', 'ms_collapse',
                                          'ms_total')] + \
               [('scan_launches', C.c_uint64), ('ms_scan_fill', C.c_double), ('index_blocks', C.c_uint64), ('batches', C.c_uint64), ('queue_reruns', C.c_uint64), ('walked_hits', C.c_uint64), ('followers', C.c_uint64), ('super_units', C.c_uint64), ('scan_kernel_launches', C.c_uint64)]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != 'reserved'}


SEED_HIT = np.dtype([('tpos', '<u4'), ('qpos', '<u4')])
HSP = np.dtype([('tstart', '<u4'), ('qstart', '<u4'), ('length', '<u4'), ('flags', '<u4'), ('score', '<i8'),
                ('raw_score', '<i8')])
ALIGNMENT = np.dtype([('tid', '<u4'), ('qid', '<u4'), ('tstart', '<u4'), ('tend', '<u4'), ('qstart', '<u4'),
                      ('qend', '<u4'), ('score', '<i8'), ('id_n', '<u4'), ('id_d', '<u4'), ('qstrand', '<u4'),
                      ('reserved', '<u4')])
INTERVAL = np.dtype([('chrom', '<u4'), ('start', '<u4'), ('end', '<u4')])
DEPTH_RUN = np.dtype([('chrom', '<u4'), ('start', '<u4'), ('end', '<u4'), ('depth', '<u4')])

_lib = None


def load():
    """Load the shared library and declare signatures (no device needed for this)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                           '(the mimeo engine has no CPU fallback)' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    lib.mimeo_abi_version.restype = C.c_int
    lib.mimeo_init.argtypes = [C.c_int]
    lib.mimeo_last_error.restype = C.c_char_p
    lib.mimeo_params_default.argtypes = [C.POINTER(Params)]
    lib.mimeo_get_stats.argtypes = [C.POINTER(Stats)]
    lib.mimeo_free.argtypes = [vp]
    lib.mimeo_free.restype = None
    lib.mimeo_genome_create.argtypes = [u32, vp, vp, C.POINTER(vp)]
    lib.mimeo_genome_destroy.argtypes = [vp]
    lib.mimeo_genome_destroy.restype = None
    lib.mimeo_genome_nscaf.argtypes = [vp, C.POINTER(u32)]
    lib.mimeo_genome_length.argtypes = [vp, u32, C.POINTER(u64)]
    lib.mimeo_genome_load_fasta.argtypes = [C.POINTER(C.c_char_p), u32, C.c_char_p, C.POINTER(vp)]
    lib.mimeo_genome_name.argtypes = [vp, u32, C.POINTER(C.c_char_p)]
    lib.mimeo_genome_keep_indexes.argtypes = [vp, C.c_int]
    lib.mimeo_genome_drop_indexes.argtypes = [vp, vp, u64]
    lib.mimeo_genome_build_indexes.argtypes = [vp, vp, u64]
    for name in ('mimeo_seed_hits', 'mimeo_ungapped_hsps'):
        if hasattr(lib, name):
            getattr(lib, name).argtypes = [vp, u32, vp, u32, u32, C.POINTER(Params), C.POINTER(vp), C.POINTER(u64)]
    if hasattr(lib, 'mimeo_align_pair'):
        lib.mimeo_align_pair.argtypes = [vp, u32, vp, u32, C.POINTER(Params), C.POINTER(vp), C.POINTER(u64)]
    if hasattr(lib, 'mimeo_align_pairs'):
        lib.mimeo_align_pairs.argtypes = [vp, vp, vp, vp, u64, C.POINTER(Params), C.POINTER(vp), C.POINTER(u64)]
    if hasattr(lib, 'mimeo_align_units'):
        lib.mimeo_align_units.argtypes = [vp, vp, vp, vp, vp, u64, C.POINTER(Params), C.POINTER(vp), C.POINTER(u64)]
        lib.mimeo_get_failed_pairs.argtypes = [vp, vp, u64, C.POINTER(u64)]
    if hasattr(lib, 'mimeo_chain_hsps'):
        lib.mimeo_chain_hsps.argtypes = [vp, u64, vp]
    if hasattr(lib, 'mimeo_coverage_collapse'):
        lib.mimeo_coverage_collapse.argtypes = [vp, u64, vp, u32, u32, u32, C.POINTER(vp), C.POINTER(u64)]
    if hasattr(lib, 'mimeo_coverage_bedgraph'):
        lib.mimeo_coverage_bedgraph.argtypes = [vp, u64, vp, u32, C.POINTER(vp), C.POINTER(u64)]
    if hasattr(lib, 'mimeo_tandem_masked'):
        lib.mimeo_tandem_masked.argtypes = [vp, vp, u64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp]
    if lib.mimeo_abi_version() != ABI_VERSION:
        raise RuntimeError('libmimeo_hip.so ABI %d != expected %d' % (lib.mimeo_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise RuntimeError('libmimeo_hip: %s (code %d)' % (load().mimeo_last_error().decode(), rc))


def take(ptr, n, dtype):
    """Copy a library-owned array into numpy and free it."""
    lib = load()
    n = int(n.value)
    try:
        if n == 0 or not ptr.value:
            return np.zeros(0, dtype)
        buf = (C.c_char * (n * dtype.itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dtype, count=n).copy()
    finally:
        if ptr.value:
            lib.mimeo_free(ptr)

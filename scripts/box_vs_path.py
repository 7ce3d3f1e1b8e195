"""Study (CPU, oracle only): how often does the box rule of alignment specification v1 (an anchor inside the (t, q) BOX
of an earlier alignment of its (pair, strand) is skipped — DESIGN.md §2 rule 6) decide differently from LASTZ's documented
path rule (an anchor ON THE PATH of an earlier alignment is skipped; reference call site src/mimeo/wrappers.py:1031
`--gapped`), and what does it change downstream (alignment records, TAB rows kept by A11, regions of the coverage
collapse at --minIdt 80 --minLen 100 --minCov 3)?

C2-like synthetic genomes (mimeo_amd.synth: SURVEY §8d — 40 families, divergence U[0, 0.15], 0.5 % indels) at a size the
oracle finishes in minutes: S scaffolds of L bases, every ordered pair, both strands — the (A, A) plus strand included: its
chain is the whole-scaffold diagonal alone, so nothing is left to differ there.

    python scripts/box_vs_path.py [S] [L] [repeat_frac] [seed] [families] [tandem arrays per scaffold]   -> a JSON line
(tandem arrays: 8-30 diverged copies of a 150-900 bp unit in a row, the same few units in every scaffold — chained HSPs on
neighbouring diagonals inside one box are the case where the two rules could part)
"""
import ctypes as C
import json
import os
import subprocess
import sys
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mimeo_amd.synth import synth_genome  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import pipeline as P  # noqa: E402

LIB = os.path.join(ROOT, 'oracle', '_build', 'libmimeo_oracle_study.so')
_lib = None
_G = {}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle'), 'study'])
        _lib = C.CDLL(LIB)
        _lib.orc_align_pair_rule.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.POINTER(O.Params), C.c_int,
                                             C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        _lib.orc_align_pair_rule.restype = C.c_int
        _lib.orc_free.argtypes = [C.c_void_p]
    return _lib


def align_rule(T, Q, rule):
    p = O.default_params()
    ptr, n = C.c_void_p(), C.c_uint64()
    counts = (C.c_uint64 * 4)()
    rc = lib().orc_align_pair_rule(T, len(T), Q, len(Q), C.byref(p), rule, C.byref(ptr), C.byref(n), counts)
    assert rc == 0
    k = int(n.value)
    out = np.zeros(0, O.ALN)
    if k:
        out = np.frombuffer((C.c_char * (k * O.ALN.itemsize)).from_address(ptr.value), dtype=O.ALN, count=k).copy()
    if ptr.value:
        lib().orc_free(ptr)
    return out, list(counts)


def _init(seqs):
    _G['seqs'] = seqs


def _pair(tq):
    t, q = tq
    T, Q = _G['seqs'][t], _G['seqs'][q]
    box, cb = align_rule(T, Q, 0)
    ref = O.align_pair(T, Q)
    assert box.tobytes() == ref.tobytes(), 'the restated gapped stage must reproduce the oracle under the box rule'
    path, cp = align_rule(T, Q, 1)
    for a in (box, path):
        a['tid'], a['qid'] = t, q
    return t, q, box, path, cb, cp


def regions(names, seqs, alns, min_len=100, min_idt=80, min_cov=3):
    """A11 filter (length1 >= minLen, %.1f identity >= minIdt), BED projection on origin-one start1, collapse"""
    iv = []
    kept = 0
    for a in alns:
        length1 = int(a['tend']) - int(a['tstart'])
        pct = float('%.1f' % (100.0 * int(a['id_n']) / int(a['id_d']))) if a['id_d'] else 0.0
        if length1 >= min_len and pct >= min_idt:
            kept += 1
            iv.append((names[a['tid']], int(a['tstart']) + 1, int(a['tend'])))
    lens = {n: len(s) for n, s in zip(names, seqs)}
    return kept, P.coverage_collapse(iv, lens, min_cov, min_len)


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    L = int(float(sys.argv[2])) if len(sys.argv) > 2 else 500_000
    frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 50
    fams = int(sys.argv[5]) if len(sys.argv) > 5 else 40
    names, arrs = synth_genome(seed, S * L, S, repeat_frac=frac, families=fams)
    ntand = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    if ntand:
        rng = np.random.default_rng(seed + 7)
        units = [rng.integers(0, 4, size=int(rng.integers(150, 900)), dtype=np.uint8) for _ in range(3)]
        acgt = np.frombuffer(b'ACGT', np.uint8)
        arrs = [a.copy() for a in arrs]
        for a in arrs:
            for _ in range(ntand):
                u = units[int(rng.integers(0, 3))]
                copies = []
                for _c in range(int(rng.integers(8, 31))):
                    c = u.copy()
                    m = rng.random(c.size) < rng.random() * 0.12
                    c[m] = (c[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
                    if rng.random() < 0.3:   # an indel of 1-20 bases between copies
                        c = np.concatenate([c, rng.integers(0, 4, size=int(rng.integers(1, 21)), dtype=np.uint8)]) if rng.random() < 0.5 else c[:-int(rng.integers(1, 21))]
                    copies.append(c)
                arr = acgt[np.concatenate(copies)]
                if arr.size < a.size // 2:
                    pos = int(rng.integers(0, a.size - arr.size))
                    a[pos:pos + arr.size] = arr
    seqs = [a.tobytes() for a in arrs]
    lib()
    pairs = [(t, q) for t in range(S) for q in range(S)]
    with Pool(min(8, os.cpu_count() or 1), initializer=_init, initargs=(seqs,)) as pool:
        res = pool.map(_pair, pairs, chunksize=1)
    tot_b, tot_p = np.zeros(4, np.int64), np.zeros(4, np.int64)
    box_all, path_all, pairs_diff = [], [], 0
    for t, q, box, path, cb, cp in res:
        tot_b += cb
        tot_p += cp
        box_all.append(box)
        path_all.append(path)
        if box.tobytes() != path.tobytes():
            pairs_diff += 1
    box_all, path_all = np.concatenate(box_all), np.concatenate(path_all)
    kb, rb = regions(names, seqs, box_all)
    kp, rp = regions(names, seqs, path_all)
    sb, sp = set(rb), set(rp)
    cols = ['tid', 'qid', 'tstart', 'tend', 'qstart', 'qend', 'score', 'id_n', 'id_d', 'qstrand']
    setb = set(map(tuple, box_all[cols].tolist()))
    setp = set(map(tuple, path_all[cols].tolist()))
    print(json.dumps({
        'genome': {'scaffolds': S, 'scaffold_bp': L, 'repeat_frac': frac, 'families': fams, 'seed': seed, 'tandem_arrays_per_scaffold': ntand},
        'pair_strands': 2 * len(pairs), 'anchors': int(tot_b[0]),
        'box_rule': {'skipped': int(tot_b[1]), 'alignments': int(tot_b[3]), 'tab_rows_kept': kb, 'regions': len(rb)},
        'path_rule': {'skipped': int(tot_p[1]), 'alignments': int(tot_p[3]), 'tab_rows_kept': kp, 'regions': len(rp),
                      'anchors_in_a_box_but_off_every_path': int(tot_p[2])},
        'pairs_with_different_alignments': pairs_diff,
        'alignments_only_box': len(setb - setp), 'alignments_only_path': len(setp - setb),
        'regions_only_box': len(sb - sp), 'regions_only_path': len(sp - sb),
        'bases_in_regions': {'box': int(sum(e - s for _, s, e in rb)), 'path': int(sum(e - s for _, s, e in rp))},
    }))


if __name__ == '__main__':
    main()

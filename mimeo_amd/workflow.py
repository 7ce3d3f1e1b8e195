"""The three mimeo workflows with the shell pipeline replaced by engine calls.

Mirrors the reference's command builders + executor:
  self_repeats  <- wrappers.py:899-1271 self_LZ_cmds   + utils.py:213-254 run_cmd
  x_repeats     <- wrappers.py:683-896  xspecies_LZ_cmds
  map_hits      <- wrappers.py:525-680  map_LZ_cmds, :33-117 import_Align, :443-522 writeGFFlines
Same argument names and meaning as those functions where they still apply; `lzpath` /
`bdtlsPath` are gone because no external tool is run.
"""
import logging
import os

import numpy as np

from . import engine, formats
from .dist import Dist, deal_units, shard_pairs_by_target


def all_pairs(n_a, n_b=None):
    """utils.py:65-106 get_all_pairs: full ordered Cartesian product; self mode includes (A,A)
    and both orders.  The reference inherits the order of glob(); here it is FASTA order."""
    if n_b is None:
        return [(a, b) for a in range(n_a) for b in range(n_a)]
    return [(a, b) for a in range(n_a) for b in range(n_b)]


PACK_MEMBER_BP = 6 << 20   # the engine packs scaffolds of up to this size into super-scaffolds when there are at least ...
PACK_MIN = 8               # ... this many of them (mimeo_hip.h, mimeo_align_pairs)


def align_blocks(A, B, pairs, params, min_len, min_idt, dist=None):
    """Align every pair (sharded over ranks when dist.world > 1) and return {(t, q): [TAB lines]} on every rank, the raw
    record count and the (n, 4) array of (tid, qid, start1, end1) of the rows kept — what the BED projection of the TAB reads
    back (wrappers.py:1120-1128).

    Sharding (by target scaffold, no data-path collective).  A self job over the full pair matrix of large
    scaffolds deals UNITS (dist.deal_units): a rank gets its target rows' minus-strand units and the plus-strand units of the
    unordered pairs dealt to those rows, in both orders, so that the engine computes each plus-strand pair once.  Fragmented
    assemblies (the engine packs them into super-scaffolds when it is handed a full cross product) and two-genome jobs are
    sharded by target with whole pairs, as before."""
    dist = dist or Dist()
    QG = B if B is not None else A
    n = len(A.names)
    failed = []
    if dist.world > 1:
        small = sum(1 for ln in A.lengths if ln <= PACK_MEMBER_BP)
        full_self = B is None and len(pairs) == n * n and len(set(pairs)) == n * n
        if full_self and small < PACK_MIN:
            cost = {t: A.lengths[t] * sum(A.lengths) for t in range(n)}
            units = deal_units(n, cost, dist.world, dist.rank)
            alns = engine.align_units(A, None, units, params) if units else np.zeros(0, dtype=engine._ffi.ALIGNMENT)
            failed = [(units[i][0], units[i][1]) for i, _ in engine.failed_pairs()] if units else []
        else:
            qsum = {}
            for t, q in pairs:
                qsum[t] = qsum.get(t, 0) + QG.lengths[q]
            cost = {t: A.lengths[t] * s for t, s in qsum.items()}
            mine = shard_pairs_by_target(pairs, cost, dist.world, dist.rank)
            alns = engine.align_pairs(A, B, mine, params) if mine else np.zeros(0, dtype=engine._ffi.ALIGNMENT)
            failed = [mine[i] for i, _ in engine.failed_pairs()] if mine else []
    else:
        mine = list(pairs)
        alns = engine.align_pairs(A, B, mine, params) if mine else np.zeros(0, dtype=engine._ffi.ALIGNMENT)
        failed = [mine[i] for i, _ in engine.failed_pairs()] if mine else []
    for t, q in failed:   # the reference's script loses a failing lastz run's rows and goes on (utils.py:125-128)
        logging.warning('Alignment of %s onto %s hit an engine limit and was left out: %s', QG.names[q], A.names[t], engine.last_error())
    alns = dist.allgather_records(alns)
    blocks, kept = formats.tab_blocks(alns, A.names, QG.names, min_len, min_idt)
    return blocks, int(alns.size), kept


def write_tab(path, pairs, blocks, select=None):
    """Header + one sorted block per pair, appended in pair order (wrappers.py:996, :1056)."""
    with open(path, 'w') as f:
        f.write(formats.TAB_HEADER + '\n')
        for pr in pairs:
            if select is not None and not select(pr):
                continue
            for line in blocks.get(pr, ()):  # a pair without surviving rows adds nothing
                f.write(line + '\n')


def collapse_to_gff(tab_path, names, lengths, min_cov, min_len, source, label, prefix, kept=None):
    """wrappers.py:1116-1177: TAB -> BED -> depth >= minCov -> merge -> minLen -> GFF rows.  `kept`: the (tid, start1, end1)
    of the rows this process has just written to `tab_path` (tid = index into `names`) — the same numbers the BED projection
    would read back from the file, without parsing 6e5 lines again; None: read the file (--recycle, imported TABs)."""
    names_sorted = sorted(names, key=lambda s: s.encode())
    cid = {n: i for i, n in enumerate(names_sorted)}
    length_of = dict(zip(names, lengths))
    if kept is not None:
        rank = np.array([cid[n] for n in names], dtype=np.uint32)
        iv = np.stack([rank[kept[:, 0]], kept[:, 1].astype(np.uint32), kept[:, 2].astype(np.uint32)], axis=1) if kept.shape[0] else np.zeros((0, 3), np.uint32)
    else:
        iv = formats.bed_intervals(formats.parse_tab(tab_path), cid)
    if iv.shape[0] == 0:
        return []
    regions = engine.coverage_collapse(iv, [length_of[n] for n in names_sorted], min_cov, min_len)
    return formats.gff_repeat_lines(regions, names_sorted, source, label, prefix)


def self_repeats(A, pairs, outtab, outgff, minIdt=60, minLen=100, hspthresh=3000, minCov=3, intraCov=5,
                 splitSelf=False, reuseTab=False, label='Self_repeats', prefix=None, dist=None, source='mimeo-self',
                 B=None):
    """`mimeo self` (and, with B and source='mimeo', `mimeo x`)."""
    dist = dist or Dist()
    outtab_intra = outtab + '_intra.tab'
    kept = None
    if not reuseTab or not os.path.isfile(outtab):
        params = engine.default_params(hspthresh=hspthresh)
        blocks, _, kept = align_blocks(A, B, pairs, params, minLen, minIdt, dist)
        if len(set(pairs)) != len(pairs):
            kept = None   # a pair listed twice is written twice (the reference would run it twice): read the file back instead
        if dist.rank == 0:
            if splitSelf:
                write_tab(outtab, pairs, blocks, select=lambda pr: pr[0] != pr[1])
                write_tab(outtab_intra, pairs, blocks, select=lambda pr: pr[0] == pr[1])
            else:
                write_tab(outtab, pairs, blocks)
    if dist.rank != 0:
        return None
    k_main = k_intra = None
    if kept is not None:   # the rows just written, split like the files
        intra = kept[:, 0] == kept[:, 1] if B is None else np.zeros(kept.shape[0], bool)
        k_main = kept[~intra][:, [0, 2, 3]] if splitSelf else kept[:, [0, 2, 3]]
        k_intra = kept[intra][:, [0, 2, 3]]
    lines = collapse_to_gff(outtab, A.names, A.lengths, minCov, minLen, source, str(label), str(prefix), kept=k_main)
    if splitSelf:
        if reuseTab and not os.path.isfile(outtab_intra) and os.path.isfile(outtab):
            logging.warning("Warning: Could not find intra-chrom results file: %s \nRe-run in '--strictSelf' "
                            "mode if required." % outtab_intra)
        else:
            # the reference restarts the ID counter with the same prefix (wrappers.py:1259-1264)
            lines += collapse_to_gff(outtab_intra, A.names, A.lengths, intraCov, minLen, source,
                                     str(label) + '_intra', str(prefix), kept=k_intra)
    with open(outgff, 'w') as f:
        f.write(formats.GFF_HEADER + '\n')
        for line in lines:
            f.write(line + '\n')
    return lines


def map_hits(A, B, pairs, outtab, minIdt=95, minLen=100, hspthresh=3000, reuseTab=False, dist=None):
    """`mimeo map` alignment stage (wrappers.py:525-680): TAB only, no coverage collapse."""
    dist = dist or Dist()
    if not reuseTab or not os.path.isfile(outtab):
        params = engine.default_params(hspthresh=hspthresh)
        blocks, _, _ = align_blocks(A, B, pairs, params, minLen, minIdt, dist)
        if dist.rank == 0:
            write_tab(outtab, pairs, blocks)


def trf_filter(rows, A, prefix=None, tmatch=2, tmismatch=7, tminscore=50, tmaxperiod=50, maxtandem=40, tdelta=7):
    """wrappers.py:120-262 trfFilter with the on-GPU tandem scorer (K8) in the place of TRF: the
    slice is seq[int(tStart):int(tEnd)] exactly as the reference cuts it (origin-one start used as
    a 0-based index, wrappers.py:190), a hit stays if masked/len*100 < maxtandem (:237-240), then
    the survivors are re-sorted and renumbered (:243-259).  tmatch / tmismatch / tdelta / tminscore /
    tmaxperiod are TRF's weights and thresholds; its detection statistics tPM / tPI have no counterpart."""
    cid = {n: i for i, n in enumerate(A.names)}
    iv = np.array([(cid[r[0]], int(r[2]), int(r[3])) for r in rows], dtype=np.uint32).reshape(-1, 3)
    masked = engine.tandem_masked(A, iv, tmatch, tmismatch, tminscore, tmaxperiod, tdelta)
    keep = []
    for r, m, (c, s, e) in zip(rows, masked.tolist(), iv.tolist()):
        ln = min(e, A.lengths[c]) - s
        if ln > 0 and m / ln * 100 < float(maxtandem):
            keep.append(r)
    return formats.renumber(keep, prefix)

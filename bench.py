#!/usr/bin/env python3
"""bench.py — mimeo-self hot path on synthetic genomes (BASELINE.json metric: Gbp-aligned/s).

A "step" = one pass of the hot path over the whole workload: seed index build, seed scan,
gap-free extension, chain, gapped extension for every ordered scaffold pair x 2 strands, then
the A11 filter and the coverage-depth collapse — i.e. everything run_jobs.sh does in the
reference (src/mimeo/wrappers.py:1015-1177), with the packed genome already resident in HBM.

Workload at every N: BASELINE.json configs[1] = C2, `mimeo self` on a 50 Mbp synthetic genome
(10 scaffolds x 5 Mbp, 5 % planted repeats, seed 50), --minIdt 80 --minLen 100 --minCov 3.
N > 1 shards the ordered pairs over ranks (contiguous, cost-balanced slices of a block order), gathers the
alignment records with one all-gatherv over RCCL, and rank 0 filters + collapses: total work is
fixed, so "scaling" is "strong".

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|small|c4|c3|c5|c3small|c5small] [--no-cpu-baseline]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (mode, seed A, seed B, bp per genome, scaffolds per genome, minIdt, minCov)
    'c2': ('self', 50, None, 50_000_000, 10, 80, 3),
    'small': ('self', 50, None, 4_000_000, 4, 80, 3),
    'c4': ('self', 1000, None, 1_000_000_000, 100, 80, 3),
    # the other BASELINE configs, runnable for development (not bench lines; SURVEY §8 sizes)
    'c3': ('x', 201, 202, 200_000_000, 20, 80, 5),
    'c3small': ('x', 201, 202, 8_000_000, 4, 80, 5),
    'c5': ('map', 1001, 1002, 1_000_000_000, 100, 98, 0),
    'c5small': ('map', 1001, 1002, 8_000_000, 4, 98, 0),
}
MIN_LEN = 100


def split_contiguous(pairs, cost, world, rank):
    """Contiguous, cost-balanced slices of the target-major pair list: rank r takes the pairs whose
    cumulative cost midpoint falls in [r, r+1) * total / world."""
    if world <= 1:
        return list(pairs)
    c = np.array([cost(p) for p in pairs], dtype=np.float64)
    mid = np.cumsum(c) - c / 2
    owner = np.minimum((mid * world / c.sum()).astype(np.int64), world - 1)
    return [p for p, o in zip(pairs, owner) if o == rank]


def band_order(pairs, nscaf, world):
    """Order the pairs so that a contiguous slice is a compact block of the (target, query) matrix:
    targets in bands of B ~ S*sqrt(2/world) rows, column-major inside a band.  A rank then needs about
    B target indexes and 2*S*S/(world*B) query-strand indexes instead of one target and all 2*S query
    strands (22 -> 11 index builds per rank for C2 on 8 GPUs)."""
    if world <= 1:
        return list(pairs)
    B = max(1, min(nscaf, int(round(nscaf * (2.0 / world) ** 0.5))))
    return sorted(pairs, key=lambda p: (p[0] // B, p[1], p[0]))


def cpu_baseline(names, seqs, self_pair, cross_pair):
    """The C oracle (a single-thread port of the reference's lastz-driven path, oracle/) timed on a
    bounded sample of this workload — one (A,A) pair and one (A,B) pair — and extrapolated to the
    S self pairs and S^2 - S cross pairs of the whole job.  The reference runs its script serially
    (utils.py:247), so the 1-core figure is the like-for-like one; `allcores` (SURVEY §8d ii) runs one
    sampled cross pair per host core concurrently, i.e. the rate a pair-parallel CPU job would reach."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.lib()
    times = []
    for t, q in (self_pair, cross_pair):
        t0 = time.time()
        O.align_pair(seqs[t].tobytes(), seqs[q].tobytes())
        times.append(time.time() - t0)
    S = len(names)
    total_s = S * times[0] + (S * S - S) * times[1]
    total_bp = sum(len(s) for s in seqs)
    out = {'value': total_bp / 1e9 / total_s, 'unit': 'Gbp-aligned/s', 'cores': 1, 'kind': 'port',
           'sample': 'pairs %d-%d (%.1f s) and %d-%d (%.1f s) of %d ordered pairs; whole job extrapolated as '
                     'S*t_self + (S*S-S)*t_cross = %.0f s' % (self_pair + (times[0],) + cross_pair + (times[1], S * S, total_s))}
    # at most 16 threads: the CPU share of a one-GPU box, and it keeps this leg near half a minute
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1))
    if cores > 1 and S > 1:
        sample = [(t, q) for t in range(S) for q in range(S) if t != q][:cores]
        bufs = [seqs[i].tobytes() for i in range(S)]
        t0 = time.time()
        with ThreadPoolExecutor(len(sample)) as ex:  # ctypes drops the GIL inside the C call
            list(ex.map(lambda pr: O.align_pair(bufs[pr[0]], bufs[pr[1]]), sample))
        wall = time.time() - t0
        per_pair = wall / len(sample)  # effective seconds per cross pair with all cores busy
        mt_s = (S * S - S) * per_pair + S * times[0] / min(cores, S)
        out['allcores'] = {'value': total_bp / 1e9 / mt_s, 'unit': 'Gbp-aligned/s', 'cores': len(sample), 'kind': 'port',
                           'sample': '%d cross pairs run concurrently on %d threads in %.1f s; whole job extrapolated to %.0f s'
                                     % (len(sample), len(sample), wall, mt_s)}
    return out


def pmc_traffic(workload):
    """HBM bytes per seed-scan launch from the committed rocprofv3 PMC passes (FETCH_SIZE doubled per the
    gfx950 correction, calibrated on k3_join_count; profiles/r01_pmc_seed_scan.json).  Only C2 was measured."""
    if workload != 'c2':
        return None
    try:
        with open(os.path.join(ROOT, 'profiles', 'r01_pmc_seed_scan.json')) as f:
            return json.load(f)['k3_join_fill']['corrected_bytes_per_launch_cross_unit']
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='c2', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--emulate', default=None, help='R/N: time the shard rank R of N would get, on this one GPU, without communication (development aid)')
    args = ap.parse_args()

    from mimeo_amd import _ffi, engine, formats, workflow
    from mimeo_amd.dist import Dist
    from mimeo_amd.synth import make_families, synth_genome

    dist = Dist().init()
    if dist.world != max(1, args.gpus) and dist.rank == 0:
        print('warning: --gpus %d but WORLD_SIZE=%d' % (args.gpus, dist.world), file=sys.stderr)
    # MIMEO_FORCE_DEVICE lets several ranks share one GPU (rehearsing the N>1 path on a 1-GPU box)
    engine.init(int(os.environ.get('MIMEO_FORCE_DEVICE', dist.local_rank)))
    mode, seed, seed_b, total_bp, nscaf, MIN_IDT, MIN_COV = WORKLOADS[args.workload]
    B = None
    if mode == 'self':
        names, seqs = synth_genome(seed, total_bp, nscaf)
    else:  # two genomes carrying copies of the same repeat families
        fams = make_families(seed, 40)
        msat = 0.01 if mode == 'map' else 0.0
        names, seqs = synth_genome(seed, total_bp, nscaf, shared_families=fams, prefix='a', microsat_frac=msat)
        bnames, bseqs = synth_genome(seed_b, total_bp, nscaf, shared_families=fams, prefix='b', microsat_frac=msat)
        B = engine.Genome(bnames, bseqs)
    A = engine.Genome(names, seqs)  # packed, device resident: outside the timed region
    pairs = workflow.all_pairs(nscaf, nscaf if B is not None else None)
    L = A.lengths
    LQ = B.lengths if B is not None else L
    ew, er = (int(args.emulate.split('/')[1]), int(args.emulate.split('/')[0])) if args.emulate else (dist.world, dist.rank)
    mine = split_contiguous(band_order(pairs, nscaf, ew), lambda p: L[p[0]] * LQ[p[1]] * (1.3 if B is None and p[0] == p[1] else 1.0), ew, er)
    params = engine.default_params()
    names_sorted = sorted(names, key=lambda s: s.encode())
    cid = {n: i for i, n in enumerate(names_sorted)}
    lens_sorted = [L[names.index(n)] for n in names_sorted]

    def step():
        alns = engine.align_pairs(A, B, mine, params) if mine else np.zeros(0, dtype=_ffi.ALIGNMENT)
        st = engine.stats()
        alns = dist.allgather_records(alns)
        regions = None
        if dist.rank == 0:
            # A11 filter (length1 >= minLen, printed identity >= minIdt) and BED projection
            ln = alns['tend'].astype(np.int64) - alns['tstart']
            keep = ln >= MIN_LEN
            pct = np.array([float(formats.identity_pct(int(n), int(d))) for n, d in zip(alns['id_n'][keep], alns['id_d'][keep])])
            a = alns[keep][pct >= MIN_IDT]
            iv = np.stack([np.array([cid[names[t]] for t in a['tid']], dtype=np.uint32).reshape(-1),
                           a['tstart'] + 1, a['tend']], 1).astype(np.uint32) if a.size else np.zeros((0, 3), np.uint32)
            if mode == 'map':  # A16 has no collapse; the tandem scorer (K8) filters the target slices instead
                m = engine.tandem_masked(A, np.stack([a['tid'], a['tstart'] + 1, a['tend']], 1).astype(np.uint32)) if a.size else np.zeros(0, np.uint32)
                regions = a[m.astype(np.float64) * 100 < 40.0 * (a['tend'] - a['tstart'] - 1).clip(1)] if a.size else a
            else:
                regions = engine.coverage_collapse(iv, lens_sorted, MIN_COV, MIN_LEN)
        return st, alns, regions

    import torch
    def sync():
        dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.time()
    for _ in range(args.steps):
        st, alns, regions = step()
    sync()
    dt = dist.max_float(time.time() - t0)
    ms_per_step = 1000.0 * dt / max(1, args.steps)

    if dist.rank == 0:
        launches = max(1, st['scan_launches'])
        t_fill = st['ms_scan_fill'] / 1e3 / launches          # s per seed-scan (fill) launch, HIP events
        b_alg = st['scan_bytes_algorithmic'] / launches        # SURVEY §8(d) B_scan per launch
        achieved = b_alg / t_fill / 1e9 if t_fill > 0 else 0.0
        line = {
            'metric': 'Gbp-aligned/sec (mimeo %s, --minIdt %d --minLen %d%s)' % (mode, MIN_IDT, MIN_LEN, ' --minCov %d' % MIN_COV if MIN_COV else ' --maxtandem 40'),
            'value': total_bp / 1e9 / (ms_per_step / 1e3), 'unit': 'Gbp-aligned/s',
            'n_gpus': dist.world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'int32', 'data': 'synthetic',
            'config': {'workload': '%s: mimeo-%s, %d Mbp synthetic genome%s, %d scaffolds x %.1f Mbp, 5%% planted repeats, seed %s'
                                   % (args.workload.upper(), mode, total_bp // 1_000_000, '' if B is None else ' x2 (A, B)', nscaf,
                                      total_bp / nscaf / 1e6, seed if B is None else '%d/%d' % (seed, seed_b)),
                       'pairs': len(pairs), 'pair_strands_rank0': int(st['pair_strands']), 'parallelism': 'pairs-sharded x%d' % dist.world},
            'roofline': {'kernel': 'k3_join_fill (seed scan)', 'bound': 'hbm', 'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s',
                         'frac': achieved / 8000.0, 'traffic': pmc_traffic(args.workload), 'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC, profiles/r01_pmc_seed_scan.json)',
                         'kernel_bytes_per_launch': st['scan_bytes_kernel'] / launches,
                         'algorithmic_bytes_per_launch': b_alg, 'avg_launch_ms': t_fill * 1e3},
            'stage_ms_rank0': {k: round(st[k], 3) for k in ('ms_index', 'ms_scan', 'ms_scan_fill', 'ms_extend', 'ms_chain', 'ms_gapped', 'ms_total')},
            'counts_rank0': {k: int(st[k]) for k in ('seed_hits', 'hsps', 'chained_hsps', 'alignments')},
            'result': {'alignments': int(alns.size), 'regions': int(regions.size)},
        }
        if not args.no_cpu_baseline and dist.world == 1 and B is None:
            line['cpu_baseline'] = cpu_baseline(names, seqs, (0, 0), (0, 1))
        print(json.dumps(line))
    A.close()
    if B is not None:
        B.close()


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""bench.py — mimeo-self hot path on synthetic genomes (BASELINE.json metric: Gbp-aligned/s).

A "step" = one pass of the hot path over the whole workload: seed index build, seed scan,
gap-free extension, chain, gapped extension for every ordered scaffold pair x 2 strands, then
the A11 filter and the coverage-depth collapse — i.e. everything run_jobs.sh does in the
reference (src/mimeo/wrappers.py:1015-1177), with the packed genome already resident in HBM.

Workload at every N: BASELINE.json configs[1] = C2, `mimeo self` on a 50 Mbp synthetic genome
(10 scaffolds x 5 Mbp, 5 % planted repeats, seed 50), --minIdt 80 --minLen 100 --minCov 3.
N > 1 shards the ordered pairs over ranks (contiguous, cost-balanced, target-major), gathers the
alignment records with one all-gatherv over RCCL, and rank 0 filters + collapses: total work is
fixed, so "scaling" is "strong".

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|small|c3] [--no-cpu-baseline]
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
    # name: (seed, total bp, scaffolds)
    'c2': (50, 50_000_000, 10),
    'small': (50, 4_000_000, 4),
    'c4': (1000, 1_000_000_000, 100),
}
MIN_IDT, MIN_LEN, MIN_COV = 80, 100, 3


def split_contiguous(pairs, cost, world, rank):
    """Contiguous, cost-balanced slices of the target-major pair list: rank r takes the pairs whose
    cumulative cost midpoint falls in [r, r+1) * total / world."""
    if world <= 1:
        return list(pairs)
    c = np.array([cost(p) for p in pairs], dtype=np.float64)
    mid = np.cumsum(c) - c / 2
    owner = np.minimum((mid * world / c.sum()).astype(np.int64), world - 1)
    return [p for p, o in zip(pairs, owner) if o == rank]


def cpu_baseline(names, seqs, self_pair, cross_pair):
    """The C oracle (a single-thread port of the reference's lastz-driven path, oracle/) timed on a
    bounded sample of this workload — one (A,A) pair and one (A,B) pair — and extrapolated to the
    S self pairs and S^2 - S cross pairs of the whole job."""
    from oracle import oracle as O
    times = []
    for t, q in (self_pair, cross_pair):
        t0 = time.time()
        O.align_pair(seqs[t].tobytes(), seqs[q].tobytes())
        times.append(time.time() - t0)
    S = len(names)
    total_s = S * times[0] + (S * S - S) * times[1]
    total_bp = sum(len(s) for s in seqs)
    return {'value': total_bp / 1e9 / total_s, 'unit': 'Gbp-aligned/s', 'cores': 1, 'kind': 'port',
            'sample': 'pairs %d-%d (%.1f s) and %d-%d (%.1f s) of %d ordered pairs; whole job extrapolated as '
                      'S*t_self + (S*S-S)*t_cross = %.0f s' % (self_pair + (times[0],) + cross_pair + (times[1], S * S, total_s))}


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
    from mimeo_amd.synth import synth_genome

    dist = Dist().init()
    if dist.world != max(1, args.gpus) and dist.rank == 0:
        print('warning: --gpus %d but WORLD_SIZE=%d' % (args.gpus, dist.world), file=sys.stderr)
    # MIMEO_FORCE_DEVICE lets several ranks share one GPU (rehearsing the N>1 path on a 1-GPU box)
    engine.init(int(os.environ.get('MIMEO_FORCE_DEVICE', dist.local_rank)))
    seed, total_bp, nscaf = WORKLOADS[args.workload]
    names, seqs = synth_genome(seed, total_bp, nscaf)
    A = engine.Genome(names, seqs)  # packed, device resident: outside the timed region
    pairs = workflow.all_pairs(nscaf)
    L = A.lengths
    ew, er = (int(args.emulate.split('/')[1]), int(args.emulate.split('/')[0])) if args.emulate else (dist.world, dist.rank)
    mine = split_contiguous(pairs, lambda p: L[p[0]] * L[p[1]] * (3.0 if p[0] == p[1] else 1.0), ew, er)
    params = engine.default_params()
    names_sorted = sorted(names, key=lambda s: s.encode())
    cid = {n: i for i, n in enumerate(names_sorted)}
    lens_sorted = [L[names.index(n)] for n in names_sorted]

    def step():
        alns = engine.align_pairs(A, None, mine, params) if mine else np.zeros(0, dtype=_ffi.ALIGNMENT)
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
            'metric': 'Gbp-aligned/sec (mimeo self, --minIdt 80 --minLen 100 --minCov 3)',
            'value': total_bp / 1e9 / (ms_per_step / 1e3), 'unit': 'Gbp-aligned/s',
            'n_gpus': dist.world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'int32', 'data': 'synthetic',
            'config': {'workload': '%s: mimeo-self, %d Mbp synthetic genome, %d scaffolds x %.1f Mbp, 5%% planted repeats, seed %d'
                                   % (args.workload.upper(), total_bp // 1_000_000, nscaf, total_bp / nscaf / 1e6, seed),
                       'pairs': len(pairs), 'pair_strands_rank0': int(st['pair_strands']), 'parallelism': 'pairs-sharded x%d' % dist.world},
            'roofline': {'kernel': 'k3_join_fill (seed scan)', 'bound': 'hbm', 'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s',
                         'frac': achieved / 8000.0, 'traffic': pmc_traffic(args.workload), 'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC, profiles/r01_pmc_seed_scan.json)',
                         'kernel_bytes_per_launch': st['scan_bytes_kernel'] / launches,
                         'algorithmic_bytes_per_launch': b_alg, 'avg_launch_ms': t_fill * 1e3},
            'stage_ms_rank0': {k: round(st[k], 3) for k in ('ms_index', 'ms_scan', 'ms_scan_fill', 'ms_extend', 'ms_chain', 'ms_gapped', 'ms_total')},
            'counts_rank0': {k: int(st[k]) for k in ('seed_hits', 'hsps', 'chained_hsps', 'alignments')},
            'result': {'alignments': int(alns.size), 'regions': int(regions.size)},
        }
        if not args.no_cpu_baseline and dist.world == 1:
            line['cpu_baseline'] = cpu_baseline(names, seqs, (0, 0), (0, 1))
        print(json.dumps(line))
    A.close()


if __name__ == '__main__':
    main()

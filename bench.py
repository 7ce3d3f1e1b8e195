#!/usr/bin/env python3
"""bench.py — mimeo-self hot path on synthetic genomes (BASELINE.json metric: Gbp-aligned/s).

Default workload = the configuration BASELINE.json's metric is quoted on: C4, `mimeo self` on a 1 Gbp
synthetic genome (100 scaffolds x 10 Mbp, 5 % planted repeats, seed 1000), --minIdt 80 --minLen 100
--minCov 3.  It fits one GPU (packed genome 1.5 GB + 117 GB of seed indexes with their seed frames, in 288 GB).

A "step" (row mode, workloads c4 / c4small) = one TARGET ROW per rank through the whole hot path = 1/S of
the job's 2 S^2 (target, query, strand) units (mimeo_amd/dist.py units_of_row): the seed indexes of that
scaffold are (re)built (both strands); it is aligned as target against all S query scaffolds on the minus
strand and against itself on the plus strand; and the plus-strand units of the (S-1)/2 unordered scaffold
pairs dealt to this row are aligned ONCE and emitted for both orders — (t, q, +) and (q, t, +) have
transposed HSP sets, so the engine shares their seed scan and gap-free stage (mimeo_align_units; chain and
gapped extension run per ordered pair).  S rows name every unit of the job exactly once.  Then the A11
filter runs and the kept records of all ranks are concatenated by an all-gatherv (RCCL); the coverage-depth
collapse (per target, src/mimeo/wrappers.py:1131-1150) runs ONCE over the records of all timed steps,
inside the timed region, after the last step — a target's plus-strand records now come from several rows.
Rank r takes rows r, r+N, r+2N, ...: S/N steps on N GPUs are exactly the job run_jobs.sh does in the
reference (wrappers.py:1015-1177), every seed index built once.  The packed genome and the seed indexes of
the other scaffolds are resident in HBM when the timed region starts (the H2D copy of the packed genome,
1 GB = ~16 ms + 6 ms of K1 per job, is outside: stated in the line).  Per-GPU work is fixed as N grows:
"scaling" is "weak"; value = target bases completed by all ranks per second = (1 Gbp genome) / (time the
whole job takes at that rate).

Job mode (workloads c2, small, c3, c5, ...: the other BASELINE configs, development aids): a step is the
whole job, pairs sharded over ranks, "scaling" "strong" (the round-1 bench line for C2 is kept in profiles/).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c4small|c4job|c2|small|c3|c5|c3small|c5small|c5mid] [--no-cpu-baseline]
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
    # name: (mode, seed A, seed B, bp per genome, scaffolds per genome, minIdt, minCov, step kind)
    'c4': ('self', 1000, None, 1_000_000_000, 100, 80, 3, 'row'),
    'c4small': ('self', 1000, None, 6_000_000, 6, 80, 3, 'row'),   # the row-mode code path at test size
    # the other BASELINE configs, runnable for development (not bench lines; SURVEY §8 sizes)
    'c4job': ('self', 1000, None, 1_000_000_000, 100, 80, 3, 'job'),   # the whole 1 Gbp job in ONE call (a minute per step)
    'c2': ('self', 50, None, 50_000_000, 10, 80, 3, 'job'),
    'small': ('self', 50, None, 4_000_000, 4, 80, 3, 'job'),
    'c3': ('x', 201, 202, 200_000_000, 20, 80, 5, 'job'),
    'c3small': ('x', 201, 202, 8_000_000, 4, 80, 5, 'job'),
    'c5': ('map', 1001, 1002, 1_000_000_000, 100, 98, 0, 'job'),
    'c5small': ('map', 1001, 1002, 8_000_000, 4, 98, 0, 'job'),
    'c5mid': ('map', 1001, 1002, 40_000_000, 4, 98, 0, 'job'),   # 16 pairs of C5's own 10 Mbp scaffolds: a profile-sized piece of C5
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


def row_of(step_no, world, rank, nscaf):
    """Target scaffold of `rank` in global step `step_no` (row mode): rows are dealt round-robin, so the
    N ranks of one step hold N different targets and S/N steps cover the job."""
    return (step_no * world + rank) % nscaf


def a11_filter(alns, min_len, min_idt):
    """The awk filters of wrappers.py:1043-1052 on engine records: length1 = end1 - start1 + 1 >= minLen
    and the PRINTED one-decimal identity >= minIdt (formats.printed_tenths: the digits '%.1f' prints, element-wise)."""
    if not alns.size:
        return alns
    from mimeo_amd import formats
    a = alns[alns['tend'].astype(np.int64) - alns['tstart'] >= min_len]
    return a[formats.printed_tenths(a['id_n'], a['id_d']) / 10.0 >= min_idt]


def cpu_oracle_times(pair_jobs, threads=1):
    """Wall seconds of the C oracle (oracle/: a single-thread port of the lastz-driven path) on
    `pair_jobs` = [(target bytes, query bytes)], `threads` of them at a time (ctypes drops the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.lib()
    t0 = time.time()
    if threads <= 1:
        for t, q in pair_jobs:
            O.align_pair(t, q)
    else:
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(lambda pr: O.align_pair(pr[0], pr[1]), pair_jobs))
    return time.time() - t0


def host_cores():
    """every host core this process may run on (all host cores, count recorded)"""
    return len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)


def cpu_baseline_job(names, seqs, self_pair, cross_pair):
    """Job mode: the oracle timed on one (A,A) pair and one (A,B) pair and extrapolated to the S self
    pairs and S^2 - S cross pairs of the whole job.  The reference runs its script serially (utils.py:247),
    so the 1-core figure is the like-for-like one; `allcores` (SURVEY §8d ii) runs one sampled cross pair
    per host core concurrently, i.e. the rate a pair-parallel CPU job would reach."""
    times = [cpu_oracle_times([(seqs[t].tobytes(), seqs[q].tobytes())]) for t, q in (self_pair, cross_pair)]
    S = len(names)
    total_s = S * times[0] + (S * S - S) * times[1]
    total_bp = sum(len(s) for s in seqs)
    out = {'value': total_bp / 1e9 / total_s, 'unit': 'Gbp-aligned/s', 'cores': 1, 'kind': 'port',
           'sample': 'pairs %d-%d (%.1f s) and %d-%d (%.1f s) of %d ordered pairs; whole job extrapolated as '
                     'S*t_self + (S*S-S)*t_cross = %.0f s' % (self_pair + (times[0],) + cross_pair + (times[1], S * S, total_s))}
    cores = min(host_cores(), 16)   # job mode: whole pairs are minutes of oracle time each; sixteen of them at once
    if cores > 1 and S > 1:
        sample = [(t, q) for t in range(S) for q in range(S) if t != q][:cores]
        bufs = [seqs[i].tobytes() for i in range(S)]
        wall = cpu_oracle_times([(bufs[t], bufs[q]) for t, q in sample], threads=len(sample))
        per_pair = wall / len(sample)  # effective seconds per cross pair with all cores busy
        mt_s = (S * S - S) * per_pair + S * times[0] / min(cores, S)
        out['cores%d' % len(sample)] = {'value': total_bp / 1e9 / mt_s, 'unit': 'Gbp-aligned/s', 'cores': len(sample), 'kind': 'port',
                           'sample': '%d cross pairs run concurrently on %d threads in %.1f s; whole job extrapolated to %.0f s'
                                     % (len(sample), len(sample), wall, mt_s)}
    return out


def host_description():
    """core count and CPU model of the box the baseline ran on (recorded next to every number)"""
    model = ''
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return {'cpu_model': model, 'cpus_online': os.cpu_count(), 'cpus_allowed': len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else None}


def cpu_baseline_rows(seqs, samples=32, slice_bp=400_000, seed=8):
    """Row mode (C4), SURVEY §8d: seeded-random (target scaffold, query scaffold) pairs — at least `samples`, and at least one
    per host core —, the oracle built -O3 -march=native on this box, run on ALL host cores this process may use at once (one
    job per thread, at most 64 at once; the key says how many: `allcores` or `cores64`) and, for the like-for-like figure of the reference's serial script, the
    per-job cost of one core over FOUR of the samples.  A whole 10 Mbp x 10 Mbp pair takes the oracle about a minute, so a
    sample aligns the whole target against a random `slice_bp` window of the query, both strands; seed hits — and with them
    the time — grow with Lt x Lq, so a pair costs Lq / slice_bp samples."""
    from oracle import oracle as O
    native = O.use_native_build()
    rng = np.random.Generator(np.random.PCG64(seed))
    S, L = len(seqs), len(seqs[0])
    sl = min(slice_bp, L)
    # every core the process may use, up to 64 at once: each job builds a 10 Mbp seed table of its own, as every lastz run does, and
    # beyond that the host's memory system, not its cores, sets the time (256 jobs on 256 threads: 167 s, a lower rate than 16 threads)
    cores = min(host_cores(), 64)
    samples = max(samples, cores)
    tbytes = {}
    jobs = []
    for _ in range(samples):
        t, q = int(rng.integers(0, S)), int(rng.integers(0, S))
        o = int(rng.integers(0, L - sl + 1))
        if t not in tbytes:
            tbytes[t] = seqs[t].tobytes()
        jobs.append((tbytes[t], seqs[q][o:o + sl].tobytes()))
    wall_mt = cpu_oracle_times(jobs, threads=cores)
    nsingle = min(4, len(jobs))
    one = cpu_oracle_times(jobs[:nsingle]) / nsingle           # one core, four of the samples
    pairs = S * S
    per_pair_1 = one * (L / sl)
    total_1 = per_pair_1 * pairs
    total_mt = (wall_mt / samples) * (L / sl) * pairs
    total_bp = S * L
    host = host_description()
    out = {'value': total_bp / 1e9 / total_1, 'unit': 'Gbp-aligned/s', 'cores': 1, 'kind': 'port',
           'sample': '%d of the %d sampled (scaffold, %.1f Mbp query window) jobs on one core: %.1f s each; a 10 Mbp x 10 Mbp pair = %.0f such '
                     'windows, the job %d ordered pairs: %.0f s extrapolated; oracle/ built %s'
                     % (nsingle, samples, sl / 1e6, one, L / sl, pairs, total_1, '-O3 -march=native on this box' if native else '-O3 (portable build)'),
           'host': host}
    out['allcores' if cores == host_cores() else 'cores%d' % cores] = {
        'value': total_bp / 1e9 / total_mt, 'unit': 'Gbp-aligned/s', 'cores': cores, 'kind': 'port',
        'sample': '%d seeded-random (target, query window) jobs on %d threads (the process may use %d cores) in %.1f s; whole job extrapolated to %.0f s'
                  % (samples, cores, host_cores(), wall_mt, total_mt)}
    return out


def reference_tools_check():
    """probe the box for the reference's external binaries (outside the timed region).  With a
    real lastz on PATH the A/B diff of scripts/crosscheck_lastz.py is run and summarised; without one the
    line says that the alignment stages stay parity-unpinned."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import crosscheck_lastz
    found = crosscheck_lastz.probe()
    if found['lastz']:
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'crosscheck_lastz.py'), '--json'], capture_output=True,
                               text=True, timeout=600)
            found['crosscheck'] = json.loads(r.stdout.strip().split('\n')[-1]) if r.returncode == 0 else {'error': r.stderr[-500:]}
        except Exception as e:  # the probe must never cost the bench line
            found['crosscheck'] = {'error': repr(e)}
    return found


def pmc_traffic(workload):
    """HBM bytes per launch of the seed-scan kernel K34 from the committed rocprofv3 PMC passes
    (profiles/r03b_pmc_seed_scan.json — the final build —, scripts/gpu_pmc_traffic.sh): FETCH_SIZE doubled (gfx950 tallies the 128-byte
    requests of wide coalesced reads as 64 bytes: MI355X_MICROARCH.md, HBM; calibrated in round 1 on a kernel of known
    traffic) + WRITE_SIZE as is."""
    key = {'c2': 'c2_unit', 'c4': 'c4_unit', 'c4job': 'c4_unit'}.get(workload)
    if key is None:
        return None
    try:
        with open(os.path.join(ROOT, 'profiles', 'r03b_pmc_seed_scan.json')) as f:
            d = json.load(f)
        return int(1024 * (2 * d[key + '_FETCH_SIZE_KB_per_launch']['k34_scan_extend'] + d[key + '_WRITE_SIZE_KB_per_launch']['k34_scan_extend']))
    except Exception:
        return None


def valu_issue(avg_launch_ms):
    """What bounds K34: VALU issue.  insts_per_launch = SQ_INSTS_VALU of a first-pass launch on a C4 unit
    (profiles/r03b_pmc_k34_sq.json: the level form of the first pass, the shipped default); cycles_per_inst_mix = its opcode mix
    from the ISA priced with the measured per-opcode issue rates (scripts/k34_isa_mix.py -> profiles/r03b_k34_isa_mix.json,
    profiles/r03_valu_rate.txt); frac_of_issue_peak = the share of the 1024 SIMDs' issue time those instructions take at this
    run's launch duration.  MIMEO_K34_FORM=lane (the form of rounds 2 and 3) is priced with the r03 files."""
    tag = 'r03' if os.environ.get('MIMEO_K34_FORM') == 'lane' else 'r03b'
    if os.environ.get('MIMEO_K34_FORM') in ('cut', 'half'):
        return {'error': 'no counters on file for MIMEO_K34_FORM=' + os.environ['MIMEO_K34_FORM']}
    try:
        with open(os.path.join(ROOT, 'profiles', tag + '_k34_isa_mix.json')) as f:
            mix = json.load(f)
        with open(os.path.join(ROOT, 'profiles', tag + '_pmc_k34_sq.json')) as f:
            pmc = json.load(f)['kernels']['k34_scan_extend (first pass)']
        insts, c = pmc['SQ_INSTS_VALU'], mix['cycles_per_inst_mix']
        return {'insts_per_launch': insts, 'cycles_per_inst_mix': c,
                'frac_of_issue_peak': insts * c / (1024 * avg_launch_ms * 1e-3 * 2.4e9) if avg_launch_ms > 0 else None,
                'full_rate_share_of_valu': mix['dynamic_c4_unit']['full_rate_share_of_valu'],
                'issue_cycles_at_2.4GHz': {k: mix['issue_cycles_at_2.4GHz'][k] for k in ('full_rate_class', 'half_rate_class')},
                'lanes_active_per_valu_inst': mix['dynamic_c4_unit'].get('lanes_active_per_valu_inst'),
                'source': 'profiles/%s_pmc_k34_sq.json (SQ_INSTS_VALU per first-pass launch, C4 unit), profiles/%s_k34_isa_mix.json (ISA opcode mix), '
                          'profiles/r03_valu_rate.txt (cycles per opcode, MI355X)' % (tag, tag)}
    except Exception as e:
        return {'error': repr(e)}


def standalone_seed_scan(engine, A, tq):
    """Calibration OUTSIDE the timed region: the stand-alone seed scan of round 1 (K3 index join, materialises the hit
    array; kept behind mimeo_seed_hits) on one unit of the workload, alone on the device — the seed scan's own
    roofline fraction by the survey's byte model, next to the in-pipeline figure of the fused kernel."""
    t, q = tq
    hits = engine.seed_hits(A, t, A, q, 0)
    st = engine.stats()
    Lq = int(A.lengths[q])
    b_alg = (Lq + 3) // 4 + 8 * 13 * max(0, Lq - 18) + 12 * int(hits.size)
    ms = st['ms_scan_fill']
    return {'kernel': 'k3_join_fill (stand-alone seed scan: hits written to HBM)', 'avg_launch_ms': ms,
            'algorithmic_bytes_per_launch': b_alg, 'achieved': b_alg / (ms / 1e3) / 1e9 if ms > 0 else None,
            'frac': b_alg / (ms / 1e3) / 8e12 if ms > 0 else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='c4', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--emulate', default=None, help='job mode, R/N: time the shard rank R of N would get, on this one GPU, without communication (development aid)')
    args = ap.parse_args()

    from mimeo_amd import _ffi, engine, workflow
    from mimeo_amd.dist import Dist
    from mimeo_amd.synth import make_families, synth_genome

    dist = Dist().init()
    if dist.world != max(1, args.gpus) and dist.rank == 0:
        print('warning: --gpus %d but WORLD_SIZE=%d' % (args.gpus, dist.world), file=sys.stderr)
    # MIMEO_FORCE_DEVICE lets several ranks share one GPU (rehearsing the N>1 path on a 1-GPU box)
    engine.init(int(os.environ.get('MIMEO_FORCE_DEVICE', dist.local_rank)))
    mode, seed, seed_b, total_bp, nscaf, MIN_IDT, MIN_COV, kind = WORKLOADS[args.workload]
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
    params = engine.default_params()
    names_sorted = sorted(names, key=lambda s: s.encode())
    cid_of_tid = np.array([names_sorted.index(n) for n in names], dtype=np.uint32)  # chrom id = rank in C-locale name order
    lens_sorted = [L[names.index(n)] for n in names_sorted]
    step_no = [0]

    if kind == 'row':
        A.build_indexes()  # resident like the packed genome; every step rebuilds its own target's two
        mine = None
    else:
        ew, er = (int(args.emulate.split('/')[1]), int(args.emulate.split('/')[0])) if args.emulate else (dist.world, dist.rank)
        mine = split_contiguous(band_order(pairs, nscaf, ew), lambda p: L[p[0]] * LQ[p[1]] * (1.3 if B is None and p[0] == p[1] else 1.0), ew, er)

    def intervals_of(a):
        iv = np.zeros(a.size, dtype=_ffi.INTERVAL)
        if a.size:
            iv['chrom'], iv['start'], iv['end'] = cid_of_tid[a['tid']], a['tstart'] + 1, a['tend']
        return iv

    kept_rows = []

    def step_row():
        """One target row per rank: indexes of that scaffold, its 2*S units (plus-strand pairs shared), A11 filter, gather."""
        from mimeo_amd.dist import units_of_row
        t = row_of(step_no[0], dist.world, dist.rank, nscaf)
        step_no[0] += 1
        A.drop_indexes([t])
        alns = engine.align_units(A, None, units_of_row(t, nscaf), params)
        st = engine.stats()
        a = dist.allgather_records(a11_filter(alns, MIN_LEN, MIN_IDT))
        kept_rows.append(a)
        return st, a, None

    def step_job():
        alns = engine.align_pairs(A, B, mine, params) if mine else np.zeros(0, dtype=_ffi.ALIGNMENT)
        st = engine.stats()
        alns = dist.allgather_records(alns)
        regions = None
        if dist.rank == 0:
            a = a11_filter(alns, MIN_LEN, MIN_IDT)
            if mode == 'map':  # A16 has no collapse; the tandem scorer (K8) filters the target slices instead
                m = engine.tandem_masked(A, np.stack([a['tid'], a['tstart'] + 1, a['tend']], 1).astype(np.uint32)) if a.size else np.zeros(0, np.uint32)
                regions = a[m.astype(np.float64) * 100 < 40.0 * (a['tend'] - a['tstart'] - 1).clip(1)] if a.size else a
            else:
                regions = engine.coverage_collapse(intervals_of(a), lens_sorted, MIN_COV, MIN_LEN)
        return st, alns, regions

    step = step_row if kind == 'row' else step_job

    import torch
    def sync():
        dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    del kept_rows[:]
    sync()
    t0 = time.time()
    agg = {}
    n_aln = n_reg = 0
    for _ in range(args.steps):
        st, alns, regions = step()
        for k, v in st.items():
            agg[k] = agg.get(k, 0) + v
        n_aln += int(alns.size)
        n_reg += int(regions.size) if regions is not None else 0
    if kind == 'row' and dist.rank == 0:
        # the coverage collapse of everything the timed rows produced (a whole job: of every target), once, inside the timed region
        a = np.concatenate(kept_rows) if kept_rows else np.zeros(0, dtype=_ffi.ALIGNMENT)
        regions = engine.coverage_collapse(intervals_of(a), lens_sorted, MIN_COV, MIN_LEN)
        n_reg = int(regions.size)
    sync()
    dt = dist.max_float(time.time() - t0)
    ms_per_step = 1000.0 * dt / max(1, args.steps)

    standalone = None
    if dist.rank == 0 and B is None and nscaf > 1:
        try:
            standalone = standalone_seed_scan(engine, A, (0, 1))
        except Exception as e:   # the calibration must never cost the bench line
            standalone = {'error': repr(e)}
    if dist.rank == 0:
        # the seed-scan kernel K34 takes a whole batch of units per launch: duration by HIP events around each launch (first pass +
        # split pass), algorithmic bytes = SURVEY §8(d) B_scan summed over the units of the launch
        launches = max(1, agg.get('scan_kernel_launches') or agg['scan_launches'])
        units = max(1, agg['scan_launches'])
        t_fill = agg['ms_scan_fill'] / 1e3 / launches          # s per seed-scan launch, HIP events
        b_alg = agg['scan_bytes_algorithmic'] / launches        # B_scan per launch
        achieved = b_alg / t_fill / 1e9 if t_fill > 0 else 0.0
        traffic = pmc_traffic(args.workload)
        scaf_mbp = total_bp / nscaf / 1e6
        if kind == 'row':
            # target bases completed by all ranks per second; S/N such steps are the whole job
            value = dist.world * (total_bp / nscaf) / 1e9 / (ms_per_step / 1e3)
            what = ('a step = one target row per rank = %d units = 1/%d of the job (%d ordered pairs x 2 strands): the row\'s %d minus-strand units, '
                    'its own plus-strand unit and the plus-strand units of the unordered pairs dealt to it, each computed once and emitted for both '
                    'orders (shared seed scan and gap-free stage); its seed indexes rebuilt, A11 filter, records all-gathered; the coverage collapse '
                    'runs once over the records of all timed steps, inside the timed region; rows dealt round-robin; packed genome and the other '
                    'scaffolds\' indexes resident (H2D of the genome, ~22 ms per job, outside the timed region)'
                    % (2 * nscaf, nscaf, len(pairs), nscaf))
            par = 'target rows x%d (no data-path collective; all-gatherv of records)' % dist.world
        else:
            value = total_bp / 1e9 / (ms_per_step / 1e3)
            what = 'a step = the whole job (%d ordered pairs)' % len(pairs)
            par = 'pairs-sharded x%d' % dist.world
        line = {
            'metric': 'Gbp-aligned/sec (mimeo %s, %s genome, --minIdt %d --minLen %d%s)'
                      % (mode, '%g Gbp' % (total_bp / 1e9) if total_bp >= 1e9 else '%d Mbp' % (total_bp // 1_000_000), MIN_IDT, MIN_LEN,
                         ' --minCov %d' % MIN_COV if MIN_COV else ' --maxtandem 40'),
            'value': value, 'unit': 'Gbp-aligned/s',
            'n_gpus': dist.world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak' if kind == 'row' else 'strong', 'vs_baseline': None, 'dtype': 'int32', 'data': 'synthetic',
            'config': {'workload': '%s: mimeo-%s, %d Mbp synthetic genome%s, %d scaffolds x %.1f Mbp, 5%% planted repeats, seed %s; %s'
                                   % (args.workload.upper(), mode, total_bp // 1_000_000, '' if B is None else ' x2 (A, B)', nscaf,
                                      scaf_mbp, seed if B is None else '%d/%d' % (seed, seed_b), what),
                       'pairs': len(pairs), 'pair_strands_rank0': int(st['pair_strands']), 'parallelism': par},
            'roofline': {'kernel': 'k34_scan_extend (seed scan fused with the gap-free pre-filter; one launch per BATCH of (target, query, strand) units: grid.y = unit)',
                         'bound': 'valu', 'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s', 'frac': achieved / 8000.0,
                         'valu': valu_issue(t_fill * 1e3 * launches / units) if args.workload in ('c4', 'c4job') else None,
                         'traffic': (traffic * units / launches) if traffic else None, 'traffic_unit': 'HBM bytes per launch = per-unit bytes x units per launch (rocprofv3 PMC on one C4 unit, profiles/r03b_pmc_seed_scan.json; FETCH_SIZE doubled per the guide)',
                         'traffic_frac': (traffic * units / launches / t_fill / 8e12) if (traffic and t_fill > 0) else None,
                         'kernel_bytes_per_launch': agg['scan_bytes_kernel'] / launches,
                         'algorithmic_bytes_per_launch': b_alg, 'avg_launch_ms': t_fill * 1e3, 'launches_timed_rank0': int(launches),
                         'units_per_launch': units / launches, 'ms_per_unit': t_fill * 1e3 * launches / units,
                         'note': 'achieved / peak / frac = the HBM roofline SURVEY 8(d) defines: B_scan per unit / HIP-event duration of the launch / 8 TB/s.  '
                                 'The fused kernel never writes the hit array that byte model charges for; what bounds it is VALU issue of the '
                                 'pre-filter (bound = valu; the valu block: instructions per launch x cycles per instruction of its opcode mix / '
                                 'issue time of the 1024 SIMDs).  traffic_frac is what it really moves; frac_standalone is the seed scan alone (K3).',
                         'frac_standalone': standalone},
            'stage_ms_per_step_rank0': {k: round(agg[k] / max(1, args.steps), 3) for k in ('ms_index', 'ms_scan', 'ms_scan_fill', 'ms_extend', 'ms_chain', 'ms_gapped', 'ms_total')},
            'stage_note': 'ms_scan = heavy phase (K34 + the walk-queue kernel per unit), ms_scan_fill = the K34 launches alone, ms_extend = tails once per batch; K34 timed region includes nothing else: inputs (packed genome, seed indexes of the other scaffolds) are resident in HBM, the H2D of the genome (1 GB, ~16 ms + 6 ms K1 per 50 s job) is outside',
            'counts_per_step_rank0': {k: int(agg[k] // max(1, args.steps)) for k in ('seed_hits', 'walked_hits', 'followers', 'hsps', 'chained_hsps', 'alignments', 'batches', 'queue_reruns')},
            'result': ({'records_kept': n_aln, 'regions': n_reg} if kind == 'row' else {'alignments': int(alns.size), 'regions': int(regions.size)}),
        }
        if kind == 'row':
            line['whole_job_s_at_this_rate'] = total_bp / 1e9 / value
        line['reference_tools'] = reference_tools_check()
        if not args.no_cpu_baseline and dist.world == 1 and B is None:
            line['cpu_baseline'] = cpu_baseline_rows(seqs) if kind == 'row' and nscaf > 1 else cpu_baseline_job(names, seqs, (0, 0), (0, 1))
        print(json.dumps(line))
    A.close()
    if B is not None:
        B.close()


if __name__ == '__main__':
    main()

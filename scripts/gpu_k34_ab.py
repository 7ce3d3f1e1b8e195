#!/usr/bin/env python3
"""GPU box: the first pass of K34 in its four forms, checked and timed in one process.

    python scripts/gpu_k34_ab.py  ->  gpurun_out/k34_ab.json

Forms (MIMEO_K34_FORM, read per call): level = two-segment tiles cut at the middle key + level emission;
cut = tiles cut by entry count + level emission; half = middle-key cut + lane-major emission; lane = the first pass as it was (13 probes per entry and segment, prefix
sum, lane-major descriptor emission).  Checks: (a) 0.3 Mbp x 8 Mbp (two segments per tile) and three 0.2 Mbp scaffolds (one
segment; a self unit) against the C oracle — HSPs and the seed-hit count, both strands; (b) units of 10 Mbp x 10 Mbp (C4)
against MIMEO_HEAVY=v1, the round-1 decomposition (hit array + K4), byte for byte.  Timing: the K34 launches of a 16-pair
call on four 10 Mbp scaffolds (stats ms_scan_fill, HIP events), three calls per form.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mimeo_amd import engine                      # noqa: E402
from mimeo_amd.synth import make_families, synth_genome   # noqa: E402
from oracle import oracle as O                    # noqa: E402

HCOLS = ['tstart', 'qstart', 'length', 'score', 'raw_score']
FORMS = ('level', 'cut', 'half', 'lane')
out = {'forms': {'level': 'middle-key cut + level emission', 'cut': 'entry-count cut + level emission', 'half': 'middle-key cut + lane-major emission', 'lane': 'first pass as before'},
       'checks': [], 'timing': {}}
ok = True


def setenv(**kw):
    for k in ('MIMEO_K34_FORM', 'MIMEO_HEAVY'):
        os.environ.pop(k, None)
    for k, v in kw.items():
        os.environ[k] = str(v)


def note(name, passed, **kw):
    global ok
    ok = ok and bool(passed)
    out['checks'].append(dict(name=name, passed=bool(passed), **kw))
    print(('PASS ' if passed else 'FAIL ') + name, kw, flush=True)


def dump():
    out['all_passed'] = ok
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'k34_ab.json'), 'w') as f:
        json.dump(out, f, indent=1)


engine.init(0)
t00 = time.time()

# (a) against the oracle
fams = make_families(83, 12, (300, 4000))
_, a = synth_genome(84, 300_000, 1, repeat_frac=0.04, shared_families=fams)
_, b = synth_genome(85, 8_000_000, 1, repeat_frac=0.04, shared_families=fams)
g = engine.Genome(['t', 'q'], [a[0], b[0]])
T, Q = a[0].tobytes(), b[0].tobytes()
for strand in (0, 1):
    exp = np.sort(O.ungapped_hsps(T, Q, strand, O.default_params(chain=0))[HCOLS], order=HCOLS)
    nhits = O.seed_hits(T, Q, strand).size
    for form in FORMS:
        setenv(MIMEO_K34_FORM=form)
        got = np.sort(engine.ungapped_hsps(g, 0, g, 1, strand, engine.default_params(chain=0))[HCOLS], order=HCOLS)
        st = engine.stats()
        note('0.3Mx8M strand %d form %s vs oracle' % (strand, form),
             got.size == exp.size and (got == exp).all() and st['seed_hits'] == nhits, hsps=int(got.size), oracle_hsps=int(exp.size),
             seed_hits=int(st['seed_hits']), oracle_seed_hits=int(nhits))
g.close()
dump()

names, seqs = synth_genome(3, 600_000, 3, repeat_frac=0.1, families=4, cons_len=(300, 1500))
g = engine.Genome(names, seqs)
for t, q in ((0, 1), (2, 2), (1, 0)):
    T, Q = seqs[t].tobytes(), seqs[q].tobytes()
    for strand in (0, 1):
        exp = np.sort(O.ungapped_hsps(T, Q, strand, O.default_params(chain=0))[HCOLS], order=HCOLS)
        nhits = O.seed_hits(T, Q, strand).size
        for form in FORMS:
            setenv(MIMEO_K34_FORM=form)
            got = np.sort(engine.ungapped_hsps(g, t, g, q, strand, engine.default_params(chain=0))[HCOLS], order=HCOLS)
            st = engine.stats()
            note('0.2M (%d,%d) strand %d form %s vs oracle' % (t, q, strand, form),
                 got.size == exp.size and (got == exp).all() and st['seed_hits'] == nhits, hsps=int(got.size), seed_hits=int(st['seed_hits']))
setenv()
got = engine.align_pair(g, 0, g, 1)
exp = O.align_pair(seqs[0].tobytes(), seqs[1].tobytes())
cols = ['tstart', 'tend', 'qstart', 'qend', 'score', 'id_n', 'id_d', 'qstrand']
note('0.2M (0,1) alignments vs oracle', got.size == exp.size and (np.sort(got[cols], order=cols) == np.sort(exp[cols], order=cols)).all(), n=int(got.size))
g.close()
dump()
print('oracle checks done at %.0f s' % (time.time() - t00), flush=True)

# (b) C4 units against the round-1 decomposition, and the timing
names, seqs = synth_genome(1000, 40_000_000, 4, repeat_frac=0.05)
g = engine.Genome(names, seqs)
units = ((0, 1, 0), (0, 1, 1), (1, 1, 0), (2, 3, 1))
setenv(MIMEO_HEAVY='v1')
ref = []
for t, q, s in units:
    h = engine.ungapped_hsps(g, t, g, q, s)
    ref.append((h.tobytes(), engine.stats()['seed_hits']))
for form in FORMS:
    setenv(MIMEO_K34_FORM=form)
    ms = []
    for (t, q, s), (rb, rh) in zip(units, ref):
        h = engine.ungapped_hsps(g, t, g, q, s)
        st = engine.stats()
        ms.append(round(st['ms_scan_fill'], 4))
        note('C4 unit (%d,%d,%d) form %s vs v1' % (t, q, s, form), h.tobytes() == rb and st['seed_hits'] == rh, hsps=int(h.size), seed_hits=int(st['seed_hits']))
    out['timing'].setdefault('single_unit_ms_k34', {})[str(form)] = ms
dump()
pairs = [(t, q) for t in range(4) for q in range(4)]
md5 = {}
import hashlib
for rep in range(3):
    for form in FORMS:
        setenv(MIMEO_K34_FORM=form)
        t0 = time.time()
        al = engine.align_pairs(g, None, pairs)
        wall = time.time() - t0
        st = engine.stats()
        md5.setdefault(form, set()).add(hashlib.md5(al.tobytes()).hexdigest())
        out['timing'].setdefault('pairs16_ms', {}).setdefault(str(form), []).append(
            dict(k34=round(st['ms_scan_fill'], 3), heavy=round(st['ms_scan'], 3), tails=round(st['ms_extend'], 3), gapped=round(st['ms_gapped'], 3),
                 wall=round(1e3 * wall, 1), launches=int(st.get('scan_kernel_launches', 0)), seed_hits=int(st['seed_hits'])))
        print('form', form, out['timing']['pairs16_ms'][str(form)][-1], flush=True)
note('16 pairs of four 10 Mbp scaffolds: one md5 over the forms and repeats', len(set().union(*md5.values())) == 1, md5={str(k): sorted(v) for k, v in md5.items()})
setenv()
g.close()
dump()
print(json.dumps(out['timing'], indent=1))
print('ALL PASSED' if ok else 'SOME CHECKS FAILED', 'in %.0f s' % (time.time() - t00), flush=True)
sys.exit(0 if ok else 1)

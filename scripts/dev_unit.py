"""Run a few (target, query, strand) units of a C2-like pair for profiling: python scripts/dev_unit.py [L] [reps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimeo_amd import engine
from mimeo_amd.synth import synth_genome
L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
engine.init(0)
names, seqs = synth_genome(50, 2 * L, 2, repeat_frac=0.05)
g = engine.Genome(names, seqs)
for it in range(reps):
    t = time.time(); a = engine.align_pairs(g, None, [(0, 1), (0, 0)]); dt = time.time() - t
    s = engine.stats()
    print('alns', a.size, 'wall %.3f' % dt, {k: round(v, 2) if isinstance(v, float) else v for k, v in s.items() if k.startswith('ms_') or k in ('seed_hits', 'hsps')})

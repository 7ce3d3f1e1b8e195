"""development aid: one unit through mimeo_ungapped_hsps with statistics"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimeo_amd import engine
from mimeo_amd.synth import synth_genome
L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 600_000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 91
engine.init(0)
names, seqs = synth_genome(seed, 2 * L, 2, repeat_frac=0.05)
g = engine.Genome(names, seqs)
for rep in range(3):
    t = time.time()
    h = engine.ungapped_hsps(g, 0, g, 1, 0)
    st = engine.stats()
    print('L', L, 'hsps', h.size, 'wall %.1f ms' % (1e3 * (time.time() - t)), {k: st[k] for k in ('seed_hits', 'walked_hits', 'followers', 'queue_reruns', 'ms_index', 'ms_scan', 'ms_extend')}, flush=True)

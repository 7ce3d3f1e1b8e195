import sys, time, numpy as np
sys.path.insert(0, '.')
from mimeo_amd import engine
from mimeo_amd.synth import synth_genome
engine.init(0)
L = int(float(sys.argv[1]))
t = time.time()
names, seqs = synth_genome(7, L, 1, repeat_frac=0.02, families=10)
print('synth', round(time.time() - t, 1), flush=True)
g = engine.Genome(names, seqs)
t = time.time()
a = engine.align_pair(g, 0, g, 0)
st = engine.stats()
print('L', L, 'alns', a.size, 'wall %.2f s' % (time.time() - t), 'hits %.3g' % st['seed_hits'], 'hsps', st['hsps'], flush=True)
triv = a[(a['tstart'] == 0) & (a['tend'] == L) & (a['qstrand'] == 0)]
print('trivial', triv.size, int(triv['score'][0]) if triv.size else None)
g.close()

"""Fragmented assembly: `mimeo self` alignment stage of S scaffolds x L bp, packed into super-scaffolds (default) against
the unit-per-pair path on a sub-matrix of the pairs (the whole matrix would take minutes there).
usage: gpu_pack_bench.py [S] [L] [sub]"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mimeo_amd import engine
from mimeo_amd.synth import synth_genome
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
sub = int(sys.argv[3]) if len(sys.argv) > 3 else 40
engine.init(0)
t = time.time()
names, seqs = synth_genome(4242, S * L, S, repeat_frac=0.05, families=40)
print('genome %d x %d bp synthesised in %.1f s' % (S, L, time.time() - t), flush=True)
g = engine.Genome(names, seqs)
pairs = [(a, b) for a in range(S) for b in range(S)]
os.environ.pop('MIMEO_PACK', None)
for rep in range(2):
    t = time.time()
    a = engine.align_pairs(g, None, pairs)
    wall = time.time() - t
    st = engine.stats()
    print('packed: %d pairs, %d alignments, wall %.2f s (library %.2f s): super_units %d batches %d hsps %d; index %.0f ms heavy %.0f tails %.0f chain %.0f gapped %.0f' % (
        len(pairs), a.size, wall, st['ms_total'] / 1e3, st['super_units'], st['batches'], st['hsps'], st['ms_index'], st['ms_scan'], st['ms_extend'], st['ms_chain'], st['ms_gapped']), flush=True)
packed_s = st['ms_total'] / 1e3
# sub-matrix on the other path
os.environ['MIMEO_PACK'] = '0'
ids = list(range(0, S, max(1, S // sub)))[:sub]
sp = [(x, y) for x in ids for y in ids]
t = time.time()
b = engine.align_pairs(g, None, sp)
st = engine.stats()
unp = st['ms_total'] / 1e3
print('unit per pair: %d pairs in %.2f s -> %.0f s for all %d pairs; packed is %.0fx faster' % (len(sp), unp, unp * len(pairs) / len(sp), len(pairs), unp * len(pairs) / len(sp) / packed_s), flush=True)
idset = np.zeros(S, bool); idset[ids] = True
asub = a[idset[a['tid']] & idset[a['qid']]]
print('sub-matrix records: packed %d unpacked %d identical %s' % (asub.size, b.size, hashlib.md5(asub.tobytes()).hexdigest() == hashlib.md5(b.tobytes()).hexdigest()), flush=True)

"""One very large self pair (default 150 Mbp): whole-pair time and stage split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimeo_amd import engine
from mimeo_amd.synth import synth_genome
L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 150_000_000
engine.init(0)
t = time.time()
names, seqs = synth_genome(777, L, 1, repeat_frac=0.05)
print('synthesised in %.1f s' % (time.time() - t), flush=True)
g = engine.Genome(names, seqs)
t = time.time()
a = engine.align_pairs(g, None, [(0, 0)])
st = engine.stats()
print('L %d: %d alignments, wall %.2f s; hsps %d chained %d; ms index %.0f heavy %.0f tails %.0f chain %.0f gapped %.0f total %.0f; batches %d reruns %d' % (
    L, a.size, time.time() - t, st['hsps'], st['chained_hsps'], st['ms_index'], st['ms_scan'], st['ms_extend'], st['ms_chain'], st['ms_gapped'], st['ms_total'], st['batches'], st['queue_reruns']), flush=True)

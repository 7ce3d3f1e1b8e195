"""development aid: one unit with scaffolds of different lengths (target LT, query LQ) through mimeo_ungapped_hsps"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mimeo_amd import engine
from mimeo_amd.synth import synth_genome
LT = int(float(sys.argv[1])); LQ = int(float(sys.argv[2]))
engine.init(0)
names, seqs = synth_genome(1000, 2 * max(LT, LQ), 2, repeat_frac=0.05)
seqs = [seqs[0][:LT].copy(), seqs[1][:LQ].copy()]
g = engine.Genome(names, seqs)
for rep in range(3):
    h = engine.ungapped_hsps(g, 0, g, 1, 0)
    st = engine.stats()
    print('LT', LT, 'LQ', LQ, 'hsps', h.size, {k: st[k] for k in ('seed_hits', 'walked_hits', 'ms_scan', 'ms_scan_fill')}, flush=True)

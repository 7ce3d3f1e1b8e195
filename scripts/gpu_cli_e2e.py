"""End to end through the CLI: FASTA on disk -> `python -m mimeo_amd self` -> TAB + GFF3 (reference: src/mimeo/run_self.py:169-255),
wall time beside the engine's own time, and the host text stages on their own.

    python scripts/gpu_cli_e2e.py [c2|c4|small]   -> one JSON line (gpurun_out/r03_cli_e2e_<workload>.json by the caller)

c2: 50 Mbp / 10 scaffolds (seed 50); c4: 1 Gbp / 100 scaffolds (seed 1000; the FASTA is 1 GB on /tmp); flags --minIdt 80
--minLen 100 --minCov 3.  The CLI runs in THIS process (run_self.main), so `import torch` / library load are not counted."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mimeo_amd import engine, formats, run_self, workflow  # noqa: E402
from mimeo_amd.synth import synth_genome, write_fasta  # noqa: E402

W = {'small': (50, 4_000_000, 4), 'c2': (50, 50_000_000, 10), 'c4': (1000, 1_000_000_000, 100)}


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else 'c2'
    seed, total, nscaf = W[wl]
    names, seqs = synth_genome(seed, total, nscaf)
    td = tempfile.mkdtemp(prefix='mimeo_e2e_', dir='/tmp')
    fa = os.path.join(td, 'A.fa')
    write_fasta(fa, names, seqs)
    del seqs
    out = os.path.join(td, 'out')
    engine.init(0)
    t0 = time.time()
    run_self.main(['--afasta', fa, '-d', out, '--minIdt', '80', '--minLen', '100', '--minCov', '3', '--loglevel', 'WARNING'])
    wall = time.time() - t0
    ntab = sum(1 for _ in open(os.path.join(out, 'mimeo_alignment.tab'))) - 1
    ngff = sum(1 for l in open(os.path.join(out, 'mimeo-self_repeats.gff3')) if not l.startswith('#'))
    # the same steps once more, timed one by one
    t = time.time(); A = engine.Genome.from_fasta(fa); t_ingest = time.time() - t
    pairs = workflow.all_pairs(len(A.names))
    t = time.time(); alns = engine.align_pairs(A, None, pairs); t_align = time.time() - t
    st = engine.stats()
    t = time.time(); blocks, kept = formats.tab_blocks(alns, A.names, A.names, 100, 80); t_tab = time.time() - t
    tab2 = os.path.join(td, 'again.tab')
    t = time.time(); workflow.write_tab(tab2, pairs, blocks); t_write = time.time() - t
    t = time.time(); lines = workflow.collapse_to_gff(tab2, A.names, A.lengths, 3, 100, 'mimeo-self', 'Self_Repeat', 'Self_Repeat', kept=kept[:, [0, 2, 3]]); t_collapse = time.time() - t
    t = time.time(); rows = formats.parse_tab(tab2); iv = formats.bed_intervals(rows, {n: i for i, n in enumerate(sorted(A.names))}); t_reparse = time.time() - t
    assert open(tab2).read() == open(os.path.join(out, 'mimeo_alignment.tab')).read() and len(lines) == ngff and iv.shape[0] == ntab
    print(json.dumps({
        'workload': wl, 'genome_bp': total, 'scaffolds': nscaf, 'fasta_bytes': os.path.getsize(fa),
        'cli_wall_s': round(wall, 3), 'tab_rows': ntab, 'gff_rows': ngff, 'alignment_records': int(alns.size),
        'steps_s': {'fasta_ingest': round(t_ingest, 3), 'align_pairs_call': round(t_align, 3), 'engine_ms_total': round(st['ms_total'], 1),
                    'tab_blocks (A11 filter + sort + format)': round(t_tab, 3), 'write_tab': round(t_write, 3),
                    'collapse_to_gff (K7 + GFF rows, intervals from the kept rows)': round(t_collapse, 3),
                    'parse_tab + bed_intervals (only --recycle / imported TABs take this path now)': round(t_reparse, 3)},
        'host_text_stages_s': round(t_tab + t_write + t_collapse, 3),
        'gbp_aligned_per_s_cli': round(total / 1e9 / wall, 5),
    }))


if __name__ == '__main__':
    main()

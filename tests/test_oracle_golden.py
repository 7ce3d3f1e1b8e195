"""The pinned half of the oracle (stages A11-A16) against the vectors produced by the
reference's own pipeline text / pandas code (tests/golden/make_golden.py) and the hand-derived
coverage-collapse known answers."""
import json
import os

import pytest

from oracle import pipeline as P

G = os.path.join(os.path.dirname(__file__), 'golden')


def _load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


HEADER = '#name1\tstrand1\tstart1\tend1\tname2\tstrand2\tstart2+\tend2+\tscore\tidentity'


@pytest.mark.parametrize('k', range(4))
def test_filter_stage_matches_reference_pipeline(k):
    d = _load('filter_stage.json')
    case = d['cases'][k]
    main, intra = [HEADER], [HEADER]
    for t, q in case['pairs']:
        block = P.filter_project_sort(d['general']['%s_onto_%s' % (q, t)], case['minLen'], case['minIdt'])
        (intra if case['strictSelf'] and t == q else main).extend(block)
    assert '\n'.join(main) + '\n' == case['outtab']
    if case['strictSelf']:
        assert '\n'.join(intra) + '\n' == case['outtab_intra']
    if 'sorted_bed' in case:
        assert '\n'.join(P.bed_project_sort(main)) + '\n' == case['sorted_bed']


def test_gff_formatter_matches_reference_awk():
    for case in _load('gff_stage.json'):
        regions = []
        for line in case['merged_bed'].strip().split('\n'):
            c, s, e = line.split('\t')
            if int(e) - int(s) >= case['minLen']:
                regions.append((c, int(s), int(e)))
        src = 'mimeo-self' if case['mode'] == 'self' else 'mimeo'
        got = '\n'.join(P.gff_self_lines(regions, case['label'], case['prefix'], source=src)) + '\n'
        assert got == case['gff']


def test_import_align_and_map_gff_match_reference_pandas():
    d = _load('map_import.json')
    rows = P.import_align(d['tab'], d['prefix'], d['minLen'], d['minIdt'])
    assert [r[10] for r in rows] == d['uids']
    got = '\n'.join(P.gff_map_lines(rows, d['chromlens'], d['label'])) + '\n'
    assert got == d['gff']


def test_collapse_known_answers():
    for case in _load('collapse_kat.json')['cases']:
        got = P.coverage_collapse([tuple(x) for x in case['intervals']], case['chromlens'],
                                  case['min_cov'], case['min_len'])
        assert [list(x) for x in got] == case['expect'], case['name']


def test_chain_stage_of_the_c_oracle_against_the_rule_spelled_out():
    """`orc_chain_hsps` (oracle/mimeo_oracle.c: chain_hsps — what tests/test_gpu_chain.py holds the three GPU chain kernels to)
    against DESIGN.md §2 rule 5 written out in plain Python: HSPs in (tstart, qstart, length) order, best[j] = score[j] +
    max(0, best[i]) over the i that end at or before j's start in both sequences, ties to the earliest predecessor and the
    earliest end.  PARITY UNPINNED like every LASTZ-shaped stage: the reference holds no chain fixture."""
    import numpy as np
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    for n, span, equal in ((1, 100, False), (2, 100, False), (60, 3000, False), (400, 20000, False), (300, 1500, True)):
        h = np.zeros(n, dtype=O.HSP)
        h['tstart'], h['qstart'] = rng.integers(0, span, n), rng.integers(0, span, n)
        h['length'] = 40 if equal else rng.integers(20, 300, n)
        h['score'] = 3000 if equal else h['length'].astype(np.int64) * 90 - rng.integers(0, 500, n)
        key = np.stack([h['tstart'], h['qstart'], h['length']], 1)
        h = h[np.unique(key, axis=0, return_index=True)[1]]
        got = O.chain_hsps(h[rng.permutation(h.size)])
        s = h[np.lexsort((h['length'], h['qstart'], h['tstart']))]
        best, pred = [0] * s.size, [-1] * s.size
        for j in range(s.size):
            b, p = 0, -1
            for i in range(j):
                if s['tstart'][i] + s['length'][i] <= s['tstart'][j] and s['qstart'][i] + s['length'][i] <= s['qstart'][j] and best[i] > b:
                    b, p = best[i], i
            best[j], pred[j] = b + int(s['score'][j]), p
        flags = np.zeros(s.size, dtype=np.uint32)
        k = int(np.argmax(best))   # first maximum
        while k >= 0:
            flags[k] = 1
            k = pred[k]
        assert np.array_equal(got['tstart'], s['tstart']) and np.array_equal(got['length'], s['length'])
        assert np.array_equal(got['flags'] & 1, flags), (n, span, equal)


def test_collapse_is_the_four_commands_one_after_the_other():
    """The collapse oracle (per-base depth, maximal runs >= minCov, minLen) against the reference's four commands taken literally
    (src/mimeo/wrappers.py:1131-1177): `bedtools genomecov -bg` (runs of equal depth > 0: oracle's genomecov_bg) | awk '$4 >= cov' |
    sort | `bedtools merge` (overlapping and book-ended rows become one: documented default -d 0) | awk '$3 - $2 >= minLen', on random
    interval sets with ends beyond the chromosome, empty chromosomes and stacks of identical intervals.  Both sides state bedtools'
    documented semantics (bedtools is absent: hand-derived, like collapse_kat.json); what this pins is that the one-pass restatement
    K7 is held to equals the literal pipeline."""
    import random
    rng = random.Random(7)
    for case in range(200):
        chromlens = {c: rng.choice([50, 400, 3000]) for c in ('s1', 's10', 's2', 'Z')}
        iv = []
        for _ in range(rng.randint(0, 60)):
            c = rng.choice(list(chromlens))
            s = rng.randint(0, chromlens[c] + 20)
            e = s + rng.choice([0, 1, 5, 40, 300, 5000])
            iv += [(c, s, e)] * rng.choice([1, 1, 1, 2, 4])
        min_cov, min_len = rng.choice([1, 2, 3, 5]), rng.choice([1, 10, 100])
        got = P.coverage_collapse(iv, chromlens, min_cov, min_len)
        rows = [r for r in P.genomecov_bg(iv, chromlens) if r[3] >= min_cov]              # genomecov -bg | awk '$4 >= cov'
        rows.sort(key=lambda r: (r[0].encode(), r[1], r[2]))                                # sort -k 1,1 -k 2n,3n (LC_ALL=C)
        merged = []
        for c, s, e, _ in rows:                                                             # bedtools merge
            if merged and merged[-1][0] == c and s <= merged[-1][2]:
                merged[-1][2] = max(merged[-1][2], e)
            else:
                merged.append([c, s, e])
        exp = [(c, s, e) for c, s, e in merged if e - s >= min_len]                         # awk '$3 - $2 >= minLen'
        assert got == exp, (case, min_cov, min_len)

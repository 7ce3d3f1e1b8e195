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

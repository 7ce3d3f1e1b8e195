"""The host runtime under the CPU sanitizers: the threaded FASTA ingest
(mimeo_amd/csrc/ingest_host.h — the code ingest.hip runs, with malloc in the place of pinned memory) and the planning
logic of the pipeline (mimeo_amd/csrc/host_plan.h) are compiled with g++ -fsanitize=address,undefined and with
-fsanitize=thread and run through tests/sanitize/host_sanitize.cc, which checks them against plain restatements.
GPU AddressSanitizer is not available on the pool; the device code has no host-side counterpart to sanitize."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'sanitize', 'host_sanitize.cc')


@pytest.mark.parametrize('flags,tag', [('-fsanitize=address,undefined -fno-sanitize-recover=all', 'asan_ubsan'), ('-fsanitize=thread', 'tsan')])
def test_host_runtime_under_sanitizers(tmp_path, flags, tag):
    if shutil.which('g++') is None:
        pytest.skip('no g++')
    exe = tmp_path / ('host_sanitize_' + tag)
    cmd = ['g++', '-std=c++17', '-O1', '-g', '-pthread'] + flags.split() + [SRC, '-o', str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    work = tmp_path / 'work'
    work.mkdir()
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', TSAN_OPTIONS='halt_on_error=1', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    r = subprocess.run([str(exe), str(work)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert 'host_sanitize: ok' in r.stdout
    assert 'Sanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-4000:]


def test_oracle_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """The checker itself: oracle/mimeo_oracle.c built -fsanitize=address,undefined and driven through the cases of
    tests/test_oracle_rules.py's generator (N runs, soft-masked stretches, reverse-complemented copies, indels) against the Python
    restatement — a wild read in the oracle's band bookkeeping would otherwise only show as a rare wrong expectation."""
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    asan_rt = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan_rt) or not os.path.exists(asan_rt):
        pytest.skip('no libasan')
    lib = tmp_path / 'libmimeo_oracle_asan.so'
    r = subprocess.run(['gcc', '-O1', '-g', '-fPIC', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-shared', '-o', str(lib),
                        os.path.join(ROOT, 'oracle', 'mimeo_oracle.c'), '-lm'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    code = '''
import sys
sys.path.insert(0, %r)
from oracle import oracle as O
O.lib(%r)
from tests.test_oracle_rules import _pair
from tests import spec_v1 as S
n = 0
for seed, kw in ((5, {}), (6, {'lower': True, 'ns': True, 'indel': 0.03}), (7, {'rc_copy': True, 'indel': 0.02})):
    T, Q = _pair(seed, 800, 700, 4, cons=(120, 300), **kw)
    for minus in (0, 1):
        q = S.revcomp(Q.decode()) if minus else Q.decode()
        got = O.seed_hits(T, Q, minus)
        assert [(int(a), int(b)) for a, b in zip(got['tpos'], got['qpos'])] == S.seed_hits(T.decode(), q)
        got = O.ungapped_hsps(T, Q, minus, O.default_params(chain=0))
        assert sorted((int(h['tstart']), int(h['qstart']), int(h['length']), int(h['score']), int(h['raw_score'])) for h in got) == sorted(S.ungapped_hsps(T.decode(), q))
    got = O.align_pair(T, Q)
    exp = S.align_strand(T.decode(), Q.decode(), 0) + S.align_strand(T.decode(), Q.decode(), 1)
    assert sorted((int(a['tstart']), int(a['tend']), int(a['qstart']), int(a['qend']), int(a['score']), int(a['id_n']), int(a['id_d']), int(a['qstrand'])) for a in got) == sorted(exp)
    n += len(exp)
h = O.chain_hsps(O.ungapped_hsps(T, Q, 0, O.default_params(chain=0)))
print('oracle_sanitize: ok', n)
''' % (ROOT, str(lib))
    env = dict(os.environ, LD_PRELOAD=asan_rt, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    r = subprocess.run([os.sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert 'oracle_sanitize: ok' in r.stdout
    assert 'Sanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-4000:]

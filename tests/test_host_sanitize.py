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

"""Streaming FASTA ingest (replaces utils.py:274-309 splitFasta and the Biopython parses):
the natively parsed genome must behave exactly like the one packed from in-memory arrays, for awkward
FASTA text too (CRLF, blank lines, ragged line widths, lower case, N, empty records, no final newline)."""
import os

import numpy as np
import pytest

from mimeo_amd import engine, formats
from mimeo_amd.synth import synth_genome, write_fasta

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    engine.init(0)
    return engine


def _same_hits(G1, G2, n):
    for t in range(n):
        for q in range(n):
            for strand in (0, 1):
                a = engine.ungapped_hsps(G1, t, G1, q, strand)
                b = engine.ungapped_hsps(G2, t, G2, q, strand)
                assert a.tobytes() == b.tobytes()


def test_fasta_ingest_equals_in_memory_pack(eng, tmp_path):
    names, seqs = synth_genome(7, 600_000, 3)
    seqs = [s.copy() for s in seqs]
    seqs[1][1000:3000] = np.frombuffer(bytes(seqs[1][1000:3000]).lower(), dtype=np.uint8)  # soft-masked stretch
    seqs[2][500:520] = ord('N')
    fa = tmp_path / 'g.fa'
    write_fasta(str(fa), names, seqs)
    A = engine.Genome(names, seqs)
    B = engine.Genome.from_fasta(str(fa), split_dir=str(tmp_path))
    assert B.names == names and B.lengths == A.lengths
    _same_hits(A, B, 3)
    # the --adir side effect: one <id>.fa per record, parseable, same bases
    for n, s in zip(names, seqs):
        rn, rs = formats.read_fasta(os.path.join(str(tmp_path), n + '.fa'))
        assert rn == [n] and rs[0].tobytes() == s.tobytes()
    A.close()
    B.close()


def test_fasta_ingest_awkward_text(eng, tmp_path):
    names, seqs = synth_genome(8, 200_000, 2)
    a, b = seqs[0].tobytes(), seqs[1].tobytes()
    text = (b'; comment before the first record\n\n>' + names[0].encode() + b' some description\tmore\r\n' +
            b'\r\n'.join(a[i:i + 71] for i in range(0, len(a), 71)) + b'\r\n\r\n' +
            b'>empty_one\n' +
            b'>' + names[1].encode() + b'\n' + b'\n'.join(b[i:i + 13_337] for i in range(0, len(b), 13_337)))  # no final newline
    fa = tmp_path / 'awk.fa'
    fa.write_bytes(text)
    G = engine.Genome.from_fasta([str(fa)])
    assert G.names == [names[0], 'empty_one', names[1]]
    assert G.lengths == [len(a), 0, len(b)]
    R = engine.Genome([names[0], 'empty_one', names[1]], [seqs[0], np.zeros(0, np.uint8), seqs[1]])
    for t, q in ((0, 2), (2, 0), (0, 0)):
        assert engine.ungapped_hsps(G, t, G, q, 1).tobytes() == engine.ungapped_hsps(R, t, R, q, 1).tobytes()
    assert engine.seed_hits(G, 1, G, 0).size == 0
    # python-side reader agrees on the same text
    pn, ps = formats.read_fasta(str(fa))
    assert pn == G.names and [len(x) for x in ps] == G.lengths
    G.close()
    R.close()


def test_fasta_ingest_errors(eng, tmp_path):
    fa = tmp_path / 'dup.fa'
    fa.write_bytes(b'>x\nACGT\n>y\nAC\n>x\nGG\n')
    with pytest.raises(RuntimeError, match='Non-unique name in genome: x'):
        engine.Genome.from_fasta(str(fa))
    with pytest.raises(RuntimeError, match='cannot open'):
        engine.Genome.from_fasta(str(tmp_path / 'missing.fa'))
    # several files, in the order given (a --adir directory)
    (tmp_path / 'b.fa').write_bytes(b'>b1\nACGTACGTAC\n')
    (tmp_path / 'a.fa').write_bytes(b'>a1\nAC\n>a2\nGGG\n')
    G = engine.Genome.from_fasta([str(tmp_path / 'a.fa'), str(tmp_path / 'b.fa')])
    assert G.names == ['a1', 'a2', 'b1'] and G.lengths == [2, 3, 10]
    G.close()

"""Synthetic genomes for tests and bench.py (SURVEY.md §8d "Synthetic inputs").

Upper-case iid-uniform ACGT scaffolds ``scafNNNN`` of equal length with planted repeat
families: ``families`` consensus sequences of length ~U[300, 6000]; copies are placed at
uniform non-overlapping positions on a random strand until ``repeat_frac`` of the genome is
covered; every copy carries substitution divergence ~U[0, max_div] and ``indel_rate`` indels
of length 1-3.  Deterministic in ``seed`` (numpy PCG64).
"""
import numpy as np

_ACGT = np.frombuffer(b'ACGT', dtype=np.uint8)
_COMP = np.array([3, 2, 1, 0], dtype=np.uint8)


def _mutate(rng, cons, div, indel_rate):
    c = cons.copy()
    n = c.size
    sub = rng.random(n) < div
    k = int(sub.sum())
    if k:
        c[sub] = (c[sub] + rng.integers(1, 4, size=k, dtype=np.uint8)) & 3
    nindel = rng.binomial(n, indel_rate)
    if nindel:
        pos = np.sort(rng.integers(0, n, size=nindel))
        pieces, last = [], 0
        for p in pos:
            p = int(p)
            if p < last:
                continue
            ln = int(rng.integers(1, 4))
            pieces.append(c[last:p])
            if rng.random() < 0.5:  # deletion
                last = min(n, p + ln)
            else:  # insertion
                pieces.append(rng.integers(0, 4, size=ln, dtype=np.uint8))
                last = p
        pieces.append(c[last:])
        c = np.concatenate(pieces)
    return c


def synth_genome(seed, total_bp, nscaf, repeat_frac=0.05, families=40, max_div=0.15,
                 indel_rate=0.005, cons_len=(300, 6000), prefix='scaf', shared_families=None, microsat_frac=0.0):
    """Return (names, [uint8 ASCII arrays]).  ``shared_families`` lets two genomes (mimeo x)
    carry copies of the same consensus set: pass the list returned by ``make_families``.
    ``microsat_frac`` (C5: 0.01) overwrites that share of the genome with perfect microsatellites
    of period 1-6 and length 50-500 bp."""
    rng = np.random.Generator(np.random.PCG64(seed))
    L = total_bp // nscaf
    codes = [rng.integers(0, 4, size=L, dtype=np.uint8) for _ in range(nscaf)]
    fams = shared_families if shared_families is not None else make_families(rng, families, cons_len)
    gran = 32
    occ = [np.zeros(L // gran + 2, dtype=bool) for _ in range(nscaf)]
    target = int(repeat_frac * L * nscaf)
    covered, tries = 0, 0
    while covered < target and fams and tries < 50 * (target // cons_len[0] + 10):
        tries += 1
        f = fams[int(rng.integers(0, len(fams)))]
        cp = _mutate(rng, f, float(rng.random()) * max_div, indel_rate)
        if rng.random() < 0.5:
            cp = _COMP[cp[::-1]]
        if cp.size >= L:
            continue
        s = int(rng.integers(0, nscaf))
        p = int(rng.integers(0, L - cp.size))
        a, b = p // gran, (p + cp.size - 1) // gran + 1
        if occ[s][a:b].any():
            continue
        occ[s][a:b] = True
        codes[s][p:p + cp.size] = cp
        covered += cp.size
    target, covered = int(microsat_frac * L * nscaf), 0
    while covered < target:
        period, ln = int(rng.integers(1, 7)), int(rng.integers(50, 501))
        unit = rng.integers(0, 4, size=period, dtype=np.uint8)
        if period > 1 and (unit == unit[0]).all():
            unit[-1] = (unit[0] + 1) & 3
        s, p = int(rng.integers(0, nscaf)), int(rng.integers(0, max(1, L - ln)))
        ln = min(ln, L - p)
        codes[s][p:p + ln] = np.resize(unit, ln)
        covered += ln
    names = ['%s%04d' % (prefix, i) for i in range(nscaf)]
    return names, [_ACGT[c] for c in codes]


def make_families(rng, families=40, cons_len=(300, 6000)):
    if isinstance(rng, (int, np.integer)):
        rng = np.random.Generator(np.random.PCG64(int(rng)))
    return [rng.integers(0, 4, size=int(rng.integers(cons_len[0], cons_len[1] + 1)), dtype=np.uint8)
            for _ in range(families)]


def write_fasta(path, names, seqs, width=60):
    with open(path, 'wb') as f:
        for n, s in zip(names, seqs):
            f.write(b'>' + n.encode() + b'\n')
            b = s.tobytes()
            for i in range(0, len(b), width * 1000):
                chunk = b[i:i + width * 1000]
                f.write(b'\n'.join(chunk[j:j + width] for j in range(0, len(chunk), width)) + b'\n')

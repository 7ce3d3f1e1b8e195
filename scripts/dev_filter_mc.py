"""development: Monte Carlo of pre-filter variants on random seed hits (fractions of hits passed on to the exact walk).
Frame: 192 columns, seed window = columns 109..127, left walk = columns 127 down, right walk = columns 128 up."""
import numpy as np
rng = np.random.default_rng(1)
N = 400_000
CARE = np.array([0, 1, 2, 4, 7, 8, 11, 13, 15, 16, 17, 18])
t = rng.integers(0, 4, size=(N, 192), dtype=np.int8)
q = rng.integers(0, 4, size=(N, 192), dtype=np.int8)
# plant the seed hit at columns 109..127 (bit i of the frame = column i)
q[:, 109 + CARE] = t[:, 109 + CARE]
kind = rng.integers(0, 13, size=N)          # 0 = exact word, k = transition at care position k-1
rows = np.flatnonzero(kind > 0)
cols = 109 + CARE[kind[rows] - 1]
q[rows, cols] = t[rows, cols] ^ 2           # code A0 C1 G2 T3: a transition flips bit 1
dl = (t ^ q) & 1
dh = ((t ^ q) >> 1) & 1
cg = ((t & 1) ^ (t >> 1)) & 1
mism = dl | dh
match = 1 - mism
tr = (1 - dl) & dh
tv = dl
a = match & cg
b = dl & dh
c = b & cg
score = 91 * match + 9 * a - 31 * tr - 114 * tv + (114 - 123) * b - 2 * c
up = 91 * match + 9 * a - 31 * tr - 114 * tv - 9 * b           # drops the -2c term
XD, TH = 910, 3000

def bounds(cols_blocks, exact=True):
    """cols_blocks: list of column index arrays in walk order; returns stop, ub."""
    U = np.zeros(N, np.int64); lomax = np.zeros(N, np.int64); ub = np.zeros(N, np.int64); nb = np.zeros(N, np.int64)
    stop = np.zeros(N, bool); stop_at = np.full(N, 99)
    for k, cb in enumerate(cols_blocks):
        m = match[:, cb].sum(1)
        ub = np.maximum(ub, U + 100 * m)
        if exact:
            U = U + up[:, cb].sum(1)
            nb = nb + b[:, cb].sum(1)
            lo = U - 2 * nb
            newstop = U + XD < lomax
        else:
            # loose: upper: match 100, transition -31, transversion -114; lower: 91 / -31 / -125
            Uup = U + (100 * match[:, cb] - 31 * tr[:, cb] - 114 * tv[:, cb]).sum(1)
            nb = nb + (9 * match[:, cb] + 11 * tv[:, cb]).sum(1)   # accumulated slack
            U = Uup
            lo = U - nb
            newstop = U + XD < lomax
        stop_at = np.where(~stop & newstop, k, stop_at)
        stop |= newstop
        lomax = np.maximum(lomax, lo)
    return stop, ub, stop_at

def alarm(care, use_tv):
    """per left step s (0..95): superset alarm for the seed starting at column 108 - s"""
    al = np.zeros((N, 96), bool)
    for s in range(96):
        st = 108 - s
        mm = mism[:, st + care].sum(1)
        bad = mm >= 2
        if use_tv:
            bad |= tv[:, st + care].sum(1) >= 1
        al[:, s] = ~bad
    return al

L16 = [np.arange(127 - 16 * j, 111 - 16 * j, -1) for j in range(6)]
R16 = [np.arange(128 + 16 * j, 144 + 16 * j) for j in range(4)]
Lmix = L16[:4] + [np.arange(63, 31, -1)]
Rmix = R16[:2] + [np.arange(160, 192)]
C12, C10, C8 = CARE, CARE[:10], np.array([0, 1, 2, 4, 7, 8, 11, 13])

def run(tag, Lb, Rb, care, use_tv, exact=True, steps_per_block=None):
    ls, lub, lat = bounds(Lb, exact)
    rs, rub, _ = bounds(Rb, exact)
    al = alarm(care, use_tv)
    # boundaries count while the stop is not yet proven: steps before the end of the proving block
    ends = np.cumsum([len(x) for x in Lb])
    reach = np.where(lat < 99, ends[np.minimum(lat, len(Lb) - 1)], 96)
    veto = (al & (np.arange(96)[None, :] < reach[:, None])).any(1)
    need = ~(ls & rs & (lub + rub < TH) & ~veto)
    print('%-34s pass %.3f%%   (left unproven %.3f%%  right %.3f%%  bound %.3f%%  alarm %.3f%%)' % (
        tag, 100 * need.mean(), 100 * (~ls).mean(), 100 * (~rs).mean(), 100 * (lub + rub >= TH).mean(), 100 * veto.mean()))

run('exact16  C8+tv  (round 1 style)', L16, R16, C8, True)
run('exact16  C10+tv (now)', L16, R16, C10, True)
run('exact16  C10 no tv', L16, R16, C10, False)
run('exact16  C12 no tv', L16, R16, C12, False)
run('exact16  C12+tv', L16, R16, C12, True)
run('mixed    C10 no tv', Lmix, Rmix, C10, False)
run('loose16  C10 no tv', L16, R16, C10, False, exact=False)
run('loose mixed C10 no tv', Lmix, Rmix, C10, False, exact=False)

print()
def cost(nl, nr, ncare, use_tv, exact, ngroups):
    return 17 + ngroups * ncare * (6 if use_tv else 4) + (nl + nr) * (25 if exact else 15)
def run2(tag, Lb, Rb, care, use_tv, exact):
    ls, lub, lat = bounds(Lb, exact)
    rs, rub, _ = bounds(Rb, exact)
    nsteps = sum(len(x) for x in Lb)
    al = alarm(care, use_tv)[:, :nsteps]
    ends = np.cumsum([len(x) for x in Lb])
    reach = np.where(lat < 99, ends[np.minimum(lat, len(Lb) - 1)], nsteps)
    veto = (al & (np.arange(nsteps)[None, :] < reach[:, None])).any(1)
    need = ~(ls & rs & (lub + rub < TH) & ~veto)
    c = cost(len(Lb), len(Rb), len(care), use_tv, exact, (nsteps + 31) // 32)
    print('%-40s pass %6.3f%%  filter %3d + walks %5.1f = %5.1f instr / batch' % (tag, 100 * need.mean(), c, 600 * need.mean(), c + 600 * need.mean()))
for exact in (True, False):
    for nl, nr in ((6, 4), (4, 4), (4, 3), (6, 3), (5, 3)):
        for care, cn in ((C8, 'C8'), (C10, 'C10'), (C12, 'C12')):
            for tvf in (False, True):
                run2('%s L%d R%d %s%s' % ('exact' if exact else 'loose', nl * 16, nr * 16, cn, '+tv' if tvf else ''), L16[:nl], R16[:nr], care, tvf, exact)

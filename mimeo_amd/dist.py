"""Multi-GPU: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards by TARGET scaffold: a target's seed index is built once on its owner
and coverage is per target, so nothing is exchanged while aligning.  The only collective is one
all-gatherv of the packed alignment records at the end (all_gather of counts, then all_gather of
max-padded byte buffers — RCCL has no v-variant)."""
import os

import numpy as np


class Dist:
    def __init__(self):
        self.rank = int(os.environ.get('RANK', '0'))
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.backend = None
        self._torch = None
        # MIMEO_DIST_FORCE=1 creates the process group even for a single rank, so that the RCCL code
        # path (device tensors, all_gather, all_reduce, barrier) can be exercised on a one-GPU box
        self.active = False
        self._force = os.environ.get('MIMEO_DIST_FORCE') == '1'

    def init(self, backend=None):
        if self.world <= 1 and not self._force:
            return self
        import torch
        import torch.distributed as td
        self._torch = torch
        if backend is None:
            backend = os.environ.get('MIMEO_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        self.backend = backend
        if not td.is_initialized():
            if backend == 'nccl':
                torch.cuda.set_device(self.local_rank)
            td.init_process_group(backend=backend)
        self.active = True
        # first collective now: communicator setup (seconds with RCCL) stays out of any timed region
        warm = torch.zeros(1, dtype=torch.int64, device=self.device)
        td.all_reduce(warm)
        if backend == 'nccl':
            torch.cuda.synchronize()
        return self

    @property
    def device(self):
        if self.backend == 'nccl':
            return self._torch.device('cuda', self.local_rank)
        return 'cpu'

    def barrier(self):
        if self.active:
            import torch.distributed as td
            td.barrier()

    def max_float(self, x):
        """max over ranks of a python float (bench timing)."""
        if not self.active:
            return x
        import torch.distributed as td
        t = self._torch.tensor([x], dtype=self._torch.float64, device=self.device)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        return float(t.item())

    def sum_int(self, x):
        if not self.active:
            return int(x)
        import torch.distributed as td
        t = self._torch.tensor([int(x)], dtype=self._torch.int64, device=self.device)
        td.all_reduce(t, op=td.ReduceOp.SUM)
        return int(t.item())

    def allgather_records(self, arr):
        """Concatenate a structured numpy array over ranks, in rank order, on every rank."""
        if not self.active:
            return arr
        import torch.distributed as td
        torch = self._torch
        raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        n = torch.tensor([raw.size], dtype=torch.int64, device=self.device)
        sizes = [torch.zeros(1, dtype=torch.int64, device=self.device) for _ in range(self.world)]
        td.all_gather(sizes, n)
        sizes = [int(s.item()) for s in sizes]
        cap = max(max(sizes), 1)
        buf = torch.zeros(cap, dtype=torch.uint8, device=self.device)
        if raw.size:
            buf[:raw.size] = torch.from_numpy(raw.copy()).to(self.device)
        outs = [torch.zeros(cap, dtype=torch.uint8, device=self.device) for _ in range(self.world)]
        td.all_gather(outs, buf)
        parts = [o[:s].cpu().numpy().view(arr.dtype) for o, s in zip(outs, sizes)]
        return np.concatenate(parts) if parts else arr


def shard_pairs_by_target(pairs, cost_of_target, world, rank):
    """Longest-processing-time assignment of target scaffolds to ranks; returns this rank's
    pairs in their original order.  `cost_of_target[t]` ~ Lt * sum of its query lengths."""
    if world <= 1:
        return list(pairs)
    targets = sorted({t for t, _ in pairs}, key=lambda t: (-cost_of_target[t], t))
    load = [0] * world
    owner = {}
    for t in targets:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[t] = r
        load[r] += cost_of_target[t]
    return [(t, q) for t, q in pairs if owner[t] == rank]


PLUS, MINUS, BOTH = 1, 2, 3


def plus_owner(a, b, nscaf):
    """Which of the rows a, b computes the plus-strand unit of the unordered scaffold pair {a, b} of a self job (and emits it
    for both orders: the HSP sets of (a, b, +) and (b, a, +) are transposes of one another, DESIGN.md "Shared plus strand").
    Circulant dealing: row a takes the pairs {a, a + d mod S} for d < S / 2, so every row owns (S - 1) / 2 of them; the
    pairs at distance exactly S / 2 (S even) go to the even member, or to the smaller one when both have the same parity."""
    if a == b:
        return a
    d = (b - a) % nscaf
    if 2 * d < nscaf:
        return a
    if 2 * d > nscaf:
        return b
    lo, hi = min(a, b), max(a, b)
    if (lo % 2) != (hi % 2):
        return lo if lo % 2 == 0 else hi
    return lo


def units_of_row(t, nscaf):
    """The units of target row t in a self job of `nscaf` scaffolds, as (target, query, strands) for mimeo_align_units:
    every minus-strand unit (t, q, -), the row's own (t, t, +), and for every unordered pair {t, q} this row owns the
    plus-strand unit in BOTH orders.  The rows together name each of the 2 S^2 units of the job exactly once."""
    units = []
    for q in range(nscaf):
        own = plus_owner(t, q, nscaf) == t
        units.append((t, q, BOTH if own else MINUS))
    for q in range(nscaf):
        if q != t and plus_owner(t, q, nscaf) == t:
            units.append((q, t, PLUS))
    return units


def deal_units(nscaf, cost_of_target, world, rank):
    """A rank's share of a self job: target rows by longest-processing-time (as shard_pairs_by_target), each with
    units_of_row.  No data-path collective: a row's units need nothing from another rank; the rows' alignments meet in the
    all-gatherv at the end."""
    rows = list(range(nscaf))
    if world > 1:
        targets = sorted(rows, key=lambda t: (-cost_of_target[t], t))
        load = [0] * world
        owner = {}
        for t in targets:
            r = min(range(world), key=lambda k: (load[k], k))
            owner[t] = r
            load[r] += cost_of_target[t]
        rows = [t for t in rows if owner[t] == rank]
    units = []
    for t in rows:
        units += units_of_row(t, nscaf)
    return units

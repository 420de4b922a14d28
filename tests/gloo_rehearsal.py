"""Test helper: the i-partition of the targets over several ranks, with the collectives carried by
torch.distributed (gloo) instead of libnbmg's RCCL communicator.  The product's multi-GPU path is
libnbmg (include/nbmg.h); this module exists so that the partition arithmetic and the sequence of
collectives can be rehearsed where RCCL cannot run: two gloo ranks on the CPU (tests/test_distributed.py)
and several ranks sharing ONE GPU (bench.py with NBODY_BENCH_ONE_DEVICE=1; RCCL refuses two ranks
on one device).

The reference is single-process (one `omp target` region, src/particles.cpp:169-173).  Each rank owns
a contiguous slice of the particle index range, keeps its slice of x, y, z resident, and per evaluation

  1. all-gathers the three coordinate arrays (masses are gathered once),
  2. computes accelerations for its own slice against all sources,

with no reduction afterwards.  `compute` is injected: on a GPU it is
`Library.accumulate_device`; nothing here touches the arithmetic.
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import torch
import torch.distributed as dist


def partition(n: int, world: int) -> tuple[list[int], list[int]]:
    """Contiguous near-equal slices: (begins, counts); the first n % world ranks get one more."""
    counts = [n // world + (1 if r < n % world else 0) for r in range(world)]
    begins = [sum(counts[:r]) for r in range(world)]
    return begins, counts


class SlicedParticles:
    """A rank's view of the particle set: its own slice plus gather buffers for all sources."""

    def __init__(self, n: int, rank: int, world: int, device, dtype=torch.float64):
        self.n, self.rank, self.world = n, rank, world
        self.begins, self.counts = partition(n, world)
        self.begin, self.count = self.begins[rank], self.counts[rank]
        self.stride = max(self.counts)  # all_gather needs equal contributions: pad to the max
        self.device = device
        self.own = [torch.zeros(self.stride, dtype=dtype, device=device) for _ in range(3)]
        self._gathered = [torch.empty(self.stride * world, dtype=dtype, device=device)
                          for _ in range(3)]
        self._full = [torch.empty(n, dtype=dtype, device=device) for _ in range(3)]

    def set_own(self, x, y, z):
        for buf, src in zip(self.own, (x, y, z)):
            buf[: self.count].copy_(src)

    def set_own_from_global(self, x, y, z):
        b, c = self.begin, self.count
        self.set_own(x[b:b + c], y[b:b + c], z[b:b + c])

    def gather_scalar_field(self, own_values: torch.Tensor) -> torch.Tensor:
        """All-gather one per-particle array (used once for the masses)."""
        pad = torch.zeros(self.stride, dtype=own_values.dtype, device=self.device)
        pad[: self.count].copy_(own_values)
        return self._compact(self._all_gather(pad))

    def _all_gather(self, own: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return own
        out = torch.empty(self.stride * self.world, dtype=own.dtype, device=self.device)
        dist.all_gather_into_tensor(out, own)
        return out

    def _compact(self, gathered: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return gathered[: self.n]
        if self.n % self.world == 0:
            return gathered
        out = torch.empty(self.n, dtype=gathered.dtype, device=self.device)
        for r in range(self.world):
            out[self.begins[r]: self.begins[r] + self.counts[r]] = \
                gathered[r * self.stride: r * self.stride + self.counts[r]]
        return out

    def gather_positions(self) -> List[torch.Tensor]:
        """x, y, z of ALL particles, in global index order, on this rank's device."""
        if self.world == 1:
            return [o[: self.n] for o in self.own]
        for c in range(3):
            dist.all_gather_into_tensor(self._gathered[c], self.own[c])
        if self.n % self.world == 0:
            return self._gathered
        for c in range(3):
            for r in range(self.world):
                self._full[c][self.begins[r]: self.begins[r] + self.counts[r]] = \
                    self._gathered[c][r * self.stride: r * self.stride + self.counts[r]]
        return self._full


def force_evaluation(parts: SlicedParticles, mass_all: torch.Tensor,
                     acc: Sequence[torch.Tensor], compute: Callable) -> None:
    """One PP force evaluation for this rank's slice: gather, zero, accumulate.

    `compute(x, y, z, m, ax, ay, az, tgt_begin, n_tgt)` adds the PP accelerations of targets
    [tgt_begin, tgt_begin + n_tgt) into ax/ay/az[0:n_tgt].  Zeroing first is the caller contract
    of PP-only mode (src/integrator.cpp:246-251)."""
    x, y, z = parts.gather_positions()
    for a in acc:
        a.zero_()
    compute(x, y, z, mass_all, acc[0], acc[1], acc[2], parts.begin, parts.count)


def force_evaluation_pairs(parts: SlicedParticles, mass_all: torch.Tensor,
                           acc: Sequence[torch.Tensor], compute_pairs: Callable,
                           work: List[torch.Tensor] | None = None) -> List[torch.Tensor]:
    """One direct-summation evaluation that uses every unordered pair once across the ranks.

    `compute_pairs(x, y, z, m, fx, fy, fz, part, n_parts)` adds, for this rank's share of the
    pairs, the accelerations of BOTH members of each pair into the full-length fx/fy/fz
    (Library.accumulate_pairs_device on a GPU).  The shares are then summed across ranks with a
    reduce-scatter, which leaves every rank with the complete accelerations of its own slice in
    `acc` (overwritten: PP-only mode starts from zero, src/integrator.cpp:246-251).

    `work` are three reusable buffers of stride*world elements (returned for reuse)."""
    x, y, z = parts.gather_positions()
    total = parts.stride * parts.world
    if work is None:
        work = [torch.empty(total, dtype=x.dtype, device=parts.device) for _ in range(3)]
    exact = parts.n % parts.world == 0      # slices are equal: padded layout == global layout
    full = work if exact else [torch.empty(parts.n, dtype=x.dtype, device=parts.device) for _ in range(3)]
    for f in full:
        f.zero_()
    compute_pairs(x, y, z, mass_all, full[0], full[1], full[2], parts.rank, parts.world)
    for c in range(3):
        if parts.world == 1:
            acc[c].copy_(full[c][: parts.count])
            continue
        if not exact:   # expand to the rank-major padded layout the collective needs
            work[c].zero_()
            for r in range(parts.world):
                work[c][r * parts.stride: r * parts.stride + parts.counts[r]] = \
                    full[c][parts.begins[r]: parts.begins[r] + parts.counts[r]]
        if dist.get_backend() == "gloo":
            # no usable reduce-scatter there (CPU tensors: not implemented; CUDA tensors: observed
            # to block forever): all-reduce and keep the own slice
            dist.all_reduce(work[c])
            out = work[c][parts.rank * parts.stride: (parts.rank + 1) * parts.stride]
        else:
            out = torch.empty(parts.stride, dtype=x.dtype, device=parts.device)
            dist.reduce_scatter_tensor(out, work[c])
        acc[c].copy_(out[: parts.count])
    return work

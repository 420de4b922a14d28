#!/usr/bin/env python3
"""P3M short range on a heavily clustered particle set (late-time-like), timed for 1, 2, 4 and 8
lanes per target: shows whether lane imbalance or arithmetic bounds the kernel there.

usage: tools/p3m_clustered_probe.py [n_side] [sigma]
       tools/p3m_clustered_probe.py --snapshot <file written by `nbody_host_tests run ... <prefix>`>
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd  # noqa: E402
from nbody_amd import Config, ffi  # noqa: E402
import datasets  # noqa: E402

if len(sys.argv) > 2 and sys.argv[1] == "--snapshot":
    raw = np.fromfile(sys.argv[2], dtype=np.float64)
    n, n_side = int(raw[0]), int(raw[1])
    pos = [np.ascontiguousarray(raw[6 + c * n: 6 + (c + 1) * n]) for c in range(3)]
    print(f"snapshot: {n} particles, mesh {n_side}, a = {raw[2]:.4f}")
else:
    n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.004
    n = n_side ** 3
    pos = datasets.clustered(n, n_blobs=400, sigma=sigma, seed=3)
config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True)
p = config.pp_params()
lib = nbody_amd.load(device=0)
lib.set_option(ffi.OPT_TIMING, 1)
if os.environ.get("P3M_PROBE_RADIUS"):   # search radius in cells (default 3)
    lib.set_option(ffi.OPT_P3M_SEARCH_RADIUS, int(os.environ["P3M_PROBE_RADIUS"]))
d = [torch.as_tensor(a, device="cuda:0") for a in pos] + [torch.full((n,), config.DM_PARTICLE_MASS, dtype=torch.float64, device="cuda:0")]
acc = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
for lanes in (0, 1, 2, 4, 8, 16):
    lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, lanes)
    for _ in range(3):
        lib.accumulate_device(p, *d, *acc)
        torch.cuda.synchronize()
    st = lib.stats()
    print(f"lanes {lanes}: pairs kernel {st['pair_kernel_ms']:.3f} ms, binning {st['prepare_ms']:.3f} ms, "
          f"{st['pairs_accepted']:.4g} accepted pairs, {st['pairs_accepted'] / st['pair_kernel_ms'] * 1e-6:.1f} G/s in the kernel")

if os.environ.get("P3M_PROBE_QUICK"):
    sys.exit(0)

# Where the time goes: particles regrouped by the occupancy of their own cell, each class then
# evaluated as a target slice of its own against all sources.
lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, 0)
cell = [np.minimum((a / config.CELL_SIZE).astype(np.int64), n_side - 1) for a in pos]
key = (cell[0] * n_side + cell[1]) * n_side + cell[2]
occupancy = np.bincount(key, minlength=n_side ** 3)[key]
cls = np.floor(np.log2(occupancy)).astype(np.int64)
perm = np.argsort(cls, kind="stable")
d = [torch.as_tensor(np.ascontiguousarray(a[perm]), device="cuda:0") for a in pos] + [d[3]]
bounds = np.searchsorted(cls[perm], np.arange(cls.max() + 2))
print("own-cell occupancy | targets | accepted/target | kernel ms | G accepted/s")
total = 0.0
for k in range(cls.max() + 1):
    b, e = int(bounds[k]), int(bounds[k + 1])
    if e == b:
        continue
    for _ in range(3):
        lib.accumulate_device(p, *d, *acc, tgt_begin=b, n_tgt=e - b)
        torch.cuda.synchronize()
    st = lib.stats()
    total += st["pair_kernel_ms"]
    print(f"{2 ** k:5d}..{2 ** (k + 1) - 1:<5d} | {e - b:8d} | {st['pairs_accepted'] / (e - b):8.1f} | {st['pair_kernel_ms']:.3f} | "
          f"{st['pairs_accepted'] / st['pair_kernel_ms'] * 1e-6:.1f}")
print(f"sum of the classes: {total:.3f} ms")

# Throughput or critical path?  The sparse and the dense part separately, by lanes per target.
for name, lo, hi in (("own cell holds < 32", 0, 5), ("own cell holds >= 32", 5, cls.max() + 1)):
    b, e = int(bounds[min(lo, cls.max() + 1)]), int(bounds[hi])
    if e == b:
        continue
    for lanes in (1, 2, 4, 8):
        lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, lanes)
        for _ in range(3):
            lib.accumulate_device(p, *d, *acc, tgt_begin=b, n_tgt=e - b)
            torch.cuda.synchronize()
        st = lib.stats()
        print(f"{name}: {e - b} targets, {lanes} lanes each: {st['pair_kernel_ms']:.3f} ms, "
              f"{st['pairs_accepted'] / st['pair_kernel_ms'] * 1e-6:.1f} G accepted/s")

# Lane balance of the row-synchronised walk: a wavefront spends max-over-lanes(row length) trips
# on each of its 49 rows; compare with the mean over lanes (no imbalance) and with the largest
# per-lane total (lanes walking their rows independently).
counts = np.bincount(key, minlength=n_side ** 3).reshape(n_side, n_side, n_side)
rowlen = sum(np.roll(counts, dz, axis=2) for dz in range(-3, 4))
order = np.argsort(key, kind="stable")
tc = [c[order] for c in cell]
lens = np.empty((n, 49), dtype=np.int32)
j = 0
for dx in range(-3, 4):
    for dy in range(-3, 4):
        lens[:, j] = rowlen[(tc[0] + dx) % n_side, (tc[1] + dy) % n_side, tc[2]]
        j += 1
pad = (-n) % 64
lens = np.concatenate([lens, np.zeros((pad, 49), dtype=np.int32)]).reshape(-1, 64, 49)
sum_of_max = lens.max(axis=1).sum(axis=1).astype(np.float64)
max_of_sum = lens.sum(axis=2).max(axis=1).astype(np.float64)
mean = lens.sum(axis=(1, 2)) / 64.0
print(f"candidate trips per wavefront, summed over {lens.shape[0]} wavefronts: balanced {mean.sum():.4g}, "
      f"independent lanes {max_of_sum.sum():.4g} ({max_of_sum.sum() / mean.sum():.2f}x), "
      f"row-synchronised {sum_of_max.sum():.4g} ({sum_of_max.sum() / mean.sum():.2f}x); "
      f"heaviest wavefront {sum_of_max.max():.0f} trips, mean {sum_of_max.mean():.0f}")


def trips(order_key, label):
    o = np.argsort(order_key, kind="stable")
    t = [c[o] for c in cell]
    L = np.empty((n, 49), dtype=np.int32)
    j = 0
    for dx in range(-3, 4):
        for dy in range(-3, 4):
            L[:, j] = rowlen[(t[0] + dx) % n_side, (t[1] + dy) % n_side, t[2]]
            j += 1
    L = np.concatenate([L, np.zeros(((-n) % 64, 49), dtype=np.int32)]).reshape(-1, 64, 49)
    som = L.max(axis=1).sum(axis=1).astype(np.float64).sum()
    mos = L.sum(axis=2).max(axis=1).astype(np.float64).sum()
    bal = L.sum() / 64.0
    print(f"targets ordered by {label}: row-synchronised {som / bal:.2f}x, independent lanes {mos / bal:.2f}x of balanced")


trips(key, "cell, z fastest")
for bx, by, bz in ((2, 2, 16), (4, 4, 4), (2, 2, 4), (8, 8, 1), (4, 4, 1), (2, 2, 2), (1, 1, 8)):
    blk = ((cell[0] // bx) * n_side + cell[1] // by) * n_side + cell[2] // bz
    trips(blk * (n_side ** 3) + key, f"{bx}x{by}x{bz} blocks of cells")

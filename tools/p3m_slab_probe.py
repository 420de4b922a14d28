#!/usr/bin/env python3
"""One rank's share of a multi-GPU P3M evaluation, timed on one GPU: all sources, the targets of
rank 0 of `world` (a contiguous index slice of a lattice-ordered set, i.e. a slab in x).

usage: tools/p3m_slab_probe.py [n_side] [world]
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

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = n_side ** 3
pos = datasets.jittered_lattice(n_side)
config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True)
p = config.pp_params()
lib = nbody_amd.load(device=0)
lib.set_option(ffi.OPT_TIMING, 1)
d = [torch.as_tensor(a, device="cuda:0") for a in pos] + [torch.full((n,), config.DM_PARTICLE_MASS, dtype=torch.float64, device="cuda:0")]
for share in (world, 1):
    count = n // share
    acc = [torch.zeros(count, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    for _ in range(4):
        lib.accumulate_device(p, *d, *acc, tgt_begin=0, n_tgt=count)
        torch.cuda.synchronize()
    st = lib.stats()
    print(f"{n_side}^3, targets 1/{share} of the particles: binning {st['prepare_ms']:.3f} ms, pairs kernel "
          f"{st['pair_kernel_ms']:.3f} ms, {st['pairs_accepted']:.4g} accepted pairs")

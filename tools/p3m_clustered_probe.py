#!/usr/bin/env python3
"""P3M short range on a heavily clustered particle set (late-time-like), timed for 1, 2, 4 and 8
lanes per target: shows whether lane imbalance or arithmetic bounds the kernel there.

usage: tools/p3m_clustered_probe.py [n_side] [sigma]
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

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.004
n = n_side ** 3
pos = datasets.clustered(n, n_blobs=400, sigma=sigma, seed=3)
config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True)
p = config.pp_params()
lib = nbody_amd.load(device=0)
lib.set_option(ffi.OPT_TIMING, 1)
d = [torch.as_tensor(a, device="cuda:0") for a in pos] + [torch.full((n,), config.DM_PARTICLE_MASS, dtype=torch.float64, device="cuda:0")]
acc = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
for lanes in (1, 2, 4, 8):
    lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, lanes)
    for _ in range(2):
        lib.accumulate_device(p, *d, *acc)
        torch.cuda.synchronize()
    st = lib.stats()
    print(f"lanes {lanes}: pairs kernel {st['pair_kernel_ms']:.3f} ms, binning {st['prepare_ms']:.3f} ms, "
          f"{st['pairs_accepted']:.4g} accepted pairs, {st['pairs_accepted'] / st['pair_kernel_ms'] * 1e-6:.1f} G/s in the kernel")

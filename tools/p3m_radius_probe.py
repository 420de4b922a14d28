#!/usr/bin/env python3
"""Pooled P3M kernel (k_p3m_pool) on an evolved snapshot for search grids of 2, 3 and 4 cells per
cutoff radius.  usage: tools/p3m_radius_probe.py <snapshot.bin written by nbody_host_tests run>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nbody_amd  # noqa: E402
from nbody_amd import Config, ffi  # noqa: E402

lib = nbody_amd.load(device=0)
lib.set_option(ffi.OPT_TIMING, 1)
raw = np.fromfile(sys.argv[1], dtype=np.float64)
n, side = int(raw[0]), int(raw[1])
pos = [np.ascontiguousarray(raw[6 + c * n: 6 + (c + 1) * n]) for c in range(3)]
config = Config(MESH_SIZE=side, N_PER_SIDE=side, USE_PM=True)
p = config.pp_params()
d = [torch.as_tensor(a, device="cuda:0") for a in pos] + [
    torch.full((n,), config.DM_PARTICLE_MASS, dtype=torch.float64, device="cuda:0")]
acc = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
lib.set_option(ffi.OPT_P3M_PAIR_ONCE, -1)
for radius in (2, 3, 4):
    lib.set_option(ffi.OPT_P3M_SEARCH_RADIUS, radius)
    best = None
    for _ in range(4):
        lib.accumulate_device(p, *d, *acc)
        torch.cuda.synchronize()
        st = lib.stats()
        if best is None or st["pair_kernel_ms"] < best["pair_kernel_ms"]:
            best = st
    print(f"a = {raw[2]:.3f}  {radius} cells per cutoff radius: binning {best['prepare_ms']:.3f} ms, kernel "
          f"{best['pair_kernel_ms']:.3f} ms, pooled {best['p3m_pooled']}, occupancy {best['cell_occupancy']:.1f}, "
          f"{best['pairs_accepted'] / best['pair_kernel_ms'] * 1e-6:.1f} G accepted pairs/s", flush=True)

#!/usr/bin/env python3
"""Pair-once direct kernel on small sets: particles per lane x slices of a group's visit list.
usage: tools/direct_small_probe.py [n_side ...]"""
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

lib = nbody_amd.load(device=0)
lib.set_option(ffi.OPT_TIMING, 1)
for n_side in [int(a) for a in sys.argv[1:]] or [32, 48, 64]:
    n = n_side ** 3
    config = Config(MESH_SIZE=n_side - 1 if n_side % 2 == 0 else n_side, N_PER_SIDE=n_side, USE_PM=False)
    p = config.pp_params()
    d = [torch.as_tensor(a, device="cuda:0") for a in datasets.jittered_lattice(n_side)] + [
        torch.full((n,), config.DM_PARTICLE_MASS, dtype=torch.float64, device="cuda:0")]
    acc = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    for per_lane in (0, 1, 2, 4):
        for split in (0, 4, 8, 16, 32, 64, 128):
            lib.set_option(ffi.OPT_TARGETS_PER_LANE, per_lane)
            lib.set_option(ffi.OPT_SOURCE_SPLIT, split)
            lib.set_option(ffi.OPT_FORCE_KERNEL, ffi.KERNEL_DIRECT_SYMMETRIC)
            best = 1e9
            for _ in range(5):
                lib.accumulate_device(p, *d, *acc)
                torch.cuda.synchronize()
                best = min(best, lib.stats()["pair_kernel_ms"])
            print(f"{n_side}^3 per lane {per_lane} slices {split:3d}: {best:.4f} ms, {n * (n - 1) / best * 1e-9:.2f} T/s", flush=True)
    for k in (ffi.OPT_TARGETS_PER_LANE, ffi.OPT_SOURCE_SPLIT, ffi.OPT_FORCE_KERNEL):
        lib.set_option(k, 0)

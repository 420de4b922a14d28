#!/usr/bin/env python3
"""k_p3m_rows (4, 8, 16 lanes of a target walking its rows together) against the automatic choice
without it (-1) and the pooled one-lane-per-target kernel (-2) on snapshots written by
`nbody_host_tests run <ini> <cycles> <prefix>`.  usage: tools/p3m_rows_probe.py <prefix_0001.bin> ..."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import nbody_amd
from nbody_amd import Config, ffi
lib = nbody_amd.load(device=0)
lib.set_option(ffi.OPT_TIMING, 1)
for path in sys.argv[1:]:
    raw = np.fromfile(path, dtype=np.float64)
    n, side = int(raw[0]), int(raw[1])
    pos = [np.ascontiguousarray(raw[6 + c * n: 6 + (c + 1) * n]) for c in range(3)]
    config = Config(MESH_SIZE=side, N_PER_SIDE=side, USE_PM=True)
    p = config.pp_params()
    d = [torch.as_tensor(a, device="cuda:0") for a in pos] + [torch.full((n,), config.DM_PARTICLE_MASS, dtype=torch.float64, device="cuda:0")]
    ref = None
    for rows in (-1, -2, 2, 4, 8):      # -2: one lane per target (the pooled kernel)
        lib.set_option(ffi.OPT_P3M_ROW_LANES, max(rows, -1))
        lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, 1 if rows == -2 else 0)
        best = None
        for _ in range(4):
            acc = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
            lib.accumulate_device(p, *d, *acc)
            torch.cuda.synchronize()
            st = lib.stats()
            if best is None or st["pair_kernel_ms"] < best["pair_kernel_ms"]:
                best = st
        a = torch.stack(acc).cpu().numpy()
        if ref is None:
            ref = (a, best["pairs_accepted"])
        err = np.max(np.abs(a - ref[0])) / np.max(np.abs(ref[0]))
        print(f"{side}^3 a={raw[2]:.3f} row lanes {rows:3d}: kind {best['p3m_pooled']} binning {best['prepare_ms']:.3f} kernel {best['pair_kernel_ms']:.3f} ms "
              f"occupancy {best['cell_occupancy']:.1f} {best['pairs_accepted']/best['pair_kernel_ms']*1e-6:.1f} G/s  same count {best['pairs_accepted']==ref[1]} max diff {err:.1e}", flush=True)
    lib.set_option(ffi.OPT_P3M_ROW_LANES, 0)
    lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, 0)

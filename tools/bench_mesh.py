#!/usr/bin/env python3
"""Times the mesh (PM) force of libnbpm with inputs resident in HBM: per-phase device time and
the HBM-roofline fraction of the whole call.

usage: tools/bench_mesh.py [n_side] [mesh_size] [repeats]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd  # noqa: E402
import datasets  # noqa: E402

HBM_PEAK_GBS = 8000.0


def algorithmic_bytes(n, M):
    """Bytes every phase must move at least once (fp64): particles in/out, meshes read and
    written once per pass that cannot be fused away."""
    cells = M ** 3
    deposit = n * 32 + 8 * cells + n * 8 * 8          # x,y,z,m in; mesh zeroed; 8 corner updates
    solve = 8 * cells * (2 + 2 + 2 + 4)               # r2c, Green's function, c2r, gradient (1 in, 3 out)
    interpolate = n * 24 + 24 * cells + n * 24        # positions in, three meshes in, acc out
    return deposit, solve, interpolate


def main():
    n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    M = int(sys.argv[2]) if len(sys.argv) > 2 else n_side
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    n = n_side ** 3
    config = nbody_amd.Config(MESH_SIZE=M, N_PER_SIDE=n_side)
    pos = [torch.as_tensor(a, device="cuda:0") for a in datasets.jittered_lattice(n_side)]
    m = torch.full((n,), config.DM_PARTICLE_MASS, dtype=torch.float64, device="cuda:0")
    acc = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    mesh = nbody_amd.load_mesh(device=0)
    mesh.set_timing(True)
    p = config.mesh_params()
    mesh.accelerations_device(p, *pos, m, *acc)       # plans, allocations
    torch.cuda.synchronize()
    phases = {"deposit_ms": [], "solve_ms": [], "interpolate_ms": [], "total_ms": []}
    for _ in range(reps):
        mesh.accelerations_device(p, *pos, m, *acc)
        st = mesh.stats()
        for k in phases:
            phases[k].append(st[k])
    med = {k: float(np.median(v)) for k, v in phases.items()}
    b = algorithmic_bytes(n, M)
    out = {"workload": f"mesh force, {n_side}^3 particles on a {M}^3 mesh", **med,
           "algorithmic_bytes": {"deposit": b[0], "solve": b[1], "interpolate": b[2]},
           "achieved_GBs": {k: bb / (med[k + "_ms"] * 1e-3) / 1e9 for k, bb in
                            zip(("deposit", "solve", "interpolate"), b)},
           "hbm_frac_total": sum(b) / (med["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "Mparticles_per_s": n / (med["total_ms"] * 1e-3) / 1e6}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""One-off validation of the direct kernels at a large particle count (default 256^3, the size of
the 8-GPU configuration): full evaluation on one GPU, a sample of targets against the CPU oracle,
and total-momentum conservation.  Too long for the regular test suite (~100 s of GPU at 256^3).

usage: python tests/large_validation.py [n_side] [n_sample]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n_sample = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    import torch
    import nbody_amd
    from nbody_amd import Config, ffi
    from oracle.oracle import Oracle
    import datasets
    n = n_side ** 3
    pos = datasets.jittered_lattice(n_side)
    config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=False)
    m = np.full(n, config.DM_PARTICLE_MASS)
    p = config.pp_params()
    lib = nbody_amd.load(device=0)
    lib.set_option(ffi.OPT_TIMING, 1)
    x, y, z, mm = [torch.from_numpy(a).cuda() for a in pos + [m]]
    acc = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(3)]
    t0 = time.time()
    lib.accumulate_device(p, x, y, z, mm, *acc)
    torch.cuda.synchronize()
    st = lib.stats()
    print(f"n = {n}: kernel variant {st['kernel_variant']}, {st['pair_kernel_ms']:.1f} ms, "
          f"{n * (n - 1) / st['pair_kernel_ms'] / 1e6:.1f} Ginter/s (wall {time.time() - t0:.1f} s)", flush=True)
    got = [a.cpu().numpy() for a in acc]
    orc = Oracle(fast=True)
    rng = np.random.default_rng(1)
    sample = np.sort(rng.choice(n, n_sample, replace=False))
    # three sums of the same terms: the GPU's, the reference-order double sum (what the
    # reference's loop computes) and an independent long-double sum
    strict = Oracle()
    worst = {"gpu-vs-double": 0.0, "gpu-vs-longdouble": 0.0, "double-vs-longdouble": 0.0}
    for i in sample:
        w = strict.accumulate(p, *pos, m, tgt_begin=int(i), n_tgt=1)
        wl = strict.accumulate(p, *pos, m, tgt_begin=int(i), n_tgt=1, long_double=True)
        ref = np.array([w[0][0], w[1][0], w[2][0]])
        refl = np.array([wl[0][0], wl[1][0], wl[2][0]])
        g = np.array([got[0][i], got[1][i], got[2][i]])
        nrm = np.linalg.norm(refl)
        worst["gpu-vs-double"] = max(worst["gpu-vs-double"], float(np.max(np.abs(g - ref)) / nrm))
        worst["gpu-vs-longdouble"] = max(worst["gpu-vs-longdouble"], float(np.max(np.abs(g - refl)) / nrm))
        worst["double-vs-longdouble"] = max(worst["double-vs-longdouble"], float(np.max(np.abs(ref - refl)) / nrm))
    print(f"max normalised differences over {n_sample} sampled targets:", {k: "%.3e" % v for k, v in worst.items()})
    worst = worst["gpu-vs-longdouble"]
    typical = float(np.mean(np.sqrt(got[0] ** 2 + got[1] ** 2 + got[2] ** 2))) * m[0] * n
    mom = [abs(float(np.sum(m * g))) / typical for g in got]
    print("net momentum change / sum |m a|:", ["%.2e" % v for v in mom])
    assert worst < 1e-10 and max(mom) < 1e-9
    print("OK")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""P3M cell kernel, survivors pooled per wavefront (k_p3m_pool) against every lane draining its own
ring (k_p3m_cells), one lane per target, every pair once (k_p3m_sym) and the library's own choice, on three particle sets: the 256^3 jittered lattice, a set
of tight blobs, and (--snapshot) an evolved state written by `nbody_host_tests run ... <prefix>`.

usage: tools/p3m_pool_probe.py [--snapshot file.bin] [--json out.json]
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
from nbody_amd import Config, ffi  # noqa: E402
import datasets  # noqa: E402

lib = nbody_amd.load(device=0)
lib.set_option(ffi.OPT_TIMING, 1)
sets = [("lattice 256^3", 256, lambda: datasets.jittered_lattice(256)),
        ("400 blobs 128^3", 128, lambda: datasets.clustered(128 ** 3, n_blobs=400, sigma=0.004, seed=3))]
if "--snapshot" in sys.argv:
    raw = np.fromfile(sys.argv[sys.argv.index("--snapshot") + 1], dtype=np.float64)
    n0, side0 = int(raw[0]), int(raw[1])
    sets.append((f"evolved a={raw[2]:.3f} {side0}^3", side0,
                 lambda: [np.ascontiguousarray(raw[6 + c * n0: 6 + (c + 1) * n0]) for c in range(3)]))
rows = []
for name, n_side, make in sets:
    pos = make()
    n = pos[0].size
    config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True)
    p = config.pp_params()
    d = [torch.as_tensor(a, device="cuda:0") for a in pos] + [
        torch.full((n,), config.DM_PARTICLE_MASS, dtype=torch.float64, device="cuda:0")]
    acc = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    for pooled in (-1, 1, 2, 0):     # 0: the library's own choice
        lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, 1 if pooled else 0)
        lib.set_option(ffi.OPT_P3M_POOLED_DRAIN, 0 if pooled == 0 else (1 if pooled == 1 else -1))
        lib.set_option(ffi.OPT_P3M_PAIR_ONCE, 0 if pooled == 0 else (1 if pooled == 2 else -1))
        best = None
        for _ in range(4):
            lib.accumulate_device(p, *d, *acc)
            torch.cuda.synchronize()
            st = lib.stats()
            if best is None or st["pair_kernel_ms"] < best["pair_kernel_ms"]:
                best = st
        if pooled == 2 and best["p3m_pooled"] != 2:
            print(f"{name:28s} pair once: does not fit (fell back)", flush=True)
            continue
        kinds = ("k_p3m_cells", "k_p3m_pool", "k_p3m_sym", "k_p3m_rows", "k_p3m_rows_pool")
        rows.append({"set": name, "kernel": kinds[best["p3m_pooled"]], "chosen_by": "library" if pooled == 0 else "option",
                     "kernel_ms": best["pair_kernel_ms"], "binning_ms": best["prepare_ms"],
                     "pairs_accepted": best["pairs_accepted"], "cell_occupancy": best["cell_occupancy"],
                     "G_pairs_per_s": best["pairs_accepted"] / best["pair_kernel_ms"] * 1e-6})
        label = {-1: "per lane ", 1: "pooled   ", 2: "pair once", 0: "library's choice (" + kinds[best["p3m_pooled"]] + ")"}[pooled]
        print(f"{name:28s} {label}: binning {best['prepare_ms']:6.3f} ms, kernel {best['pair_kernel_ms']:8.3f} ms, "
              f"{best['pairs_accepted']:.4g} accepted pairs, {rows[-1]['G_pairs_per_s']:.1f} G/s, "
              f"cell occupancy {best['cell_occupancy']:.2f}", flush=True)
    lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, 0)
    lib.set_option(ffi.OPT_P3M_POOLED_DRAIN, 0)
    lib.set_option(ffi.OPT_P3M_PAIR_ONCE, 0)
if "--json" in sys.argv:
    json.dump(rows, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)

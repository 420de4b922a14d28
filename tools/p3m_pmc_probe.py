#!/usr/bin/env python3
"""Three P3M evaluations, kernel chosen by the library, on a snapshot written by
`nbody_host_tests run <ini> <cycles> <prefix>`: the program rocprofv3 counter passes are taken of.
usage: rocprofv3 ... -- python3 tools/p3m_pmc_probe.py <prefix_0001.bin> [row lanes] [rows pooled]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import nbody_amd
from nbody_amd import Config, ffi
lib = nbody_amd.load(device=0)
lib.set_option(ffi.OPT_TIMING, 1)
if len(sys.argv) > 2:
    lib.set_option(ffi.OPT_P3M_ROW_LANES, int(sys.argv[2]))
if len(sys.argv) > 3:
    lib.set_option(ffi.OPT_P3M_ROWS_POOLED, int(sys.argv[3]))
raw = np.fromfile(sys.argv[1], dtype=np.float64)
n, side = int(raw[0]), int(raw[1])
pos = [np.ascontiguousarray(raw[6 + c * n: 6 + (c + 1) * n]) for c in range(3)]
config = Config(MESH_SIZE=side, N_PER_SIDE=side, USE_PM=True)
d = [torch.as_tensor(a, device="cuda:0") for a in pos] + [torch.full((n,), config.DM_PARTICLE_MASS, dtype=torch.float64, device="cuda:0")]
best = None
for _ in range(int(os.environ.get("P3M_PROBE_REPS", "3"))):
    acc = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    lib.accumulate_device(config.pp_params(), *d, *acc)
    torch.cuda.synchronize()
    st = lib.stats()
    best = st if best is None or st["pair_kernel_ms"] < best["pair_kernel_ms"] else best
st = best
print(f"{side}^3 a={raw[2]:.3f} kind {st['p3m_pooled']} kernel {st['pair_kernel_ms']:.3f} ms occupancy {st['cell_occupancy']:.1f} accepted {st['pairs_accepted']} checksum {float(sum(torch.sum(torch.abs(c)) for c in acc)):.12e}")

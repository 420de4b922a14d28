#!/usr/bin/env python3
"""Times the P3M short-range call on the 256^3 jittered lattice (BASELINE config 3) for each library
build named in NBPP_LIBS (file names under nbody_amd/lib, default libnbpp.so), one child process per
build: pair kernel and binning time (best of 6 and mean), accepted pairs, and a checksum of the sums.

usage: [NBPP_LIBS="a.so b.so"] tools/p3m_sym_probe.py [--n-side 256] [--pair-once 1|0|-1] [--radius R] [--shuffle]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch
    import nbody_amd
    from nbody_amd import Config, ffi
    import datasets
    n_side = int(os.environ.get("PROBE_N_SIDE", "256"))
    lib = nbody_amd.load(device=0)
    lib.set_option(ffi.OPT_TIMING, 1)
    lib.set_option(ffi.OPT_P3M_PAIR_ONCE, int(os.environ.get("PROBE_PAIR_ONCE", "0")))
    lib.set_option(ffi.OPT_P3M_SEARCH_RADIUS, int(os.environ.get("PROBE_RADIUS", "0")))
    config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True)
    p = config.pp_params()
    n = n_side ** 3
    pos = datasets.jittered_lattice(n_side)
    if os.environ.get("PROBE_SHUFFLE"):      # the same particles in random memory order
        order = np.random.default_rng(7).permutation(n)
        pos = [np.ascontiguousarray(a[order]) for a in pos]
    d = [torch.as_tensor(a, device="cuda:0") for a in pos] + [
        torch.full((n,), config.DM_PARTICLE_MASS, dtype=torch.float64, device="cuda:0")]
    acc = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    ks, ps = [], []
    for it in range(8):
        for a in acc:
            a.zero_()
        lib.accumulate_device(p, *d, *acc)
        torch.cuda.synchronize()
        st = lib.stats()
        if it >= 2:
            ks.append(st["pair_kernel_ms"])
            ps.append(st["prepare_ms"])
    check = float(sum(a.abs().sum() for a in acc))
    print(json.dumps({"lib": os.path.basename(os.environ.get("NBPP_LIB", "libnbpp.so")), "kernel": st["p3m_pooled"],
                      "kernel_ms_min": min(ks), "kernel_ms_mean": float(np.mean(ks)), "binning_ms_mean": float(np.mean(ps)),
                      "pairs_accepted": st["pairs_accepted"], "sum_abs_acc": check}))


if __name__ == "__main__":
    if os.environ.get("PROBE_CHILD"):
        child()
        sys.exit(0)
    env = dict(os.environ, PROBE_CHILD="1")
    if "--n-side" in sys.argv:
        env["PROBE_N_SIDE"] = sys.argv[sys.argv.index("--n-side") + 1]
    if "--radius" in sys.argv:
        env["PROBE_RADIUS"] = sys.argv[sys.argv.index("--radius") + 1]
    if "--shuffle" in sys.argv:
        env["PROBE_SHUFFLE"] = "1"
    if "--pair-once" in sys.argv:
        env["PROBE_PAIR_ONCE"] = sys.argv[sys.argv.index("--pair-once") + 1]
    for so in os.environ.get("NBPP_LIBS", "libnbpp.so").split():
        env["NBPP_LIB"] = os.path.join(ROOT, "nbody_amd", "lib", so)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=300)
        print((r.stdout.strip() or r.stderr[-600:]) + "\n" + "\n".join(l for l in r.stderr.splitlines() if "stats" in l), flush=True)

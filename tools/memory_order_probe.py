#!/usr/bin/env python3
"""How much the MEMORY ORDER of the caller's particle arrays costs the P3M short-range call: the
C++ engine evolves tools/configs/p3m_hydro_128.ini for some cycles and writes its particles; the
call is then timed on that state as the engine holds it (lattice enumeration order, carried along
since the start), on the same particles sorted by mesh cell (z fastest, the order of the search
grid), and on a random permutation of them.  Prints binning and pair-kernel time per order.

usage: tools/memory_order_probe.py [cycles, default 12000] [more cycle counts ...]
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import nbody_amd
    from nbody_amd import Config, ffi
    lib = nbody_amd.load(device=0)
    lib.set_option(ffi.OPT_TIMING, 1)
    exe = os.path.join(ROOT, "nbody_amd", "host", "nbody_host_tests")
    ini = os.path.join(ROOT, "tools", "configs", "p3m_hydro_128.ini")
    for cycles in [int(a) for a in sys.argv[1:]] or [12000]:
        with tempfile.TemporaryDirectory() as tmp:
            text = open(ini).read().replace("debug_info_every_cycles = 2000", "debug_info_every_cycles = 0")
            text += "save_hdf5_every_delta_a = 10\n"
            open(os.path.join(tmp, "e.ini"), "w").write(text)
            r = subprocess.run([exe, "run", os.path.join(tmp, "e.ini"), str(cycles), os.path.join(tmp, "state")],
                               capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stdout + r.stderr
            raw = np.fromfile(os.path.join(tmp, "state_0001.bin" if cycles else "state_0000.bin"), dtype=np.float64)
        n, n_side, a_now = int(raw[0]), int(raw[1]), float(raw[2])
        config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True)
        p = config.pp_params()
        cols = [np.ascontiguousarray(raw[6 + c * n: 6 + (c + 1) * n]) for c in range(4)]
        cell = [np.minimum((np.mod(c, config.DOMAIN_SIZE) / config.DOMAIN_SIZE * n_side).astype(np.int64), n_side - 1)
                for c in cols[:3]]
        orders = {"engine": np.arange(n),
                  "sorted by mesh cell": np.argsort((cell[0] * n_side + cell[1]) * n_side + cell[2], kind="stable"),
                  "random": np.random.default_rng(3).permutation(n)}
        for name, order in orders.items():
            d = [torch.from_numpy(np.ascontiguousarray(c[order])).to("cuda:0") for c in cols]
            acc = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
            ks, ps = [], []
            for it in range(7):
                lib.accumulate_device(p, *d, *acc)
                torch.cuda.synchronize()
                st = lib.stats()
                if it >= 2:
                    ks.append(st["pair_kernel_ms"])
                    ps.append(st["prepare_ms"])
            print(json.dumps({"cycles": cycles, "a": round(a_now, 3), "order": name, "kernel": st["p3m_pooled"],
                              "occupancy_measure": round(st["cell_occupancy"], 2), "binning_ms": round(float(np.mean(ps)), 3),
                              "kernel_ms": round(float(np.mean(ks)), 3), "pairs_accepted": st["pairs_accepted"]}), flush=True)


if __name__ == "__main__":
    main()

"""What the host-pointer entry point nbpp_accumulate costs, bus included, against the same
evaluation on device-resident arrays: direct summation at 128^3 and P3M short range at 256^3,
from pageable caller arrays and from page-locked ones (nbpp_host_alloc).

    python tools/measure_pcie.py [--json out.json]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nbody_amd  # noqa: E402
import datasets  # noqa: E402
from nbody_amd import Config, ffi  # noqa: E402


def main():
    lib = nbody_amd.load(device=0)
    lib.set_option(ffi.OPT_TIMING, 1)
    rows = []
    for mode, n_side in (("direct", 128), ("p3m", 256)):
        n = n_side ** 3
        cfg = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=(mode == "p3m"))
        pos = datasets.jittered_lattice(n_side)
        m = np.full(n, cfg.DM_PARTICLE_MASS)
        acc = [np.zeros(n) for _ in range(3)]
        p = cfg.pp_params()
        for label in ("pageable", "page-locked (nbpp_host_alloc)"):
            if label != "pageable":
                locked = [lib.host_alloc(n) for _ in range(7)]
                for dst, src in zip(locked, (*pos, m, *acc)):
                    dst[:] = src
                pos, m, acc = locked[:3], locked[3], locked[4:]
            lib.accumulate(p, *pos, m, *acc)          # first call: buffers, page-locked staging
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                lib.accumulate(p, *pos, m, *acc)
                dt = (time.perf_counter() - t0) * 1e3
                st = lib.stats()
                if best is None or dt < best[0]:
                    best = (dt, st["total_ms"], st["h2d_ms"], st["d2h_ms"])
            rows.append({"mode": mode, "n_side": n_side, "caller_memory": label, "host_call_ms": best[0],
                         "device_ms": best[1], "h2d_ms": best[2], "d2h_and_add_ms": best[3]})
            print("%-6s %3d^3  %-34s host call %8.2f ms; device %8.2f, up %6.2f, down + add %6.2f ms"
                  % (mode, n_side, label, *best), flush=True)
            if label != "pageable":
                for a in locked:
                    lib.host_free(a)
    if "--json" in sys.argv:
        json.dump(rows, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()

import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import nbody_amd, datasets
from nbody_amd import Config, ffi
lib = nbody_amd.load(device=0); lib.set_option(ffi.OPT_TIMING, 1)
for mode, n_side in (("direct", 128), ("p3m", 256)):
    n = n_side**3
    cfg = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=(mode=="p3m"))
    pos = datasets.jittered_lattice(n_side); m = np.full(n, cfg.DM_PARTICLE_MASS)
    acc = [np.zeros(n) for _ in range(3)]
    p = cfg.pp_params()
    lib.accumulate(p, *pos, m, *acc)
    t0 = time.perf_counter(); lib.accumulate(p, *pos, m, *acc); dt = time.perf_counter() - t0
    st = lib.stats()
    print(mode, n_side, "host call %.2f ms; device total %.2f, h2d %.2f, d2h %.2f ms" % (dt*1e3, st["total_ms"], st["h2d_ms"], st["d2h_ms"]))

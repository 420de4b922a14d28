#!/usr/bin/env python3
"""Headline benchmark: direct-summation PP gravity, fp64, N = 128^3, on N GPUs of one node.

One step = one force evaluation of the whole particle set: every rank all-gathers the
positions (RCCL; a no-op on one GPU), then runs the pair kernel for its own slice of targets
against all sources.  Inputs are resident in HBM before the timed region.  Rank 0 prints one
JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n-side 128] [--mode direct|p3m]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CU x 128 flop/clk x 2.4 GHz
FLOPS_PER_PAIR_DIRECT = 25       # periodic, softened direct PP (FMA = 2, rsqrt = 1)
FLOPS_PER_PAIR_P3M = 40          # per accepted P3M pair
# what nbpp_stats.p3m_pooled says of the most recent P3M call
P3M_KERNEL_NAMES = {0: "k_p3m_cells (every ordered pair, one lane per target)",
                    1: "k_p3m_pool (survivors pooled per wavefront)",
                    2: "k_p3m_sym (every unordered pair once, reactions gathered per brick in LDS)",
                    3: "k_p3m_rows (the lanes of a target walk its rows together)",
                    4: "k_p3m_rows_pool (row walk, survivors pooled per wavefront)"}


def cpu_baseline(p, pos, m, mode, budget_s=15.0):
    """The CPU port (oracle/pp_oracle.c, -O3 AVX2+FMA build, OpenMP on every host core) on a
    bounded sample of targets of the same workload."""
    from oracle.oracle import Oracle, EXACT_CELLS, EXACT
    orc = Oracle(fast=True)
    n = m.size
    # every thread OpenMP grants this process: omp_get_max_threads(), i.e. OMP_NUM_THREADS where
    # the box sets it (its CPU share for one GPU), else every core the process may run on
    cores = orc.max_threads()
    if mode == "direct":
        n_tgt = 64 * cores
        while True:
            t0 = time.perf_counter()
            orc.accumulate(p, *pos, m, tgt_begin=0, n_tgt=n_tgt, traversal=EXACT, nthreads=cores)
            dt = time.perf_counter() - t0
            if dt > 0.25 * budget_s or n_tgt * 4 > n:
                break
            n_tgt = min(n, int(n_tgt * max(2.0, 0.6 * budget_s / max(dt, 1e-3))))
        pairs = n_tgt * (n - 1)
        sample = f"{n_tgt} targets x all {n} sources, {dt:.2f} s"
    else:
        n_tgt = min(n, 8192 * cores)
        while True:
            t0 = time.perf_counter()
            orc.accumulate(p, *pos, m, tgt_begin=0, n_tgt=n_tgt, traversal=EXACT_CELLS, nthreads=cores)
            dt = time.perf_counter() - t0
            if dt > 0.25 * budget_s or n_tgt == n:
                break
            n_tgt = min(n, int(n_tgt * max(2.0, 0.6 * budget_s / max(dt, 1e-3))))
        pairs = orc.count_within_cutoff(p, *pos, tgt_begin=0, n_tgt=n_tgt, nthreads=cores)
        sample = f"{n_tgt} targets, cell search over the reference's mesh cells, {dt:.2f} s"
    out = {"value": pairs / dt / 1e9, "unit": "Ginter/s", "cores": cores, "kind": "port",
           "sample": sample, "build": "fast" if orc.fast else "strict",
           "host_cores_visible": os.cpu_count(), "host_cores_usable": len(os.sched_getaffinity(0)),
           "omp_num_threads_env": os.environ.get("OMP_NUM_THREADS")}
    if mode == "direct":
        out["config1_kdk_step_32cubed"] = cpu_config1_kdk_step(orc, cores)
    return out


def cpu_config1_kdk_step(orc, cores):
    """BASELINE config 1 as written: PP direct summation, 32^3 particles, one whole KDK step
    (kick, drift, full force evaluation, kick: src/integrator.cpp:215-275 without mesh and gas) of
    the CPU port on the host cores, every target against every source."""
    from nbody_amd import Config
    import datasets
    n_side = 32
    n = n_side ** 3
    config = Config(MESH_SIZE=31, N_PER_SIDE=n_side, USE_PM=False)   # odd mesh: the exact regime
    p = config.pp_params()
    pos = [np.ascontiguousarray(c).copy() for c in datasets.zeldovich(n_side)]   # seed 42, as shipped
    m = np.full(n, config.DM_PARTICLE_MASS)
    vel = [np.zeros(n) for _ in range(3)]
    acc = list(orc.accumulate(p, *pos, m, nthreads=cores))            # initial accelerations
    dt = config.FIXED_DT
    t0 = time.perf_counter()
    orc.kick(vel, acc, dt / 2.0, 1.0, 0.0)
    orc.drift(pos, vel, dt, config.DOMAIN_SIZE)
    t1 = time.perf_counter()
    acc = list(orc.accumulate(p, *pos, m, nthreads=cores))
    t2 = time.perf_counter()
    orc.kick(vel, acc, dt / 2.0, 1.0, 0.0)
    t3 = time.perf_counter()
    pairs = n * (n - 1)
    return {"seconds": t3 - t0, "force_seconds": t2 - t1, "Ginter_per_s": pairs / (t2 - t1) / 1e9,
            "pairs": pairs, "cores": cores}


def extra_p3m256(lib, ffi, torch, dev, stream, steps=6, warmup=2):
    """BASELINE config 3 under the same clock: P3M short range (GPU cell list) at 256^3."""
    from nbody_amd import Config
    import datasets
    n_side = 256
    n = n_side ** 3
    config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True)
    p = config.pp_params()
    pos = datasets.jittered_lattice(n_side)
    x, y, z = [torch.from_numpy(c).to(dev) for c in pos]
    del pos
    mass = torch.full((n,), config.DM_PARTICLE_MASS, dtype=torch.float64, device=dev)
    acc = [torch.zeros(n, dtype=torch.float64, device=dev) for _ in range(3)]
    for k in (ffi.OPT_TARGETS_PER_LANE, ffi.OPT_FORCE_KERNEL, ffi.OPT_P3M_SEARCH_RADIUS,
              ffi.OPT_P3M_LANES_PER_TARGET, ffi.OPT_DETERMINISTIC):
        lib.set_option(k, 0)
    for _ in range(warmup):
        lib.accumulate_device(p, x, y, z, mass, *acc, stream=stream)
    torch.cuda.synchronize()
    kernel_ms, prep_ms = [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        lib.accumulate_device(p, x, y, z, mass, *acc, stream=stream)
        st = lib.stats()
        kernel_ms.append(st["pair_kernel_ms"])
        prep_ms.append(st["prepare_ms"])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    accepted = float(st["pairs_accepted"])
    k_ms = float(np.mean(kernel_ms))
    achieved = accepted * FLOPS_PER_PAIR_P3M / (k_ms * 1e-3) / 1e12
    return {"workload": "P3M short range, 256^3 particles, mesh 256, cutoff 2.5 cells (BASELINE config 3)",
            "value": accepted * steps / elapsed / 1e9, "unit": "G accepted pairs/s",
            "ms_per_step": elapsed / steps * 1e3, "steps": steps, "warmup": warmup,
            "pairs_accepted": accepted,
            "kernel": P3M_KERNEL_NAMES[st["p3m_pooled"]],
            "kernel_ms": k_ms,
            "prepare_ms": float(np.mean(prep_ms)),
            "roofline": {"bound": "valu_fp64", "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP64_VECTOR_PEAK_TFLOPS,
                         "flops_per_accepted_pair": FLOPS_PER_PAIR_P3M}}


def extra_kdk32(lib, ffi, torch, dev, stream, steps=200, warmup=20):
    """BASELINE config 1 on the GPU: PP direct summation, 32^3 particles from the Zel'dovich set of
    the CPU leg (cpu_config1_kdk_step), whole KDK steps through the device entry points - kick and
    drift, zeroed accelerations, full force evaluation, max |a|^2 read back (the caller contract of
    src/integrator.cpp:236-269), closing kick."""
    from nbody_amd import Config
    import datasets
    n_side = 32
    n = n_side ** 3
    config = Config(MESH_SIZE=31, N_PER_SIDE=n_side, USE_PM=False)
    p = config.pp_params()
    x = [torch.from_numpy(np.ascontiguousarray(c).copy()).to(dev) for c in datasets.zeldovich(n_side)]
    v = [torch.zeros(n, dtype=torch.float64, device=dev) for _ in range(3)]
    a = [torch.zeros(n, dtype=torch.float64, device=dev) for _ in range(3)]
    mass = torch.full((n,), config.DM_PARTICLE_MASS, dtype=torch.float64, device=dev)
    for k in (ffi.OPT_TARGETS_PER_LANE, ffi.OPT_FORCE_KERNEL, ffi.OPT_DETERMINISTIC):
        lib.set_option(k, 0)
    dt = config.FIXED_DT
    lib.accumulate_device(p, *x, mass, *a, stream=stream)

    def kdk():
        lib.kick_drift_device(x, v, a, dt / 2.0, dt, 1.0, 0.0, config.DOMAIN_SIZE, stream=stream)
        for c in a:
            c.zero_()
        lib.accumulate_device(p, *x, mass, *a, stream=stream)
        lib.max_accel_sq_device(a, stream=stream)
        lib.kick_device(v, a, dt / 2.0, 1.0, 0.0, stream=stream)

    for _ in range(warmup):
        kdk()
    torch.cuda.synchronize()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(steps):
        kdk()
        kernel_ms.append(lib.stats()["pair_kernel_ms"])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    pairs = float(n) * float(n - 1)
    k_ms = float(np.mean(kernel_ms))
    return {"workload": "PP direct summation, 32^3 particles (Zel'dovich), whole KDK steps (BASELINE config 1 on the GPU)",
            "ms_per_step": elapsed / steps * 1e3, "steps": steps, "warmup": warmup,
            "value": pairs * steps / elapsed / 1e9, "unit": "Ginter/s", "pairs_per_step": pairs,
            "kernel_ms": k_ms,
            "roofline": {"bound": "valu_fp64", "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "achieved": pairs * FLOPS_PER_PAIR_DIRECT / (k_ms * 1e-3) / 1e12,
                         "frac": pairs * FLOPS_PER_PAIR_DIRECT / (k_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                         "flops_per_interaction": FLOPS_PER_PAIR_DIRECT}}


def extra_step128(cycles=100):
    """BASELINE config 5 under the same clock: the product's own C++ engine (a child process) runs
    tools/configs/p3m_hydro_128.ini for `cycles` KDK cycles, state resident in HBM.  Two runs:
    one untimed inside (wall ms per cycle, force phases overlapped as in production), one with
    every cycle's phases timed by HIP events (the reference's PM / PP / Hydro timer split,
    src/integrator.cpp:228,238,255)."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "nbody_amd", "host", "nbody_host_tests")
    ini = os.path.join(ROOT, "tools", "configs", "p3m_hydro_128.ini")
    if not os.path.exists(exe):
        return {"error": f"{exe} not built"}
    pat = re.compile(r"timing: (\d+) cycles, (\S+) ms/cycle wall; PP (\S+), mesh (\S+), hydro sweeps (\S+) ms/cycle "
                     r"\(device, (\d+) timed cycles\)")
    out = {"workload": "P3M + HLL hydro, 128^3 particles + 128^3 gas mesh, 100 KDK cycles (BASELINE config 5)",
           "cycles": cycles}
    for label, timing_every in (("wall", "0"), ("split", "1")):
        r = subprocess.run([exe, "run", ini, str(cycles), "", "", timing_every], capture_output=True,
                           text=True, timeout=300)
        t = pat.search(r.stdout)
        if r.returncode != 0 or not t:
            return {"error": (r.stdout + r.stderr)[-400:]}
        if label == "wall":
            out["ms_per_cycle"] = float(t[2])
        else:
            out["ms_per_cycle_with_phase_timers"] = float(t[2])
            out["device_ms_per_cycle"] = {"pp_short_range": float(t[3]), "mesh_force": float(t[4]),
                                          "hydro_sweeps": float(t[5]), "timed_cycles": int(t[6])}
    gas = re.findall(r"gas mass (\S+)", r.stdout)
    if len(gas) == 2:
        out["gas_mass_drift"] = abs(float(gas[1]) - float(gas[0])) / float(gas[0])
    return out


def extra_p3m128_evolved(lib, ffi, torch, dev, stream, cycles=12000, steps=8, warmup=2):
    """The short-range force on a CLUSTERED state, under the same clock: the product's C++ engine (a
    child process) evolves tools/configs/p3m_hydro_128.ini for `cycles` KDK cycles (12 000: a = 0.51)
    and writes its particles; the short-range call alone is then timed here on that state, with the
    kernel the library picks.  Runs spend most of their cycles clustered; the jittered lattice of
    `p3m256` is the even case."""
    import subprocess
    import tempfile
    from nbody_amd import Config
    exe = os.path.join(ROOT, "nbody_amd", "host", "nbody_host_tests")
    ini = os.path.join(ROOT, "tools", "configs", "p3m_hydro_128.ini")
    if not os.path.exists(exe):
        return {"error": f"{exe} not built"}
    with tempfile.TemporaryDirectory() as tmp:
        text = open(ini).read().replace("debug_info_every_cycles = 2000", "debug_info_every_cycles = 0")
        text += "save_hdf5_every_delta_a = 10\n"          # two snapshots: the start and the last cycle
        evolve_ini = os.path.join(tmp, "evolve.ini")
        open(evolve_ini, "w").write(text)
        t0 = time.perf_counter()
        r = subprocess.run([exe, "run", evolve_ini, str(cycles), os.path.join(tmp, "state")],
                           capture_output=True, text=True, timeout=400)
        evolve_s = time.perf_counter() - t0
        last = os.path.join(tmp, "state_0001.bin")
        if r.returncode != 0 or not os.path.exists(last):
            return {"error": (r.stdout + r.stderr)[-400:]}
        raw = np.fromfile(last, dtype=np.float64)
    n, n_side, a_now = int(raw[0]), int(raw[1]), float(raw[2])
    config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True)
    p = config.pp_params()
    x, y, z = [torch.from_numpy(np.ascontiguousarray(raw[6 + c * n: 6 + (c + 1) * n])).to(dev) for c in range(3)]
    mass = torch.from_numpy(np.ascontiguousarray(raw[6 + 3 * n: 6 + 4 * n])).to(dev)
    acc = [torch.zeros(n, dtype=torch.float64, device=dev) for _ in range(3)]
    for k in (ffi.OPT_TARGETS_PER_LANE, ffi.OPT_FORCE_KERNEL, ffi.OPT_P3M_SEARCH_RADIUS,
              ffi.OPT_P3M_LANES_PER_TARGET, ffi.OPT_DETERMINISTIC):
        lib.set_option(k, 0)
    for _ in range(warmup):
        lib.accumulate_device(p, x, y, z, mass, *acc, stream=stream)
    torch.cuda.synchronize()
    kernel_ms, prep_ms = [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        lib.accumulate_device(p, x, y, z, mass, *acc, stream=stream)
        st = lib.stats()
        kernel_ms.append(st["pair_kernel_ms"])
        prep_ms.append(st["prepare_ms"])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    accepted = float(st["pairs_accepted"])
    k_ms = float(np.mean(kernel_ms))
    achieved = accepted * FLOPS_PER_PAIR_P3M / (k_ms * 1e-3) / 1e12
    return {"workload": f"P3M short range on the state the engine reaches after {cycles} cycles of "
                        f"tools/configs/p3m_hydro_128.ini (128^3 particles, a = {a_now:.3f})",
            "scale_factor": a_now, "evolve_seconds": evolve_s, "cell_occupancy_measure": st["cell_occupancy"],
            "value": accepted * steps / elapsed / 1e9, "unit": "G accepted pairs/s",
            "ms_per_step": elapsed / steps * 1e3, "steps": steps, "warmup": warmup,
            "pairs_accepted": accepted, "kernel": P3M_KERNEL_NAMES[st["p3m_pooled"]],
            "kernel_ms": k_ms, "prepare_ms": float(np.mean(prep_ms)),
            "roofline": {"bound": "valu_fp64", "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP64_VECTOR_PEAK_TFLOPS,
                         "flops_per_accepted_pair": FLOPS_PER_PAIR_P3M}}


def launch_ranks(n_ranks, argv):
    """`python bench.py --gpus N` from a bare shell: start the N ranks as children (one process per
    GPU through torch.distributed.run), pass rank 0's JSON line on and exit with the children's
    worst return code.  This parent never initialises the GPU (no torch, no HIP): a process that
    has may not be replaced, and it has nothing to compute."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = 0
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            sys.stdout.write(line)
            sys.stdout.flush()
            lines += 1
        else:                       # launcher chatter: keep stdout to the one JSON line
            sys.stderr.write(line)
    rc = proc.wait()
    if rc == 0 and lines != 1:
        print(f"bench.py: the ranks printed {lines} result lines, expected one", file=sys.stderr)
        rc = 1
    return rc


def main():
    if os.environ.get("NBODY_BENCH_WATCHDOG"):   # seconds: dump every thread's stack and exit if still running
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["NBODY_BENCH_WATCHDOG"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-side", type=int, default=0,
                    help="particles per side; default 128 (direct) or 256 (p3m), the headline sizes")
    ap.add_argument("--mode", choices=["direct", "p3m"], default="direct")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the P3M 256^3, evolved 128^3 and whole-step figures that ride along in "
                         "the default one-GPU line (`extra`)")
    ap.add_argument("--per-lane", type=int, default=0)
    ap.add_argument("--force-kernel", type=int, default=0)
    ap.add_argument("--p3m-radius", type=int, default=0)
    ap.add_argument("--p3m-lanes", type=int, default=0, help="P3M cell kernel: lanes per target (0 = automatic)")
    ap.add_argument("--p3m-pair-once", type=int, default=0,
                    help="P3M cell kernel: 1 = every unordered pair once whenever it fits, -1 = never (0 = automatic)")
    ap.add_argument("--p3m-pooled", type=int, default=0,
                    help="P3M cell kernel: 1 = survivors pooled per wavefront, -1 = never (0 = automatic)")
    ap.add_argument("--with-mesh", action="store_true",
                    help="p3m mode, one GPU: a step is the whole force phase, mesh force "
                         "(libnbpm) then short range accumulated onto it")
    ap.add_argument("--halo-exchange", type=int, default=0,
                    help="P3M on several GPUs: 1 = send each peer only the particles within a cutoff "
                         "radius of its block, -1 = all-gather every position (0 = the library's default)")
    ap.add_argument("--native-one-rank", action="store_true",
                    help="one GPU: go through libnbmg with a communicator of one rank (exercises the "
                         "multi-GPU code path on a single-GPU box)")
    ap.add_argument("--ordered", action="store_true",
                    help="bitwise-reproducible ordered-pair kernel (NBPP_OPT_DETERMINISTIC)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    import nbody_amd
    from nbody_amd import Config, ffi
    import datasets

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # Rehearsal of the multi-rank flow on a box with ONE GPU (never what the scaling numbers come
    # from): NBODY_BENCH_ONE_DEVICE=1 puts every rank on device 0.  RCCL does not accept two ranks on
    # one device, so libnbmg is asked for its host-staged transport (NBMG_TRANSPORT=host: the same
    # library calls, the collectives carried through shared host memory) and torch.distributed uses gloo.
    rehearsal = world > 1 and os.environ.get("NBODY_BENCH_ONE_DEVICE") == "1"
    if rehearsal:
        local_rank = 0
        os.environ["NBMG_TRANSPORT"] = "host"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    lib = nbody_amd.load(device=local_rank)
    lib.set_option(ffi.OPT_TIMING, 1)
    lib.set_option(ffi.OPT_TARGETS_PER_LANE, args.per_lane)
    lib.set_option(ffi.OPT_FORCE_KERNEL, args.force_kernel)
    lib.set_option(ffi.OPT_P3M_SEARCH_RADIUS, args.p3m_radius)
    lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, args.p3m_lanes)
    lib.set_option(ffi.OPT_DETERMINISTIC, 1 if args.ordered else 0)
    lib.set_option(ffi.OPT_P3M_PAIR_ONCE, args.p3m_pair_once)
    lib.set_option(ffi.OPT_P3M_POOLED_DRAIN, args.p3m_pooled)

    n_side = args.n_side or (256 if args.mode == "p3m" else 128)
    n = n_side ** 3
    use_pm = args.mode == "p3m"
    config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=use_pm)
    p = config.pp_params()
    pos = datasets.jittered_lattice(n_side)           # same seeded set on every rank
    m = np.full(n, config.DM_PARTICLE_MASS)
    stream = torch.cuda.current_stream().cuda_stream

    # Two ways through a step, never mixed and never chosen behind the caller's back:
    #   one GPU       -> nbpp_accumulate_device on the whole set (comm_backend "none");
    #   several ranks -> libnbmg only: its own communicator below the C ABI; torch.distributed carries
    #                    the 128-byte communicator id and the timing barrier.  If the library cannot
    #                    start, the run fails (non-zero exit): there is nothing to fall back to.
    mg = None
    if world == 1 and args.native_one_rank:
        mg = nbody_amd.load_multi_gpu()
        mg.init(0, 1, None, local_rank)
        mg.set_option(ffi.MG_OPT_PAIR_SHARES, -1 if args.ordered else 1)
    elif world > 1:
        mg = nbody_amd.load_multi_gpu()
        ident = torch.zeros(ffi.MG_UNIQUE_ID_BYTES, dtype=torch.uint8, device="cpu" if rehearsal else dev)
        if rank == 0:
            ident.copy_(torch.frombuffer(bytearray(mg.unique_id()), dtype=torch.uint8))
        dist.broadcast(ident, src=0)
        mg.init(rank, world, bytes(ident.cpu().numpy().tobytes()), local_rank)
        mg.set_option(ffi.MG_OPT_PAIR_SHARES, -1 if args.ordered else 0)
    if mg is not None:
        mg.set_option(ffi.MG_OPT_TIMING, 1)
        if args.halo_exchange:
            mg.set_option(ffi.MG_OPT_HALO_EXCHANGE, args.halo_exchange)
        comm_backend = {"rccl": "RCCL via libnbmg (C ABI)",
                        "host": "libnbmg (C ABI) over its host-staged transport - REHEARSAL: every rank "
                                "on one GPU; not a scaling measurement"}[mg.transport()]
        if (mg.transport() == "host") != rehearsal:
            raise SystemExit(f"libnbmg runs on its {mg.transport()} transport, rehearsal={rehearsal}")
        nb_begin, nb_count = mg.partition(n, rank, world)
        nb_own = [torch.from_numpy(c[nb_begin:nb_begin + nb_count].copy()).to(dev) for c in pos]
        nb_mass = torch.zeros(mg.padded_length(n, world), dtype=torch.float64, device=dev)
        nb_mass[:n] = config.DM_PARTICLE_MASS
        nb_acc = [torch.zeros(max(nb_count, 1), dtype=torch.float64, device=dev) for _ in range(3)]
        first_count = mg.partition(n, 0, world)[1]
    else:
        comm_backend = "none (one GPU)"
        x, y, z = [torch.from_numpy(c).to(dev) for c in pos]
        mass = torch.from_numpy(m).to(dev)
        acc = [torch.zeros(n, dtype=torch.float64, device=dev) for _ in range(3)]
        first_count = n
    comm_ms = []
    mesh = None
    if args.with_mesh:
        if not use_pm or world != 1 or mg is not None:
            raise SystemExit("--with-mesh needs --mode p3m on one GPU")
        mesh = nbody_amd.load_mesh(device=local_rank)
        mesh.set_timing(True)
        mesh_params = config.mesh_params()
    mesh_ms = []

    def step():
        if mg is not None:
            # the multi-GPU compute_pp_forces of libnbmg: all-gather of positions (or halo trade), this
            # rank's share of the pairs + reduce-scatter (direct) or this rank's targets (P3M / --ordered)
            for a in nb_acc:
                a.zero_()
            mg.accumulate_device(p, n, *nb_own, nb_mass, *nb_acc, stream=stream)
            ms = mg.stats()
            comm_ms.append((ms["allgather_ms"], ms["reduce_ms"], ms["compute_ms"]))
            return lib.stats()
        if mesh is not None:   # src/integrator.cpp:239-258: the mesh force overwrites, PP adds
            mesh.accelerations_device(mesh_params, x, y, z, mass, *acc, stream=stream)
            mesh_ms.append(mesh.stats()["total_ms"])
        else:                  # PP only: the caller zero-fills first (src/integrator.cpp:246-251)
            for a in acc:
                a.zero_()
        lib.accumulate_device(p, x, y, z, mass, *acc, stream=stream)
        return lib.stats()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    del comm_ms[:]
    fence()
    t0 = time.perf_counter()
    kernel_ms, prep_ms, accepted = [], [], 0
    st = {}
    for _ in range(args.steps):
        st = step()
        kernel_ms.append(st["pair_kernel_ms"])
        prep_ms.append(st["prepare_ms"])
        accepted = st["pairs_accepted"]
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed, float(np.mean(kernel_ms)), float(accepted)], dtype=torch.float64,
                     device="cpu" if rehearsal else dev)
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, kernel_avg_ms, accepted_total = float(tmax[0]), float(tmax[1]), float(tsum[2])
    else:
        elapsed, kernel_avg_ms, accepted_total = float(t[0]), float(t[1]), float(t[2])

    if rank == 0:
        if use_pm:
            pairs_per_step = accepted_total
            flops = FLOPS_PER_PAIR_P3M
            workload = f"P3M short range, {n_side}^3 particles, mesh {n_side}, cutoff 2.5 cells"
            if mesh is not None:
                workload = f"P3M force phase (mesh force + short range), {n_side}^3 particles, " \
                           f"mesh {n_side}, cutoff 2.5 cells"
        else:
            pairs_per_step = float(n) * float(n - 1)
            flops = FLOPS_PER_PAIR_DIRECT
            workload = f"PP direct summation fp64, N={n_side}^3, periodic box"
        value = pairs_per_step * args.steps / elapsed / 1e9
        # dominant kernel on rank 0's share: algorithmic flops of one launch / its duration
        pair_shares = mg is not None and bool(mg.stats()["pair_shares"])
        my_pairs = pairs_per_step / world if pair_shares else pairs_per_step * (first_count / n)
        kernel_names = {1: "k_pairs (ordered pairs)", 2: "k_pairs (per-pair wrap)",
                        3: P3M_KERNEL_NAMES[st.get("p3m_pooled", 0)].split(" ")[0],
                        4: "k_pairs (P3M, all pairs)",
                        5: "k_pairs_sym (each unordered pair once, both directions)"}
        achieved = my_pairs * flops / (kernel_avg_ms * 1e-3) / 1e12
        # fp64 flops the kernel really issues per ordered interaction (FMA = 2, rsqrt = 1):
        # ordered kernel 25 (no per-pair wrap, +6 for the rsqrt refinement); the pair-once
        # kernel issues 31 per unordered pair
        executed = {1: 25.0, 2: 31.0, 5: 15.5}.get(st.get("kernel_variant"))
        # HBM bytes per launch and the FP64 pipe occupancy come from the committed rocprofv3 counter
        # passes of this same command (FETCH_SIZE doubled as the gfx950 correction requires, plus
        # WRITE_SIZE) - quoted only while the profile was taken from the kernel sources of this tree
        traffic = pipe_busy = None
        stale = None
        prof = os.path.join(ROOT, "profiles", "latest_direct128.json" if not use_pm else "latest_p3m.json")
        if os.path.exists(prof) and world == 1 and n_side == (128 if not use_pm else 256):
            from source_sha import kernel_sources_sha
            pj = json.load(open(prof))
            stale = pj.get("kernel_sources_sha") != kernel_sources_sha()
            if not stale:
                d = pj.get("derived", {})
                if "hbm_fetch_bytes_x2" in d and "hbm_write_bytes" in d:
                    traffic = d["hbm_fetch_bytes_x2"] + d["hbm_write_bytes"]
                pipe_busy = d.get("fp64_pipe_busy_frac")
        line = {
            "metric": "pair interactions/sec", "value": value, "unit": "Ginter/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" + (" (REHEARSAL: all ranks share one GPU)" if rehearsal else ""),
            "comm_backend": comm_backend,
            "config": {"workload": workload, "n_particles": n, "parallelism": f"i-partition x{world}",
                       "flops_per_interaction": flops,
                       "kernel": kernel_names.get(st.get("kernel_variant"), "?"),
                       "pct_fp64_vector_peak": 100.0 * value * 1e9 * flops / (world * FP64_VECTOR_PEAK_TFLOPS * 1e12),
                       # the literature's 20 flop per direct interaction, for comparison with other codes
                       "pct_fp64_vector_peak_at_20_flops": (None if use_pm else
                                                            100.0 * value * 1e9 * 20 / (world * FP64_VECTOR_PEAK_TFLOPS * 1e12))},
            "roofline": {"bound": "valu_fp64", "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP64_VECTOR_PEAK_TFLOPS,
                         "executed_flops_per_interaction": executed,
                         "executed_frac": (achieved * executed / flops / FP64_VECTOR_PEAK_TFLOPS
                                           if executed else None),
                         "fp64_pipe_busy_frac": pipe_busy,
                         "frac_note": "frac counts the 25 (40 for P3M) algorithmic flop of every interaction; "
                                      "executed_frac counts the fp64 flop the kernel really issues (the pair-once "
                                      "kernel serves two interactions with one evaluation); fp64_pipe_busy_frac is "
                                      "the FP64 pipe occupancy from the committed rocprofv3 counter pass",
                         "traffic": traffic, "traffic_unit": "bytes per launch (rocprofv3 PMC, committed profile of this command)",
                         "stale_profile": stale,
                         "algorithmic_bytes": 56.0 * n, "kernel_ms": kernel_avg_ms,
                         "prepare_ms": float(np.mean(prep_ms)),
                         "subtiles_wrapped_frac": (st.get("subtiles_wrapped", 0) /
                                                   max(1, st.get("subtiles_total", 0)))},
        }
        if mg is not None:
            ag, rs, cp = (float(np.mean([c[k] for c in comm_ms])) for k in range(3))
            st_mg = mg.stats()
            halo = bool(st_mg["halo_exchange"])
            line["config"]["parallelism"] = (
                f"block partition x{world}, libnbmg: "
                + ("halo exchange (ncclSend / ncclRecv of the particles within a cutoff radius of a "
                   "peer's block), " if halo else "ncclAllGather of positions per evaluation, ")
                + ("shares of the unordered pairs + ncclReduceScatter" if pair_shares
                   else "own targets against all sources, no reduction"))
            line["comm"] = {"backend": comm_backend, "allgather_ms": ag, "reduce_scatter_ms": rs,
                            "compute_ms": cp, "allgather_bytes_per_rank": st_mg["allgather_bytes"],
                            "reduce_bytes_per_rank": st_mg["reduce_bytes"],
                            "halo_particles_rank0": st_mg["halo_particles"], "rank": 0}
        if mesh is not None:
            line["config"]["mesh_force_ms"] = float(np.mean(mesh_ms[args.warmup:]))
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(p, pos, m, args.mode)
        default_run = (world == 1 and mg is None and args.mode == "direct" and n_side == 128
                       and not args.with_mesh and not args.no_extras)
        if default_run:
            # the other single-GPU configurations of the baseline, measured by the same process
            # under the same clock (not part of `value`)
            del x, y, z, acc, mass
            torch.cuda.empty_cache()
            extra = {}
            for name, fn in (("p3m256", lambda: extra_p3m256(lib, ffi, torch, dev, stream)),
                             ("kdk32", lambda: extra_kdk32(lib, ffi, torch, dev, stream)),
                             ("step128", extra_step128),
                             ("p3m128_evolved", lambda: extra_p3m128_evolved(lib, ffi, torch, dev, stream))):
                try:
                    extra[name] = fn()
                except Exception as e:   # the headline line must still be printed
                    extra[name] = {"error": repr(e)}
                torch.cuda.empty_cache()
            line["extra"] = extra
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
    if mg is not None:
        torch.cuda.synchronize()
        mg.shutdown()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

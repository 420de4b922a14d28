"""libnbmg with MORE THAN ONE rank on the one GPU of the test box.  RCCL refuses two ranks on one
device, so the ranks here talk through the library's host-staged transport (NBMG_TRANSPORT=host:
the same nbmg_* calls, every collective carried through a shared-memory segment).  What runs for
real is everything of the multi-rank flow except RCCL itself: the block partition, the padded
in-place all-gather, the pair shares with their reduce-scatter, the halo trade of the P3M short
range (intervals, count matrix, packs, own + received sources), the engine's sharded step with its
collective order, and bench.py launching its own ranks.
"""
import json
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, normalised_error
from nbody_amd import Config, ffi
import datasets

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _host_transport_id():
    """A communicator id of the host-staged kind, drawn without touching the GPU."""
    os.environ["NBMG_TRANSPORT"] = "host"
    try:
        return ffi.load_multi_gpu().unique_id()
    finally:
        del os.environ["NBMG_TRANSPORT"]


def _set(n_side, shifted):
    pos = datasets.jittered_lattice(n_side, seed=77)
    m = datasets.masses(n_side ** 3, equal=False)
    if shifted:
        # legal input: positions one box length outside [0, L) (libnbpp folds them back); a block's
        # x interval must be that of the folded positions
        rng = np.random.Generator(np.random.MT19937(11))
        pos = [c + rng.integers(-1, 2, size=c.size).astype(np.float64) for c in pos]
    return pos, m


def _worker(rank, world, ident, case, n_side, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import nbody_amd
    lib = nbody_amd.load(device=0)
    mg = ffi.load_multi_gpu()
    mg.init(rank, world, ident, 0)
    assert mg.transport() == "host" and mg.lib.nbmg_rank() == rank and mg.lib.nbmg_nranks() == world
    use_pm = case.startswith("p3m")
    pos, m = _set(n_side, shifted=(case == "p3m-halo-shifted"))
    n = m.size
    p = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=use_pm).pp_params()
    b, c = mg.partition(n, rank, world)
    padded = mg.padded_length(n, world)
    own = [torch.as_tensor(a[b:b + c].copy(), device="cuda:0") for a in pos]
    # masses: gathered once, as the engine does
    m_full = torch.zeros(padded, dtype=torch.float64, device="cuda:0")
    mg.allgather_device(n, torch.as_tensor(m[b:b + c].copy(), device="cuda:0"), m_full)
    assert np.array_equal(m_full[:n].cpu().numpy(), m)
    mg.set_option(ffi.MG_OPT_TIMING, 1)
    mg.set_option(ffi.MG_OPT_PAIR_SHARES, -1 if case == "direct-ordered" else 0)
    mg.set_option(ffi.MG_OPT_HALO_EXCHANGE, {"p3m-allgather": -1, "p3m-halo-forced": 1}.get(case, 0))
    acc = [torch.full((max(c, 1),), 0.25, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    stats = None
    for _ in range(2):   # twice: the library's buffers are reused, its partial sums cleared per call
        mg.accumulate_device(p, n, *own, m_full, *acc)
        torch.cuda.synchronize()
        stats = mg.stats()
    # the small collectives the engine uses
    v = torch.full((5,), float(rank + 1), dtype=torch.float64, device="cuda:0")
    mg.allreduce_sum_device(v)
    w = torch.tensor([float(rank), -float(rank)], dtype=torch.float64, device="cuda:0")
    mg.allreduce_max_device(w)
    a_max = mg.max_accel_sq_device(acc, count=c)
    torch.cuda.synchronize()
    assert torch.equal(v.cpu(), torch.full((5,), world * (world + 1) / 2.0, dtype=torch.float64))
    assert w.cpu().tolist() == [float(world - 1), 0.0]
    np.save(os.path.join(out_dir, f"acc_{rank}.npy"), np.stack([a[:c].cpu().numpy() for a in acc]))
    json.dump({"stats": stats, "a_max": a_max, "count": c}, open(os.path.join(out_dir, f"info_{rank}.json"), "w"))
    mg.shutdown()


CASES = ["direct-shares", "direct-ordered", "p3m-allgather", "p3m-halo-auto", "p3m-halo-forced", "p3m-halo-shifted"]


@pytest.mark.parametrize("world,case", [(2, c) for c in CASES] + [(3, "direct-shares"), (3, "p3m-halo-auto")])
def test_ranks_sharing_the_gpu_equal_one_rank(lib, tmp_path, world, case):
    import torch.multiprocessing as mp
    n_side = 36 if case.startswith("p3m") else 25          # 25^3 = 15625: not divisible by 2 or 3 (padding)
    ident = _host_transport_id()
    mp.spawn(_worker, args=(world, ident, case, n_side, str(tmp_path)), nprocs=world, join=True)
    use_pm = case.startswith("p3m")
    pos, m = _set(n_side, shifted=(case == "p3m-halo-shifted"))
    n = m.size
    p = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=use_pm).pp_params()
    dev = [torch.as_tensor(a, device="cuda:0") for a in (*pos, m)]
    want = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    lib.set_option(ffi.OPT_DETERMINISTIC, 1)
    try:
        lib.accumulate_device(p, *dev, *want)
        torch.cuda.synchronize()
    finally:
        lib.set_option(ffi.OPT_DETERMINISTIC, 0)
    want = [w.cpu().numpy() for w in want]
    got = np.concatenate([np.load(tmp_path / f"acc_{r}.npy") for r in range(world)], axis=1)
    assert got.shape == (3, n)
    got = [(g - 0.25) / 2.0 for g in got]                    # two accumulating calls onto 0.25
    assert normalised_error(got, want) < TOL
    infos = [json.load(open(tmp_path / f"info_{r}.json")) for r in range(world)]
    assert sum(i["count"] for i in infos) == n
    a_max = max(1e-9, float(np.max(sum((2.0 * w + 0.25) ** 2 for w in want))))
    for i in infos:                                          # every rank derives the same time step
        assert i["a_max"] == pytest.approx(a_max, rel=1e-9)
        st = i["stats"]
        assert st["pair_shares"] == (1 if case == "direct-shares" else 0)
        assert st["halo_exchange"] == (1 if "halo" in case else 0)
        if "halo" in case:       # slabs of a lattice in index order: about two cutoff radii of halo
            assert 0 < st["halo_particles"] < 0.6 * n
            assert st["allgather_bytes"] == 32 * st["halo_particles"]
        else:
            assert st["allgather_bytes"] == 24 * (ffi.load_multi_gpu().padded_length(n, world) // world) * (world - 1)


def _mismatch_worker(rank, world, ident, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["NBMG_HOST_TRANSPORT_TIMEOUT"] = "20"
    import nbody_amd
    nbody_amd.load(device=0)
    mg = ffi.load_multi_gpu()
    mg.init(rank, world, ident, 0)
    v = torch.ones(8, dtype=torch.float64, device="cuda:0")
    message = ""
    try:
        if rank == 0:
            mg.allreduce_sum_device(v)
        else:
            mg.allreduce_max_device(v)
    except ffi.NbppError as e:
        message = str(e)
    open(os.path.join(out_dir, f"msg_{rank}.txt"), "w").write(message)
    mg.shutdown()


def test_ranks_in_different_collectives_get_an_error_not_a_hang(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_mismatch_worker, args=(2, _host_transport_id(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert "disagree on the collective" in open(tmp_path / f"msg_{r}.txt").read()


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run_engine(tmp_path, tag, ini_text, cycles, world, log=True):
    """`nbody_host_tests run` as `world` processes sharing GPU 0 (world 0: the plain engine)."""
    exe = os.path.join(ROOT, "nbody_amd", "host", "nbody_host_tests")
    ini = tmp_path / f"{tag}.ini"
    ini.write_text(ini_text)
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    procs = []
    port = _free_port()
    for r in range(max(world, 1)):
        env = dict(base)
        logdir = tmp_path / f"{tag}_log{r}"
        logdir.mkdir()
        if world:
            env.update({"WORLD_SIZE": str(world), "RANK": str(r), "LOCAL_RANK": str(r), "MASTER_PORT": str(port),
                        "NBMG_TRANSPORT": "host", "NBMG_ONE_DEVICE": "0", "NBMG_RUN_TOKEN": f"{tag}{os.getpid()}",
                        "NBMG_HOST_TRANSPORT_TIMEOUT": "60"})
        args = [exe, "run", str(ini), str(cycles), "", str(logdir) if log else ""]
        procs.append(subprocess.Popen(args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env))
    outs = []
    for pr in procs:
        out, _ = pr.communicate(timeout=300)
        outs.append(out)
    for pr, out in zip(procs, outs):
        assert pr.returncode == 0, out
    final = [ln for ln in outs[0].splitlines() if ln.startswith("final:")]
    numbers = np.array([float(v) for ln in final for v in re.findall(r"[-+]?\d+\.\d+(?:e[-+]?\d+)?", ln)])
    csv = (tmp_path / f"{tag}_log0" / "diagnostics.csv")
    rows = csv.read_text().strip().splitlines() if log and csv.exists() else []
    return numbers, rows, outs


@pytest.mark.parametrize("world", [2, 3])
def test_cpp_engine_on_several_ranks_pair_force_only_with_diagnostics(tmp_path, world):
    """Pair force only, no gas, diagnostics on: the step keeps no mesh, so every diagnostics row
    makes the engine deposit and all-reduce a density on demand - a collective that every rank has
    to issue, not only the one that writes the row.  The run must finish (the host transport turns a
    rank that went missing from a collective into an error) and equal the single-process engine."""
    ini = ("[domain]\nmesh_size = 16\n[initial_conditions]\nn_per_side = 16\n[p3m]\nuse_pm = false\nuse_pp = true\n"
           "[hydro]\nuse_hydro = false\n[output]\ndebug_info_every_cycles = 2\n")
    a, rows_a, _ = _run_engine(tmp_path, "one", ini, 6, 0)
    b, rows_b, outs = _run_engine(tmp_path, f"w{world}", ini, 6, world)
    assert a.size == b.size and a.size >= 6          # a, dt, max|a|^2, particle 0's acceleration
    assert np.allclose(a, b, rtol=1e-9, atol=0.0), outs[0]
    assert len(rows_a) == len(rows_b) >= 4          # header, cycle 0, 2, 4, 6
    for ra, rb in zip(rows_a[1:], rows_b[1:]):
        va, vb = (np.array([float(t) for t in row.split(",")]) for row in (ra, rb))
        physics = slice(0, 18)       # cycle, time, a, masses, momenta, energies (pe_dm from the on-demand
        #                              mesh solve), time steps; written with six significant digits
        assert np.allclose(va[physics], vb[physics], rtol=3e-6, atol=1e-12), (ra, rb)


def test_cpp_engine_on_two_ranks_full_step(tmp_path):
    """Mesh + short range + gas at 32^3 on two ranks: block-wise density deposits summed by
    all-reduce, P3M short range with halos, redundant mesh solve and gas."""
    ini = "[domain]\nmesh_size = 32\n[initial_conditions]\nn_per_side = 32\n[output]\ndebug_info_every_cycles = 3\n"
    a, _, _ = _run_engine(tmp_path, "one", ini, 6, 0)
    b, _, outs = _run_engine(tmp_path, "two", ini, 6, 2)
    assert a.size == b.size and a.size >= 8
    assert np.allclose(a, b, rtol=1e-9, atol=0.0), outs[0]


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` from a bare shell (no launcher, no WORLD_SIZE): the parent starts
    the ranks, one JSON line comes back, exit code 0.  On this one-GPU box both ranks share device 0
    (NBODY_BENCH_ONE_DEVICE=1), which the line says."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["NBODY_BENCH_ONE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n-side", "32", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["value"] > 0
    assert "host-staged" in line["comm_backend"] and "REHEARSAL" in line["data"]
    assert line["comm"]["reduce_bytes_per_rank"] > 0 and "ncclReduceScatter" in line["config"]["parallelism"]

"""libnbmg (the multi-GPU layer: block partition + RCCL collectives + libnbpp) on the one GPU of
the test box: a communicator of ONE rank.  Every collective the multi-rank path issues
(in-place all-gather, in-place reduce-scatter, all-reduce) is really called on RCCL; with one rank
they move nothing, so the results must equal the single-GPU entry points of libnbpp - bit for bit
where the kernel is reproducible, to the path's 1e-10 otherwise.  The partition arithmetic for
more ranks is covered on the CPU (tests/test_abi.py, tests/test_distributed.py).
"""
import numpy as np
import pytest
import torch

from conftest import normalised_error
from nbody_amd import Config, ffi
import datasets

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture(scope="module")
def mg(lib):
    L = ffi.load_multi_gpu()
    L.init(0, 1, None, 0)
    assert L.lib.nbmg_rank() == 0 and L.lib.nbmg_nranks() == 1
    L.set_option(ffi.MG_OPT_TIMING, 1)
    yield L
    L.set_option(ffi.MG_OPT_PAIR_SHARES, 0)
    # the communicator is destroyed with the process: nbmg_shutdown would also shut libnbpp down
    # under the session-wide `lib` fixture


def _set(n_side, equal=True):
    pos = datasets.jittered_lattice(n_side, seed=77)
    m = datasets.masses(n_side ** 3, equal=equal)
    return [torch.as_tensor(a, device="cuda:0") for a in (*pos, m)]


@pytest.mark.parametrize("use_pm", [True, False])
def test_one_rank_equals_single_gpu_entry_point(lib, mg, use_pm):
    n_side = 24
    n = n_side ** 3
    x, y, z, m = _set(n_side, equal=use_pm)
    p = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=use_pm).pp_params()
    lib.set_option(ffi.OPT_DETERMINISTIC, 1)
    try:
        want = [torch.full((n,), 0.5, dtype=torch.float64, device="cuda:0") for _ in range(3)]
        lib.accumulate_device(p, x, y, z, m, *want)
        mg.set_option(ffi.MG_OPT_PAIR_SHARES, -1)
        got = [torch.full((n,), 0.5, dtype=torch.float64, device="cuda:0") for _ in range(3)]
        mg.accumulate_device(p, n, x, y, z, m, *got)
        torch.cuda.synchronize()
        st = mg.stats()
        assert st["pair_shares"] == 0 and st["allgather_bytes"] == 0 and st["total_ms"] > 0.0
        for g, w in zip(got, want):
            assert torch.equal(g, w)
    finally:
        lib.set_option(ffi.OPT_DETERMINISTIC, 0)


def test_one_rank_pair_shares_with_reduce_scatter(lib, mg):
    """Direct summation the way several ranks run it: share 0 of 1 of the unordered pairs into
    zeroed full-length sums, reduce-scatter, add."""
    n_side = 32
    n = n_side ** 3
    x, y, z, m = _set(n_side, equal=False)
    p = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=False).pp_params()
    lib.set_option(ffi.OPT_DETERMINISTIC, 1)
    want = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    lib.accumulate_device(p, x, y, z, m, *want)
    lib.set_option(ffi.OPT_DETERMINISTIC, 0)
    mg.set_option(ffi.MG_OPT_PAIR_SHARES, 1)
    try:
        got = [torch.full((n,), 2.0, dtype=torch.float64, device="cuda:0") for _ in range(3)]
        for _ in range(2):   # twice: the library's sums must be cleared per call
            mg.accumulate_device(p, n, x, y, z, m, *got)
        torch.cuda.synchronize()
        assert mg.stats()["pair_shares"] == 1
        assert lib.stats()["kernel_variant"] == ffi.KERNEL_DIRECT_SYMMETRIC
        ref = [w.cpu().numpy() for w in want]
        assert normalised_error([(g.cpu().numpy() - 2.0) / 2.0 for g in got], ref) < TOL
    finally:
        mg.set_option(ffi.MG_OPT_PAIR_SHARES, 0)


def test_one_rank_collectives_are_identities(mg):
    n = 1000
    own = torch.arange(n, dtype=torch.float64, device="cuda:0") * 0.25
    full = torch.full((mg.padded_length(n, 1),), -1.0, dtype=torch.float64, device="cuda:0")
    mg.allgather_device(n, own, full)
    data = own.clone()
    mg.allreduce_sum_device(data)
    mg.allreduce_max_device(data)
    torch.cuda.synchronize()
    assert torch.equal(full, own) and torch.equal(data, own)
    a = [own, -own, 2.0 * own]
    assert mg.max_accel_sq_device(a) == float((own * own * 6.0).max())
    zero = [torch.zeros(4, dtype=torch.float64, device="cuda:0")] * 3
    assert mg.max_accel_sq_device(zero, count=0) == 1e-9


def test_x_extent_is_the_complement_of_the_longest_gap(mg):
    L = 2.0
    rng = np.random.Generator(np.random.MT19937(3))
    # a block that wraps through x = 0: [1.7, 2.0) and [0, 0.3)
    xs = np.concatenate([1.7 + 0.3 * rng.random(500), 0.3 * rng.random(500)])
    lo, hi = mg.x_extent_device(torch.as_tensor(xs, device="cuda:0"), L)
    bin_w = L / 4096
    assert lo > hi and xs[xs > 1.0].min() - bin_w <= lo <= xs[xs > 1.0].min()
    assert xs[xs < 1.0].max() <= hi <= xs[xs < 1.0].max() + bin_w
    # an ordinary block, the ends of the box included
    xs = 0.5 + 0.25 * rng.random(300)
    lo, hi = mg.x_extent_device(torch.as_tensor(xs, device="cuda:0"), L)
    assert xs.min() - bin_w <= lo <= xs.min() and xs.max() <= hi <= xs.max() + bin_w
    # x = L is the periodic image of x = 0: both in the first bin
    xs = np.array([0.0, L])
    assert mg.x_extent_device(torch.as_tensor(xs, device="cuda:0"), L) == (0.0, pytest.approx(bin_w))
    # positions outside [0, L) are legal input (libnbpp folds them into the box): the interval is that
    # of the folded positions, and selecting with it finds the same particles as numpy does on them
    xs = np.concatenate([-0.3 * L + 0.05 * L * rng.random(400), 1.3 * L + 0.05 * L * rng.random(400)])
    folded = np.mod(xs, L)                                   # [0.7 L, 0.75 L) and [0.3 L, 0.35 L)
    lo, hi = mg.x_extent_device(torch.as_tensor(xs, device="cuda:0"), L)
    assert lo < hi and folded.min() - bin_w <= lo <= folded.min() and folded.max() <= hi <= folded.max() + bin_w
    others = [torch.as_tensor(L * rng.random(xs.size), device="cuda:0") for _ in range(3)]
    k = mg.halo_select_device(torch.as_tensor(xs, device="cuda:0"), *others, L, lo, hi, 0.0)
    assert k == xs.size
    probe = L * (rng.random(5000) * 3.0 - 1.0)               # from -L to 2 L
    want = np.count_nonzero(np.mod(probe - (lo - 0.01), L) <= (hi - lo) + 0.02)
    rest = [torch.as_tensor(L * rng.random(probe.size), device="cuda:0") for _ in range(3)]
    assert mg.halo_select_device(torch.as_tensor(probe, device="cuda:0"), *rest, L, lo, hi, 0.01) == want
    # spread over the whole box: no gap wider than the block itself is found -> an interval that
    # still holds every particle
    xs = L * rng.random(100000)
    lo, hi = mg.x_extent_device(torch.as_tensor(xs, device="cuda:0"), L)
    width = hi - lo if hi > lo else hi - lo + L
    assert width > 0.99 * L
    assert mg.x_extent_device(torch.empty(0, dtype=torch.float64, device="cuda:0"), L) == (0.0, L)


def test_halo_select_equals_numpy_selection(mg):
    L = 1.0
    n = 20000
    rng = np.random.Generator(np.random.MT19937(5))
    host = [rng.random(n) * L for _ in range(3)] + [rng.random(n) + 0.5]
    dev = [torch.as_tensor(a, device="cuda:0") for a in host]
    for lo, hi, margin in [(0.25, 0.5, 0.03), (0.9, 0.1, 0.05), (0.0, 0.125, 0.02), (0.3, 0.3, 0.0), (0.1, 0.9, 0.2)]:
        width = (hi - lo if hi > lo else hi - lo + L) + 2 * margin
        d = np.mod(host[0] - (lo - margin), L)
        want = np.nonzero(d <= width)[0] if width < L else np.arange(n)
        assert mg.halo_select_device(*dev, L, lo, hi, margin) == want.size
        cap = want.size + 7
        out = torch.full((4 * cap,), -1.0, dtype=torch.float64, device="cuda:0")
        assert mg.halo_select_device(*dev, L, lo, hi, margin, out=out) == want.size
        got = out.cpu().numpy().reshape(4, cap)
        for c in range(4):
            assert np.array_equal(got[c, :want.size], host[c][want])
            assert np.all(got[c, want.size:] == -1.0)
    small = torch.empty(4 * 10, dtype=torch.float64, device="cuda:0")
    with pytest.raises(ffi.NbppError, match="room for 10"):
        mg.halo_select_device(*dev, L, 0.25, 0.5, 0.03, out=small)


@pytest.mark.parametrize("n_blocks", [2, 5])
def test_halo_exchange_between_virtual_ranks_equals_the_full_evaluation(lib, mg, n_blocks):
    """What NBMG_OPT_HALO_EXCHANGE does on several ranks, with the ranks played one after another
    on this GPU: every block learns its x interval, takes from every other block the particles
    within the cutoff radius of that interval, and evaluates its own targets against own + halo.
    The forces must be the ones of the evaluation against all particles (the halo holds every
    source inside the cutoff; the others contribute exactly nothing), and the halos must be a
    small part of the set."""
    n_side = 40
    n = n_side ** 3
    x, y, z, m = _set(n_side, equal=False)
    cfg = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True)
    p = cfg.pp_params()
    L = p.domain_size
    margin = float(np.sqrt(p.cutoff_sq)) * (1 + 1e-9)
    want = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    lib.accumulate_device(p, x, y, z, m, *want)
    blocks = [mg.partition(n, r, n_blocks) for r in range(n_blocks)]
    own = [[a[b:b + c] for a in (x, y, z, m)] for b, c in blocks]
    extents = [mg.x_extent_device(o[0], L) for o in own]
    got = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    received = 0
    for r, (b, c) in enumerate(blocks):
        parts = [own[r]]
        for q in range(n_blocks):
            if q == r:
                continue
            k = mg.halo_select_device(*own[q], L, *extents[r], margin)
            out = torch.empty(4 * k, dtype=torch.float64, device="cuda:0")
            assert mg.halo_select_device(*own[q], L, *extents[r], margin, out=out) == k
            parts.append([out[i * k:(i + 1) * k] for i in range(4)])
            received += k
        src = [torch.cat([part[i] for part in parts]) for i in range(4)]
        lib.accumulate_device(p, *src, *[g[b:b + c] for g in got], tgt_begin=0, n_tgt=c)
    torch.cuda.synchronize()
    err = normalised_error([g.cpu().numpy() for g in got], [w.cpu().numpy() for w in want])
    assert err < TOL
    # every block of a lattice in index order is an x slab: two halos of about one cutoff radius
    slab = L / n_blocks
    assert received < 1.3 * n * (2 * margin + 2 * L / n_side) / slab


def test_halo_code_path_on_one_rank(lib, mg):
    """NBMG_OPT_HALO_EXCHANGE = 2 takes the halo path of nbmg_accumulate_device on one rank too:
    the interval all-gather, the count matrix and an empty send/receive group really run on RCCL,
    and the forces equal the plain path's."""
    n_side = 24
    n = n_side ** 3
    x, y, z, m = _set(n_side, equal=True)
    p = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True).pp_params()
    lib.set_option(ffi.OPT_DETERMINISTIC, 1)
    try:
        want = [torch.full((n,), 0.5, dtype=torch.float64, device="cuda:0") for _ in range(3)]
        lib.accumulate_device(p, x, y, z, m, *want)
        mg.set_option(ffi.MG_OPT_HALO_EXCHANGE, 2)
        for _ in range(2):   # twice: the second call reuses the library's buffers
            got = [torch.full((n,), 0.5, dtype=torch.float64, device="cuda:0") for _ in range(3)]
            mg.accumulate_device(p, n, x, y, z, m, *got)
            torch.cuda.synchronize()
            st = mg.stats()
            assert st["halo_particles"] == 0 and st["allgather_bytes"] == 0 and st["total_ms"] > 0.0
            for g, w in zip(got, want):
                assert torch.equal(g, w)
    finally:
        mg.set_option(ffi.MG_OPT_HALO_EXCHANGE, 0)
        lib.set_option(ffi.OPT_DETERMINISTIC, 0)


def _halo_worker(rank, world, port, n_side, out_dir):
    """One of `world` processes on the one GPU: its block of the particles, the x interval and the
    halo selection of libnbmg on the device, the packs carried between the processes by gloo (RCCL
    refuses two ranks on one device), own + received particles through nbpp_accumulate_device."""
    import os
    import sys
    import torch.distributed as dist
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import nbody_amd
    lib = nbody_amd.load(device=0)
    mg = ffi.load_multi_gpu()
    mg.init(0, 1, None, 0)                      # a communicator of its own; the exchange below is gloo's
    n = n_side ** 3
    x, y, z, m = _set(n_side, equal=False)
    p = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True).pp_params()
    L = p.domain_size
    margin = float(np.sqrt(p.cutoff_sq)) * (1 + 1e-9)
    b, c = mg.partition(n, rank, world)
    own = [a[b:b + c].contiguous() for a in (x, y, z, m)]
    mine = torch.tensor(mg.x_extent_device(own[0], L), dtype=torch.float64)
    extents = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(extents, mine)
    # what every peer needs from me, and how much I get from each
    packs, give = [], torch.zeros(world, dtype=torch.int64)
    for q in range(world):
        if q == rank:
            packs.append(None)
            continue
        lo, hi = (float(v) for v in extents[q])
        k = mg.halo_select_device(*own, L, lo, hi, margin)
        out = torch.empty(4 * k, dtype=torch.float64, device="cuda:0")
        assert mg.halo_select_device(*own, L, lo, hi, margin, out=out) == k
        packs.append(out.cpu())
        give[q] = k
    counts = [torch.zeros(world, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, give)
    take = [int(counts[q][rank]) for q in range(world)]
    received = {}
    for q in range(world):                      # pairwise exchange, lower rank sends first
        if q == rank:
            continue
        buf = torch.empty(4 * take[q], dtype=torch.float64)
        if rank < q:
            dist.send(packs[q], q)
            dist.recv(buf, q)
        else:
            dist.recv(buf, q)
            dist.send(packs[q], q)
        received[q] = buf.to("cuda:0")
    parts = [own] + [[received[q][i * take[q]:(i + 1) * take[q]] for i in range(4)] for q in sorted(received)]
    src = [torch.cat([part[i] for part in parts]) for i in range(4)]
    acc = [torch.zeros(c, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    lib.accumulate_device(p, *src, *acc, tgt_begin=0, n_tgt=c)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"halo_acc_{rank}.npy"), np.stack([a.cpu().numpy() for a in acc]))
    np.save(os.path.join(out_dir, f"halo_bytes_{rank}.npy"), np.array([32 * sum(take), 24 * (n - c)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange_between_processes_equals_one_rank(lib, tmp_path, world):
    """World size 2 and 3 on the one GPU of the box: every process owns a block, exchanges only
    the particles within a cutoff radius of its peers' blocks, and the forces equal the one-rank
    evaluation to the path's 1e-10.  The bytes a rank receives are reported beside what the
    all-gather of every position would have brought it."""
    import socket
    import torch.multiprocessing as mp
    n_side = 40
    n = n_side ** 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_halo_worker, args=(world, port, n_side, str(tmp_path)), nprocs=world, join=True)
    x, y, z, m = _set(n_side, equal=False)
    p = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True).pp_params()
    want = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    lib.accumulate_device(p, x, y, z, m, *want)
    torch.cuda.synchronize()
    got = np.concatenate([np.load(tmp_path / f"halo_acc_{r}.npy") for r in range(world)], axis=1)
    assert normalised_error(list(got), [w.cpu().numpy() for w in want]) < TOL
    for r in range(world):
        halo_bytes, allgather_bytes = np.load(tmp_path / f"halo_bytes_{r}.npy")
        print(f"rank {r} of {world}: {halo_bytes} B of halo particles against {allgather_bytes} B all-gathered")
        assert 0 < halo_bytes < (0.5 if world == 2 else 1.0) * allgather_bytes


def _run_engine(tmp_path, ini_text, cycles, env_extra):
    import os
    import re
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "nbody_amd", "host", "nbody_host_tests")
    ini = tmp_path / "run.ini"
    ini.write_text(ini_text)
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.update(env_extra)
    r = subprocess.run([exe, "run", str(ini), str(cycles)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    final = [ln for ln in r.stdout.splitlines() if ln.startswith("final:")]
    numbers = [float(v) for ln in final for v in re.findall(r"[-+]?\d+\.\d+(?:e[-+]?\d+)?", ln)]
    return final, np.array(numbers)


def test_cpp_engine_as_one_rank_of_one_equals_the_single_gpu_engine(tmp_path):
    """The C++ engine started the way a launcher starts it per GPU (WORLD_SIZE=1, RANK=0): block
    partition, nbmg_accumulate_device with its RCCL calls, nbmg_max_accel_sq_device.  PP only with
    16^3 particles runs the reproducible kernel, so the run must equal the plain single-GPU engine
    digit for digit; the full step (mesh + short range + gas, 32^3) sums its density with atomics
    and is compared to 1e-9."""
    pp_only = ("[domain]\nmesh_size = 16\n[initial_conditions]\nn_per_side = 16\n[p3m]\nuse_pm = false\n"
               "use_pp = true\n[hydro]\nuse_hydro = false\n")
    env = {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}
    plain, _ = _run_engine(tmp_path, pp_only, 5, {})
    ranked, _ = _run_engine(tmp_path, pp_only, 5, env)
    assert len(plain) >= 2 and plain == ranked
    full = "[domain]\nmesh_size = 32\n[initial_conditions]\nn_per_side = 32\n"
    _, a = _run_engine(tmp_path, full, 5, {})
    _, b = _run_engine(tmp_path, full, 5, env)
    assert a.size == b.size and a.size >= 8
    assert np.allclose(a, b, rtol=1e-9, atol=0.0)

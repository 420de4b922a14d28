"""The multi-GPU i-partition, rehearsed with two gloo ranks on CPU tensors.  The arithmetic is
injected: here the CPU oracle stands in for the GPU kernel, so what is tested is the partition,
the padded all-gather and the slice bookkeeping (tests/gloo_rehearsal.py; the product's multi-GPU
path is libnbmg, whose partition functions the second test calls directly)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, use_pm, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import datasets
    from nbody_amd import Config
    from gloo_rehearsal import SlicedParticles, force_evaluation
    from oracle.oracle import Oracle
    orc = Oracle()
    pos = datasets.clustered(n)
    m = datasets.masses(n, equal=False)
    p = Config(MESH_SIZE=8, N_PER_SIDE=8, USE_PM=use_pm).pp_params()
    parts = SlicedParticles(n, rank, world, torch.device("cpu"))
    parts.set_own_from_global(*[torch.from_numpy(c) for c in pos])
    mass = parts.gather_scalar_field(torch.from_numpy(m[parts.begin:parts.begin + parts.count]))
    acc = [torch.full((parts.count,), 7.0, dtype=torch.float64) for _ in range(3)]

    def compute(x, y, z, mm, ax, ay, az, tgt_begin, n_tgt):
        got = orc.accumulate(p, x.numpy(), y.numpy(), z.numpy(), mm.numpy(), tgt_begin=tgt_begin,
                             n_tgt=n_tgt, nthreads=2)
        for dst, src in zip((ax, ay, az), got):
            dst += torch.from_numpy(src)

    for _ in range(2):  # twice: buffers are reused between evaluations
        force_evaluation(parts, mass, acc, compute)
    assert torch.equal(mass, torch.from_numpy(m))

    if not use_pm:
        # the pair-once path: each rank fills a full-length array with its share, here "the
        # accelerations of my own slice", and the reduce-scatter must hand every rank its slice
        from gloo_rehearsal import force_evaluation_pairs

        def compute_pairs(x, y, z, mm, fx, fy, fz, part, n_parts):
            b, c = parts.begins[part], parts.counts[part]
            got = orc.accumulate(p, x.numpy(), y.numpy(), z.numpy(), mm.numpy(), tgt_begin=b,
                                 n_tgt=c, nthreads=2)
            for dst, src in zip((fx, fy, fz), got):
                dst[b:b + c] += torch.from_numpy(src)

        acc2 = [torch.full((parts.count,), -3.0, dtype=torch.float64) for _ in range(3)]
        work = None
        for _ in range(2):
            work = force_evaluation_pairs(parts, mass, acc2, compute_pairs, work)
        for a1, a2 in zip(acc, acc2):
            assert torch.equal(a1, a2)
    np.save(os.path.join(out_dir, f"acc_{rank}.npy"), np.stack([a.numpy() for a in acc]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,use_pm", [(1001, False), (1000, True)])
def test_two_rank_partition_matches_single_process(tmp_path, oracle, n, use_pm):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n, use_pm, str(tmp_path)), nprocs=2, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import datasets
    from nbody_amd import Config
    pos = datasets.clustered(n)
    m = datasets.masses(n, equal=False)
    p = Config(MESH_SIZE=8, N_PER_SIDE=8, USE_PM=use_pm).pp_params()
    want = np.stack(oracle.accumulate(p, *pos, m))
    got = np.concatenate([np.load(tmp_path / f"acc_{r}.npy") for r in range(2)], axis=1)
    assert got.shape == want.shape
    assert np.array_equal(got, want)  # same oracle, same order per target: bitwise


def _native_worker(rank, world, port, n, out_dir):
    """The block partition of libnbmg (nbmg_partition / nbmg_padded_length, the C++ code the GPU
    path runs) driving the same sequence of collectives as nbmg_accumulate_device, with gloo
    standing in for RCCL: in-place all-gather of whole blocks of a padded array, each rank's
    targets against all sources, and for direct summation the pair-share variant (full-length
    partial sums, reduce-scatter of whole blocks, add)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import datasets
    from nbody_amd import Config, ffi
    from oracle.oracle import Oracle
    orc = Oracle()
    mg = ffi.MultiGpuLibrary()
    begin, count = mg.partition(n, rank, world)
    padded = mg.padded_length(n, world)
    stride = padded // world
    pos = datasets.clustered(n)
    m = datasets.masses(n, equal=False)
    p = Config(MESH_SIZE=8, N_PER_SIDE=8, USE_PM=False).pp_params()
    own = [torch.from_numpy(c[begin:begin + count].copy()) for c in pos]
    full = [torch.full((padded,), float("nan"), dtype=torch.float64) for _ in range(3)]
    for c in range(3):
        full[c][begin:begin + count] = own[c]
        dist.all_gather_into_tensor(full[c], full[c][rank * stride:(rank + 1) * stride].clone())
        assert np.array_equal(full[c][:n].numpy(), pos[c])
    x, y, z = (f[:n].numpy() for f in full)
    acc = orc.accumulate(p, x, y, z, m, tgt_begin=begin, n_tgt=count, nthreads=2)
    # pair shares: this rank contributes the forces from the sources of ITS block on everybody
    sums = [torch.zeros(padded, dtype=torch.float64) for _ in range(3)]
    mm = np.zeros(n)
    mm[begin:begin + count] = m[begin:begin + count]
    part = orc.accumulate(p, x, y, z, mm, nthreads=2)
    for c in range(3):
        sums[c][:n] = torch.from_numpy(part[c])
        dist.all_reduce(sums[c])                      # gloo has no reduce-scatter: keep the own block
    acc2 = [s[rank * stride: rank * stride + count].numpy() for s in sums]
    np.save(os.path.join(out_dir, f"native_{rank}.npy"), np.stack(acc))
    np.save(os.path.join(out_dir, f"native_pairs_{rank}.npy"), np.stack(acc2))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1001, 1000])
def test_two_rank_native_block_partition(tmp_path, oracle, n):
    port = _free_port()
    mp.spawn(_native_worker, args=(2, port, n, str(tmp_path)), nprocs=2, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import datasets
    from nbody_amd import Config
    pos = datasets.clustered(n)
    m = datasets.masses(n, equal=False)
    p = Config(MESH_SIZE=8, N_PER_SIDE=8, USE_PM=False).pp_params()
    want = np.stack(oracle.accumulate(p, *pos, m))
    got = np.concatenate([np.load(tmp_path / f"native_{r}.npy") for r in range(2)], axis=1)
    assert got.shape == want.shape and np.array_equal(got, want)
    got2 = np.concatenate([np.load(tmp_path / f"native_pairs_{r}.npy") for r in range(2)], axis=1)
    norm = np.linalg.norm(want, axis=0)
    assert np.max(np.abs(got2 - want) / norm) < 1e-12     # a different summation order


def test_partition_shapes():
    from gloo_rehearsal import partition
    for n, w in ((10, 3), (8, 8), (2097152, 8), (5, 8), (0, 2)):
        begins, counts = partition(n, w)
        assert sum(counts) == n and begins[0] == 0
        assert all(begins[r + 1] == begins[r] + counts[r] for r in range(w - 1))
        assert max(counts) - min(counts) <= 1

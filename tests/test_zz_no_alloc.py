"""nbpp_accumulate_device_noalloc (the entry point that takes all its working memory from a fixed
array inside the library) against nbpp_accumulate_device, through the C ABI.

Bar: the reproducible kernels (ordered direct sum, P3M cell kernel with the members of a cell in a
fixed order: NBPP_OPT_DETERMINISTIC) must agree bit for bit, since both entry points run the same
kernels on the same inputs; the pair-once direct kernel adds with
fp64 atomics in arrival order, so there the bar is the path's parity tolerance,
max_i,c |a - a_ref|_c / ||a_ref,i||_2 <= 1e-10.  Every case makes two calls in a row, the second
on a smaller input, so that whatever the first left in the fixed array is in the way of the second.
"""
import numpy as np
import pytest
import torch

from conftest import normalised_error
from nbody_amd import Config, ffi
import datasets

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _device_set(n_side, seed):
    pos = datasets.jittered_lattice(n_side, seed=seed)
    m = datasets.masses(n_side ** 3, equal=(seed % 2 == 0))
    dev = [torch.as_tensor(a, device="cuda:0") for a in (*pos, m)]
    return dev


def _both(lib, p, dev, start):
    """(existing entry point, no-alloc entry point), each accumulating onto `start`."""
    out = []
    for call in (lib.accumulate_device, lib.accumulate_device_noalloc):
        acc = [torch.full_like(dev[0], start) for _ in range(3)]
        call(p, *dev, *acc)
        torch.cuda.synchronize()
        out.append([a.cpu().numpy() for a in acc])
    return out


@pytest.mark.parametrize("mode", ["direct_ordered", "direct_pair_once", "p3m"])
def test_noalloc_matches_allocating_entry_point_twice_in_a_row(lib, mode):
    lib.set_option(ffi.OPT_DETERMINISTIC, 0 if mode == "direct_pair_once" else 1)
    try:
        for n_side, seed in ((40, 2), (27, 5)):          # second call: smaller input
            p = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=(mode == "p3m")).pp_params()
            dev = _device_set(n_side, seed)
            want, got = _both(lib, p, dev, start=0.125)
            variant = lib.stats()["kernel_variant"]
            if mode == "p3m":
                assert variant == ffi.KERNEL_P3M_CELLS
            elif mode == "direct_pair_once":
                assert variant == ffi.KERNEL_DIRECT_SYMMETRIC
            else:
                assert variant == ffi.KERNEL_DIRECT
            if mode == "direct_pair_once":
                ref = [w - 0.125 for w in want]
                assert normalised_error([g - 0.125 for g in got], ref) < TOL
            else:
                for g, w in zip(got, want):
                    assert np.array_equal(g, w)
    finally:
        lib.set_option(ffi.OPT_DETERMINISTIC, 0)


def test_noalloc_rejects_sizes_beyond_its_maximum(lib):
    n = ffi.NOALLOC_MAX_N + 1
    p = Config(MESH_SIZE=64, N_PER_SIDE=64, USE_PM=False).pp_params()
    t = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(7)]
    with pytest.raises(ffi.NbppError) as e:
        lib.accumulate_device_noalloc(p, *t)
    assert e.value.code == -4 and "at most 262144" in str(e.value)
    # a search grid too fine for the fixed array: cutoff far below the particle spacing
    n_side = 32
    dev = _device_set(n_side, 2)
    cfg = Config(MESH_SIZE=512, N_PER_SIDE=n_side, USE_PM=True)
    acc = [torch.zeros_like(dev[0]) for _ in range(3)]
    with pytest.raises(ffi.NbppError) as e:
        lib.accumulate_device_noalloc(cfg.pp_params(), *dev, *acc)
    assert e.value.code == -4 and "fixed workspace" in str(e.value)
    torch.cuda.synchronize()
    assert all(float(a.abs().max()) == 0.0 for a in acc)
    # and the allocating entry point still takes the same call
    lib.accumulate_device(cfg.pp_params(), *dev, *acc)
    torch.cuda.synchronize()


def test_noalloc_at_its_maximum_size(lib, oracle):
    """64^3 particles, P3M and pair-once direct summation, sampled targets against the CPU oracle."""
    from oracle.oracle import EXACT_CELLS
    n_side = 64
    assert n_side ** 3 == ffi.NOALLOC_MAX_N
    pos = datasets.jittered_lattice(n_side)
    m = datasets.masses(n_side ** 3)
    dev = [torch.as_tensor(a, device="cuda:0") for a in (*pos, m)]
    for use_pm in (True, False):
        p = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=use_pm).pp_params()
        acc = [torch.zeros_like(dev[0]) for _ in range(3)]
        lib.accumulate_device_noalloc(p, *dev, *acc)
        torch.cuda.synchronize()
        n_tgt = 2048 if use_pm else 64
        want = oracle.accumulate(p, *pos, m, tgt_begin=0, n_tgt=n_tgt,
                                 traversal=EXACT_CELLS if use_pm else 0)
        got = [a[:n_tgt].cpu().numpy() for a in acc]
        assert normalised_error(got, want) < TOL

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

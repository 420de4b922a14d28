"""BASELINE config 4 at its stated size, as far as one GPU goes: direct summation at N = 256^3,
the work ONE rank of eight does per force evaluation, in both forms libnbmg issues it.

* i-partition (`nbpp_accumulate_device` with the rank's block of targets against all 16.8 M
  sources - the plan with the all-gather only): sampled targets of the block against the oracle's
  long-double sum of the same terms.  (Against the reference-order double sum the bar would be met
  too on most targets, but that sum's own rounding reaches a few 1e-11 at this size.)
* pair shares (`nbpp_accumulate_pairs_device`, one eighth of the unordered pairs into full-length
  arrays - what the reduce-scatter then combines): every pair of the share is applied to both its
  members, so the share by itself exerts no net force: sum_i m_i a_i = 0 to rounding.

About 35 s of GPU.
"""
import numpy as np
import pytest

from nbody_amd import Config, ffi
import datasets

pytestmark = pytest.mark.gpu


def test_config4_one_rank_of_eight_at_256_cubed(lib, oracle):
    import torch
    n_side, ranks, rank = 256, 8, 3
    n = n_side ** 3
    pos = datasets.jittered_lattice(n_side)
    config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=False)
    m = np.full(n, config.DM_PARTICLE_MASS)
    p = config.pp_params()
    x, y, z, mm = [torch.from_numpy(a).cuda() for a in pos + [m]]
    stride = n // ranks
    begin = rank * stride

    # ---- the rank's block of targets against every source
    acc = [torch.zeros(stride, dtype=torch.float64, device="cuda") for _ in range(3)]
    lib.accumulate_device(p, x, y, z, mm, *acc, tgt_begin=begin, n_tgt=stride)
    torch.cuda.synchronize()
    st = lib.stats()
    assert st["kernel_variant"] == ffi.KERNEL_DIRECT          # the reproducible ordered-pair kernel
    got = [a.cpu().numpy() for a in acc]
    rng = np.random.default_rng(4)
    worst = 0.0
    for i in np.sort(rng.choice(stride, 12, replace=False)):
        want = oracle.accumulate(p, *pos, m, tgt_begin=begin + int(i), n_tgt=1, long_double=True)
        ref = np.array([want[0][0], want[1][0], want[2][0]])
        g = np.array([got[0][i], got[1][i], got[2][i]])
        worst = max(worst, float(np.max(np.abs(g - ref)) / np.linalg.norm(ref)))
    assert worst < 1e-10, worst
    del acc

    # ---- the rank's share of the unordered pairs, both directions, into full-length sums
    sums = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(3)]
    lib.accumulate_pairs_device(p, x, y, z, mm, *sums, part=rank, n_parts=ranks)
    torch.cuda.synchronize()
    assert lib.stats()["kernel_variant"] == ffi.KERNEL_DIRECT_SYMMETRIC
    share = [s.cpu().numpy() for s in sums]
    size = float(np.mean(np.sqrt(share[0] ** 2 + share[1] ** 2 + share[2] ** 2)))
    for c in range(3):
        assert abs(float(np.sum(m * share[c]))) < 1e-9 * size * m[0] * n
    assert np.isfinite(size) and size > 0.0


def test_pair_shares_of_eight_ranks_add_up_to_the_whole_force_at_128_cubed(lib, oracle):
    """The multi-GPU direct path at the headline size, with the eight ranks played in turn on this
    GPU: every rank's share of the unordered pairs (`nbpp_accumulate_pairs_device`, what libnbmg
    calls) into zeroed full-length sums, the eight sums added in rank order in fp64 - what the
    reduce-scatter hands each rank for its block - and 48 sampled targets against the oracle (the
    reference-order sum at 1e-10, and the long-double sum of the same terms, which shows how much of
    that is the reference order's own rounding).  All eight shares together are one full evaluation
    (about 1.6 s of GPU)."""
    import torch
    n_side, ranks = 128, 8
    n = n_side ** 3
    pos = datasets.jittered_lattice(n_side)
    config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=False)
    m = np.full(n, config.DM_PARTICLE_MASS)
    p = config.pp_params()
    x, y, z, mm = [torch.from_numpy(a).cuda() for a in pos + [m]]
    total = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(3)]
    pairs = 0
    for rank in range(ranks):
        sums = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(3)]
        lib.accumulate_pairs_device(p, x, y, z, mm, *sums, part=rank, n_parts=ranks)
        torch.cuda.synchronize()
        st = lib.stats()
        assert st["kernel_variant"] == ffi.KERNEL_DIRECT_SYMMETRIC
        pairs += st["pairs_evaluated"]
        for t, s in zip(total, sums):
            t += s                                            # rank order, fp64: ncclSum's arithmetic
    assert abs(pairs - n * (n - 1)) <= ranks                  # the shares cover every ordered pair (rounded per share)
    got = [t.cpu().numpy() for t in total]
    rng = np.random.default_rng(11)
    worst = worst_exact = 0.0
    for i in np.sort(rng.choice(n, 48, replace=False)):
        g = np.array([got[0][i], got[1][i], got[2][i]])
        ref = np.array([c[0] for c in oracle.accumulate(p, *pos, m, tgt_begin=int(i), n_tgt=1)])
        exact = np.array([c[0] for c in oracle.accumulate(p, *pos, m, tgt_begin=int(i), n_tgt=1, long_double=True)])
        worst = max(worst, float(np.max(np.abs(g - ref)) / np.linalg.norm(ref)))
        worst_exact = max(worst_exact, float(np.max(np.abs(g - exact)) / np.linalg.norm(exact)))
    print(f"8 pair shares at 128^3, 48 targets: {worst:.2e} against the reference-order sum, "
          f"{worst_exact:.2e} against the long-double sum")
    assert worst < 1e-10, worst
    assert worst_exact < 1e-11, worst_exact
    # and the whole is momentum-free: every pair acted on both its members
    for c in range(3):
        assert abs(float(np.sum(m * got[c]))) < 1e-12 * float(np.sum(m * np.abs(got[c])))

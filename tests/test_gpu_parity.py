"""GPU parity: libnbpp (through its C ABI) against the CPU oracle and the golden vectors.

Tolerance for every force comparison: max_i,c |a_gpu - a_ref|_c / ||a_ref,i||_2 <= 1e-10
(fp64; two correct summation orders already differ at the 1e-12 level).  Integer results
(accepted-pair counts) and the kick / drift / max kernels are compared exactly.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, normalised_error, pair_force_scale
from nbody_amd import Config, ParticleSystem, ffi
from oracle import oracle as O
import datasets

pytestmark = pytest.mark.gpu
TOL = 1e-10


def gpu_acc(lib, p, pos, m, acc0=None):
    acc = [np.zeros(m.size) if acc0 is None else a.copy() for a in (acc0 or [None] * 3)]
    lib.accumulate(p, pos[0], pos[1], pos[2], m, *acc)
    return acc


# --------------------------------------------------------------------------------------
# the reference's own known-answer tests, written against the mirrored interface
# --------------------------------------------------------------------------------------
def _two_body(lib):
    config = Config()
    config.MESH_SIZE = 4
    config.CELL_SIZE = 2.0
    config.DOMAIN_SIZE = 8.0
    config.NUM_DM_PARTICLES = 2
    config.G = 1.0
    config.SOFTENING_SQUARED = 0.0
    sys_ = ParticleSystem(config, lib)
    sys_.add_particle(2.0, 2.0, 2.0, 0.0, 0.0, 0.0, 1.0)
    sys_.add_particle(5.0, 2.0, 2.0, 0.0, 0.0, 0.0, 1.0)
    return config, sys_


def test_kat_pure_newtonian_pp(lib):
    """tests/test_gravity.cpp:115-134."""
    config, sys_ = _two_body(lib)
    config.USE_PM = False
    config.CUTOFF_RADIUS_SQUARED = (config.DOMAIN_SIZE / 2.0) ** 2
    sys_.bin_and_assign_mass(config)
    sys_.compute_pp_forces(config)
    expected_f = 1.0 / 9.0
    assert sys_.acc_x[0] == pytest.approx(expected_f, rel=1e-14)
    assert sys_.acc_y[0] == 0.0 and sys_.acc_z[0] == 0.0
    assert sys_.acc_x[1] == pytest.approx(-expected_f, rel=1e-14)


def test_kat_p3m_cutoff_ignores_distant(lib):
    """tests/test_gravity.cpp:136-154."""
    config, sys_ = _two_body(lib)
    config.USE_PM = True
    config.CUTOFF_RADIUS = 2.0
    config.CUTOFF_RADIUS_SQUARED = 4.0
    config.R_SWITCH_START = 1.0
    config.R_SWITCH_START_SQ = 1.0
    config.CUTOFF_TRANSITION_WIDTH = 1.0
    sys_.bin_and_assign_mass(config)
    sys_.acc_x[:] = 0.0
    sys_.compute_pp_forces(config)
    assert sys_.acc_x[0] == 0.0 and sys_.acc_x[1] == 0.0


def test_kat_wrap_cases_through_the_kernel(lib, oracle):
    """tests/test_math.cpp:14-47 exercised end to end: two bodies 62, 2 and 32 apart in a box
    of 64 must attract along the shortest path."""
    config = Config(DOMAIN_SIZE=64.0, MESH_SIZE=8, N_PER_SIDE=2, USE_PM=False)
    p = config.pp_params()
    for xa, xb, sign in ((1.0, 63.0, -1.0), (32.0, 30.0, -1.0), (10.0, 42.0, +1.0)):
        pos = [np.array([xa, xb]), np.array([5.0, 5.0]), np.array([7.0, 7.0])]
        m = np.array([1.0, 1.0])
        got = gpu_acc(lib, p, pos, m)
        want = oracle.accumulate(p, *pos, m)
        assert np.sign(got[0][0]) == sign
        assert normalised_error(got, want) < TOL


# --------------------------------------------------------------------------------------
# golden vectors (independent long-double sums)
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["lattice8", "lattice12_even_exact", "uniform1000",
                                  "clustered1500"])
@pytest.mark.parametrize("mode", [0, 1])
def test_golden_small_sets(lib, name, mode):
    g = np.load(os.path.join(GOLDEN, "pp_small.npz"))
    pos = [np.ascontiguousarray(c) for c in g[f"{name}_pos"]]
    m, want = g[f"{name}_mass"], g[f"{name}_acc_mode{mode}"]
    pv = g[f"{name}_params_mode{mode}"]
    p = ffi.Params(*pv[:9], int(pv[9]), int(pv[10]))
    got = gpu_acc(lib, p, pos, m)
    floor = pair_force_scale(p, m, m.size) if "exact" in name else 0.0
    assert normalised_error(got, want, floor) < TOL


# --------------------------------------------------------------------------------------
# direct summation vs the oracle
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", [ffi.KERNEL_DIRECT, ffi.KERNEL_DIRECT_WRAP])
@pytest.mark.parametrize("per_lane", [1, 2, 4])
def test_direct_variants(lib, oracle, variant, per_lane):
    pos = datasets.jittered_lattice(16)
    m = datasets.masses(4096)
    p = Config(MESH_SIZE=16, N_PER_SIDE=16, USE_PM=False).pp_params()
    want = oracle.accumulate(p, *pos, m)
    lib.set_option(ffi.OPT_FORCE_KERNEL, variant)
    lib.set_option(ffi.OPT_TARGETS_PER_LANE, per_lane)
    try:
        got = gpu_acc(lib, p, pos, m)
        assert lib.stats()["kernel_variant"] == variant
    finally:
        lib.set_option(ffi.OPT_FORCE_KERNEL, 0)
        lib.set_option(ffi.OPT_TARGETS_PER_LANE, 0)
    assert normalised_error(got, want) < TOL


@pytest.mark.parametrize("per_lane", [1, 2, 4])
@pytest.mark.parametrize("equal_mass", [True, False])
def test_symmetric_kernel(lib, oracle, per_lane, equal_mass):
    """Every unordered pair evaluated once, applied in both directions (fp64 atomics)."""
    for pos in (datasets.zeldovich(16), datasets.clustered(5003), datasets.uniform_random(700),
                datasets.jittered_lattice(12, jitter=0.0)):
        n = pos[0].size
        m = datasets.masses(n, equal=equal_mass)
        p = Config(MESH_SIZE=16, N_PER_SIDE=16, USE_PM=False).pp_params()
        want = oracle.accumulate(p, *pos, m)
        lib.set_option(ffi.OPT_FORCE_KERNEL, ffi.KERNEL_DIRECT_SYMMETRIC)
        lib.set_option(ffi.OPT_TARGETS_PER_LANE, per_lane)
        try:
            rng = np.random.default_rng(n)
            acc0 = [rng.standard_normal(n) * 1e-3 for _ in range(3)]
            got = gpu_acc(lib, p, pos, m, acc0)
            assert lib.stats()["kernel_variant"] == ffi.KERNEL_DIRECT_SYMMETRIC
        finally:
            lib.set_option(ffi.OPT_FORCE_KERNEL, 0)
            lib.set_option(ffi.OPT_TARGETS_PER_LANE, 0)
        want = [w + a for w, a in zip(want, acc0)]
        assert normalised_error(got, want, pair_force_scale(p, m, n)) < TOL


@pytest.mark.parametrize("n_parts", [1, 3, 8])
def test_pair_shares_sum_to_the_full_evaluation(lib, oracle, n_parts):
    """nbpp_accumulate_pairs_device: the shares a multi-GPU run hands to its ranks, evaluated one
    after the other on one GPU into the same arrays, must add up to the complete accelerations."""
    import torch
    n = 9001
    pos = datasets.clustered(n)
    m = datasets.masses(n, equal=False)
    p = Config(MESH_SIZE=16, N_PER_SIDE=21, USE_PM=False).pp_params()
    want = oracle.accumulate(p, *pos, m)
    x, y, z, mm = _t(pos + [m])
    acc = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(3)]
    for part in range(n_parts):
        lib.accumulate_pairs_device(p, x, y, z, mm, *acc, part=part, n_parts=n_parts)
    torch.cuda.synchronize()
    assert normalised_error([a.cpu().numpy() for a in acc], want) < TOL
    with pytest.raises(ffi.NbppError):
        lib.accumulate_pairs_device(p, x, y, z, mm, *acc, part=3, n_parts=3)


def test_equal_mass_path_matches_weighted_path(lib, oracle):
    """Equal masses take the kernel variant that applies G*m once at the end; it must agree
    with the per-pair-mass variant and with the oracle, including a ragged last tile."""
    pos = datasets.zeldovich(16)
    pos = [c[:4001].copy() for c in pos]
    m = np.full(4001, 0.84 / 4001)
    p = Config(MESH_SIZE=16, N_PER_SIDE=16, USE_PM=False).pp_params()
    want = oracle.accumulate(p, *pos, m)
    fast = gpu_acc(lib, p, pos, m)
    lib.set_option(ffi.OPT_NO_EQUAL_MASS_PATH, 1)
    try:
        slow = gpu_acc(lib, p, pos, m)
    finally:
        lib.set_option(ffi.OPT_NO_EQUAL_MASS_PATH, 0)
    assert normalised_error(fast, want) < TOL and normalised_error(slow, want) < TOL
    assert normalised_error(fast, slow) < 1e-12
    m2 = m.copy()
    m2[-1] *= 1.0000001   # one unequal mass switches the whole call to the weighted path
    assert normalised_error(gpu_acc(lib, p, pos, m2), oracle.accumulate(p, *pos, m2)) < TOL


@pytest.mark.parametrize("split", [1, 3, 16])
def test_direct_source_split_is_deterministic(lib, oracle, split):
    pos = datasets.clustered(5000)
    m = datasets.masses(5000, equal=False)
    p = Config(MESH_SIZE=16, N_PER_SIDE=17, USE_PM=False).pp_params()
    want = oracle.accumulate(p, *pos, m)
    lib.set_option(ffi.OPT_SOURCE_SPLIT, split)
    try:
        a = gpu_acc(lib, p, pos, m)
        b = gpu_acc(lib, p, pos, m)
    finally:
        lib.set_option(ffi.OPT_SOURCE_SPLIT, 0)
    assert normalised_error(a, want) < TOL
    for c in range(3):
        assert np.array_equal(a[c], b[c])  # bitwise reproducible from run to run


@pytest.mark.parametrize("L", [1.0, 64.0, 10.0, 0.3])
def test_direct_box_sizes(lib, oracle, L):
    """Power-of-two and other box lengths: the image choice must be the reference's."""
    pos = datasets.uniform_random(3001, L=L)
    m = datasets.masses(3001, equal=False)
    p = Config(DOMAIN_SIZE=L, MESH_SIZE=9, N_PER_SIDE=14, USE_PM=False).pp_params()
    assert normalised_error(gpu_acc(lib, p, pos, m), oracle.accumulate(p, *pos, m)) < TOL


def test_direct_32cubed_full(lib, oracle):
    """32^3 particles (the reference's CPU-runnable size), PP only, every ordered pair."""
    pos = datasets.jittered_lattice(32)
    m = datasets.masses(32768)
    p = Config(MESH_SIZE=32, N_PER_SIDE=32, USE_PM=False).pp_params()
    want = oracle.accumulate(p, *pos, m)
    got = gpu_acc(lib, p, pos, m)
    assert normalised_error(got, want) < TOL
    st = lib.stats()
    assert st["pairs_evaluated"] == 32768 * 32767


def test_half_box_pairs_on_perfect_lattice(lib, oracle):
    """Even lattice, no jitter: a whole plane of partners sits at |d| == L/2 or one ulp off,
    where the reference leaves +-L/2 unwrapped (src/math_utils.h:13-14).  Forces there are
    O(1e-2) with a sign that depends on that choice."""
    for n_side in (8, 12, 16):
        pos = datasets.jittered_lattice(n_side, jitter=0.0)
        m = datasets.masses(n_side ** 3)
        p = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=False).pp_params()
        want = oracle.accumulate(p, *pos, m)
        got = gpu_acc(lib, p, pos, m)
        assert np.max(np.abs(want[0])) > 1e-3
        assert normalised_error(got, want, pair_force_scale(p, m, m.size)) < TOL


def test_positions_on_the_faces(lib, oracle):
    """Particles at 0, at the last double below L and at L itself (the drift can produce
    x == L, src/integrator.cpp:194)."""
    pos = datasets.uniform_random(600)
    for c in range(3):
        pos[c][:4] = [0.0, np.nextafter(1.0, 0.0), 1.0, 0.5]
    m = datasets.masses(600)
    for use_pm in (False, True):
        p = Config(MESH_SIZE=8, N_PER_SIDE=8, USE_PM=use_pm).pp_params()
        assert normalised_error(gpu_acc(lib, p, pos, m), oracle.accumulate(p, *pos, m)) < TOL


def test_accumulates_into_existing_acc(lib, oracle):
    """acc += sum (src/particles.cpp:362-364), not acc = sum."""
    pos = datasets.uniform_random(777)
    m = datasets.masses(777)
    p = Config(MESH_SIZE=8, N_PER_SIDE=9, USE_PM=False).pp_params()
    rng = np.random.default_rng(1)
    acc0 = [rng.standard_normal(777) for _ in range(3)]
    want = oracle.accumulate(p, *pos, m, acc=[a.copy() for a in acc0])
    got = gpu_acc(lib, p, pos, m, acc0)
    assert normalised_error(got, want) < TOL
    assert np.max(np.abs(got[0] - acc0[0])) > 0


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 255, 256, 257, 1023])
def test_ragged_sizes(lib, oracle, n):
    pos = datasets.uniform_random(n, seed=n)
    m = datasets.masses(n, equal=False, seed=n)
    for use_pm in (False, True):
        p = Config(MESH_SIZE=8, N_PER_SIDE=4, USE_PM=use_pm).pp_params()
        got = gpu_acc(lib, p, pos, m)
        want = oracle.accumulate(p, *pos, m)
        assert normalised_error(got, want) < TOL


def test_empty_input_and_errors(lib):
    p = Config(USE_PM=False).pp_params()
    e = np.zeros(0)
    lib.accumulate(p, e, e, e, e, e.copy(), e.copy(), e.copy())  # n == 0: no-op, no error
    a = np.zeros(8)
    bad = Config(USE_PM=False).pp_params()
    bad.domain_size = -1.0
    with pytest.raises(ffi.NbppError) as err:
        lib.accumulate(bad, a, a, a, a, a.copy(), a.copy(), a.copy())
    assert err.value.code == -1
    import ctypes as C
    rc = lib.lib.nbpp_accumulate(C.byref(p), 8, None, None, None, None, None, None, None)
    assert rc == -1 and b"NULL" in lib.lib.nbpp_last_error()
    with pytest.raises(ffi.NbppError):
        lib.set_option(999, 1)


def test_zero_softening_coincident_and_self(lib, oracle):
    """SOFTENING_SQUARED == 0: the self pair is excluded (src/particles.cpp:202,302)."""
    pos = datasets.uniform_random(500)
    m = datasets.masses(500)
    p = Config(MESH_SIZE=8, N_PER_SIDE=8, USE_PM=False).pp_params()
    p.soft_sq = 0.0
    got = gpu_acc(lib, p, pos, m)
    assert np.all(np.isfinite(got[0]))
    assert normalised_error(got, oracle.accumulate(p, *pos, m)) < TOL


# --------------------------------------------------------------------------------------
# P3M short range
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", [ffi.KERNEL_P3M_CELLS, ffi.KERNEL_P3M_ALLPAIRS])
@pytest.mark.parametrize("n_side", [16, 24])
def test_p3m_vs_oracle(lib, oracle, variant, n_side):
    pos = datasets.jittered_lattice(n_side)
    m = datasets.masses(n_side ** 3)
    p = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True).pp_params()
    want = oracle.accumulate(p, *pos, m, traversal=O.EXACT_CELLS)
    lib.set_option(ffi.OPT_FORCE_KERNEL, variant)
    try:
        got = gpu_acc(lib, p, pos, m)
        st = lib.stats()
    finally:
        lib.set_option(ffi.OPT_FORCE_KERNEL, 0)
    assert st["kernel_variant"] == variant
    assert normalised_error(got, want) < TOL
    if variant == ffi.KERNEL_P3M_CELLS:
        assert st["pairs_accepted"] == oracle.count_within_cutoff(p, *pos)  # exact integer


@pytest.mark.parametrize("radius", [1, 2, 3, 4])
def test_p3m_search_grid_choices(lib, oracle, radius):
    pos = datasets.clustered(20000)
    m = datasets.masses(20000, equal=False)
    p = Config(MESH_SIZE=32, N_PER_SIDE=27, USE_PM=True).pp_params()
    want = oracle.accumulate(p, *pos, m, traversal=O.EXACT_CELLS)
    lib.set_option(ffi.OPT_P3M_SEARCH_RADIUS, radius)
    try:
        got = gpu_acc(lib, p, pos, m)
        st = lib.stats()
    finally:
        lib.set_option(ffi.OPT_P3M_SEARCH_RADIUS, 0)
    assert st["kernel_variant"] == ffi.KERNEL_P3M_CELLS
    assert normalised_error(got, want) < TOL
    assert st["pairs_accepted"] == oracle.count_within_cutoff(p, *pos)


def test_p3m_default_ini_32cubed(lib, oracle):
    """The shipped simulation.ini geometry: 32^3 particles, mesh 32, cutoff 2.5 cells."""
    ref = json.load(open(os.path.join(GOLDEN, "reference_config.json")))["simulation_ini"]
    pos = datasets.jittered_lattice(32)
    m = np.full(32768, ref["dm_particle_mass"])
    config = Config(OMEGA_BARYON=ref["omega_baryon"])
    assert config.G == ref["G"] and config.SOFTENING_SQUARED == ref["softening_squared"]
    p = config.pp_params()
    want = oracle.accumulate(p, *pos, m, traversal=O.REF_CPU)  # the reference's CPU branch
    got = gpu_acc(lib, p, pos, m)
    assert normalised_error(got, want) < TOL
    per = lib.stats()["pairs_accepted"] / 32768
    assert 60 < per < 80


def test_p3m_wide_taper_other_box(lib, oracle):
    ref = json.load(open(os.path.join(GOLDEN, "reference_config.json")))["box10_wide"]
    pos = datasets.clustered(9000, L=10.0)
    m = datasets.masses(9000, equal=False)
    config = Config(DOMAIN_SIZE=10.0, MESH_SIZE=20, N_PER_SIDE=12, CUTOFF_RADIUS_CELLS=3.0,
                    CUTOFF_TRANSITION_WIDTH_FACTOR=0.35)
    assert config.R_SWITCH_START == ref["r_switch_start"]
    p = config.pp_params()
    assert normalised_error(gpu_acc(lib, p, pos, m),
                            oracle.accumulate(p, *pos, m, traversal=O.EXACT_CELLS)) < TOL


# --------------------------------------------------------------------------------------
# device-resident entry points
# --------------------------------------------------------------------------------------
def _t(arrs):
    import torch
    return [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in arrs]


@pytest.mark.parametrize("use_pm", [False, True])
def test_device_entry_target_slices(lib, oracle, use_pm):
    """The i-partition of a multi-GPU run: all sources, a slice of targets."""
    import torch
    n = 6000
    pos = datasets.clustered(n)
    m = datasets.masses(n, equal=False)
    p = Config(MESH_SIZE=24, N_PER_SIDE=18, USE_PM=use_pm).pp_params()
    want = oracle.accumulate(p, *pos, m)
    x, y, z, mm = _t(pos + [m])
    pieces = [[], [], []]
    for begin, count in ((0, 1500), (1500, 1), (1501, 2999), (4500, 1500)):
        acc = [torch.zeros(count, dtype=torch.float64, device="cuda") for _ in range(3)]
        lib.accumulate_device(p, x, y, z, mm, *acc, tgt_begin=begin, n_tgt=count)
        torch.cuda.synchronize()
        for c in range(3):
            pieces[c].append(acc[c].cpu().numpy())
    got = [np.concatenate(pc) for pc in pieces]
    assert normalised_error(got, want) < TOL


def test_add_device_is_one_rounding_per_component(lib):
    """nbpp_add_device: a += b, what compute_pp_forces ends with (src/particles.cpp:362-364)."""
    import torch
    rng = np.random.default_rng(3)
    a = [rng.standard_normal(5001) * 10.0 ** rng.integers(-8, 8, 5001) for _ in range(3)]
    b = [rng.standard_normal(5001) * 10.0 ** rng.integers(-8, 8, 5001) for _ in range(3)]
    ta, tb = _t(a), _t(b)
    lib.add_device(ta, tb)
    torch.cuda.synchronize()
    for c in range(3):
        assert np.array_equal(ta[c].cpu().numpy(), a[c] + b[c])
        assert np.array_equal(tb[c].cpu().numpy(), b[c])
    lib.add_device(ta, tb, clear=True)     # nbpp_add_clear_device: the same sum, then b = 0
    torch.cuda.synchronize()
    for c in range(3):
        assert np.array_equal(ta[c].cpu().numpy(), (a[c] + b[c]) + b[c])
        assert not tb[c].cpu().numpy().any()


def test_kick_drift_max_exact(lib, oracle):
    import torch
    rng = np.random.default_rng(2)
    n = 100003
    v = [rng.standard_normal(n) for _ in range(3)]
    a = [rng.standard_normal(n) * 1e-2 for _ in range(3)]
    x = [rng.random(n) for _ in range(3)]
    tv, ta, tx = _t(v), _t(a), _t(x)
    dt, sc, H = 4.3e-3, 0.02, 235.7
    lib.kick_device(tv, ta, dt / 2, sc, H)
    lib.drift_device(tx, tv, dt, 1.0)
    mx = lib.max_accel_sq_device(ta)
    torch.cuda.synchronize()
    oracle.kick(v, a, dt / 2, sc, H)
    oracle.drift(x, v, dt, 1.0)
    for c in range(3):
        assert np.array_equal(tv[c].cpu().numpy(), v[c])
        assert np.array_equal(tx[c].cpu().numpy(), x[c])
    assert mx == oracle.max_accel_sq(a)
    tiny = _t([np.full(10, 1e-6)] * 3)
    assert lib.max_accel_sq_device(tiny) == 1e-9


def test_fused_kick_drift_and_add_clear_kick_equal_the_separate_calls(lib, oracle):
    """nbpp_kick_drift_device and nbpp_add_clear_kick_device: one pass over the particles each,
    the same operations per element in the same order as the separate calls (and as the oracle's
    kick and drift, src/integrator.cpp:168-198): equal bit for bit."""
    import torch
    rng = np.random.default_rng(12)
    n = 70001
    v = [rng.standard_normal(n) for _ in range(3)]
    a = [rng.standard_normal(n) * 1e-2 for _ in range(3)]
    b = [rng.standard_normal(n) * 1e-3 for _ in range(3)]
    x = [rng.random(n) * 2.5 for _ in range(3)]
    dt, sc, H, L = 3.1e-3, 0.05, 120.3, 2.5
    tx, tv, ta = _t(x), _t(v), _t(a)
    lib.kick_drift_device(tx, tv, ta, dt / 2, dt, sc, H, L)
    torch.cuda.synchronize()
    oracle.kick(v, a, dt / 2, sc, H)
    oracle.drift(x, v, dt, L)
    for c in range(3):
        assert np.array_equal(tv[c].cpu().numpy(), v[c])
        assert np.array_equal(tx[c].cpu().numpy(), x[c])
    tb = _t(b)
    lib.add_clear_kick_device(ta, tb, tv, dt / 2, sc, H)
    torch.cuda.synchronize()
    s = [a[c] + b[c] for c in range(3)]
    oracle.kick(v, s, dt / 2, sc, H)
    for c in range(3):
        assert np.array_equal(ta[c].cpu().numpy(), s[c])
        assert not tb[c].cpu().numpy().any()
        assert np.array_equal(tv[c].cpu().numpy(), v[c])
    # ... and with max |a|^2 of the sum read back in the same pass: the value of the separate call
    b2 = [rng.standard_normal(n) * 1e-3 for _ in range(3)]
    tb = _t(b2)
    got_max = lib.add_clear_kick_max_device(ta, tb, tv, dt / 2, sc, H)
    s2 = [s[c] + b2[c] for c in range(3)]
    oracle.kick(v, s2, dt / 2, sc, H)
    for c in range(3):
        assert np.array_equal(ta[c].cpu().numpy(), s2[c])
        assert not tb[c].cpu().numpy().any()
        assert np.array_equal(tv[c].cpu().numpy(), v[c])
    assert got_max == lib.max_accel_sq_device(ta) == oracle.max_accel_sq(s2)


# --------------------------------------------------------------------------------------
# full-size properties (the 128^3 headline size, sampled)
# --------------------------------------------------------------------------------------
def test_128cubed_sampled_targets_and_momentum(lib, oracle):
    """N = 128^3 direct sum on the GPU: 48 sampled targets against the oracle's exact sum over
    all 2.1 M sources, and total momentum change sum_i m_i a_i == 0.  (The kernel picked at this
    size evaluates every unordered pair once and applies it to both members, so the momentum
    check verifies that every pair really reaches both; the sampled targets check the values.)"""
    import torch
    n_side = 128
    n = n_side ** 3
    pos = datasets.jittered_lattice(n_side)
    m = datasets.masses(n)
    p = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=False).pp_params()
    x, y, z, mm = _t(pos + [m])
    acc = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(3)]
    lib.accumulate_device(p, x, y, z, mm, *acc)
    torch.cuda.synchronize()
    got = [a.cpu().numpy() for a in acc]
    rng = np.random.default_rng(4)
    sample = np.sort(rng.choice(n, 48, replace=False))
    want = [np.zeros(48) for _ in range(3)]
    for k, i in enumerate(sample):  # one target at a time through the slice interface
        w = oracle.accumulate(p, *pos, m, tgt_begin=int(i), n_tgt=1)
        for c in range(3):
            want[c][k] = w[c][0]
    assert normalised_error([g[sample] for g in got], want) < TOL
    typical = np.mean(np.sqrt(got[0] ** 2 + got[1] ** 2 + got[2] ** 2)) * m[0] * n
    for c in range(3):
        assert abs(np.sum(m * got[c])) < 1e-9 * typical


@pytest.mark.parametrize("n_side,shift", [(40, 0.0), (40, 0.37), (64, 0.9)])
def test_p3m_target_slabs_bin_only_their_slab(lib, oracle, n_side, shift):
    """One rank's share of a multi-GPU P3M run: targets that are a contiguous index slice of a
    lattice-ordered set occupy a slab in x, and only that slab plus a halo is binned.  Four and
    seven slabs of a jittered lattice whose index order starts `shift` of the way through the box
    (so that one slab straddles x = L), 64^3 with the two-cell search grid: forces of all slabs
    against the oracle, accepted pairs of the slabs adding up to the count of the whole, and a
    slab of a single lattice plane."""
    import torch
    n = n_side ** 3
    pos = datasets.jittered_lattice(n_side)
    roll = int(shift * n_side) * n_side * n_side               # whole x planes
    pos = [np.ascontiguousarray(np.roll(c, -roll)) for c in pos]
    config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True)
    m = np.full(n, config.DM_PARTICLE_MASS)
    p = config.pp_params()
    want = oracle.accumulate(p, *pos, m, traversal=O.EXACT_CELLS)
    total = oracle.count_within_cutoff(p, *pos)
    x, y, z, mm = _t(pos + [m])
    for world in (4, 7):
        bounds = [r * n // world for r in range(world + 1)]
        pieces, counted = [[], [], []], 0
        for b, e in zip(bounds, bounds[1:]):
            acc = [torch.zeros(e - b, dtype=torch.float64, device="cuda") for _ in range(3)]
            lib.accumulate_device(p, x, y, z, mm, *acc, tgt_begin=b, n_tgt=e - b)
            torch.cuda.synchronize()
            assert lib.stats()["kernel_variant"] == ffi.KERNEL_P3M_CELLS
            counted += lib.stats()["pairs_accepted"]
            for c in range(3):
                pieces[c].append(acc[c].cpu().numpy())
        assert counted == total
        assert normalised_error([np.concatenate(pc) for pc in pieces], want) < TOL
    plane = n_side * n_side
    b = 3 * plane
    acc = [torch.zeros(plane, dtype=torch.float64, device="cuda") for _ in range(3)]
    lib.accumulate_device(p, x, y, z, mm, *acc, tgt_begin=b, n_tgt=plane)
    torch.cuda.synchronize()
    assert normalised_error([a.cpu().numpy() for a in acc], [w[b:b + plane] for w in want]) < TOL


def test_p3m_256cubed_sampled_targets_counts_and_momentum(lib, oracle):
    """N = 256^3 P3M short range with the mesh force on top (the whole force phase at the
    headline P3M size): sampled target slices against the oracle's sum over all 16.8 M sources,
    accepted-pair counts of the slices exactly, and zero net force on the system from either
    part."""
    import torch
    n_side = 256
    n = n_side ** 3
    pos = datasets.jittered_lattice(n_side)
    config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True)
    m = np.full(n, config.DM_PARTICLE_MASS)
    p = config.pp_params()
    x, y, z, mm = _t(pos + [m])
    acc = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(3)]
    lib.accumulate_device(p, x, y, z, mm, *acc)
    torch.cuda.synchronize()
    assert lib.stats()["kernel_variant"] == ffi.KERNEL_P3M_CELLS
    total_accepted = lib.stats()["pairs_accepted"]
    got = [a.cpu().numpy() for a in acc]
    typical = float(np.mean(np.sqrt(got[0] ** 2 + got[1] ** 2 + got[2] ** 2))) * m[0] * n
    for c in range(3):
        assert abs(np.sum(m * got[c])) < 1e-9 * typical
    rng = np.random.default_rng(9)
    counted = 0
    for begin in rng.choice(n - 16, 6, replace=False):     # six slices of 16 consecutive targets
        begin = int(begin)
        want = oracle.accumulate(p, *pos, m, tgt_begin=begin, n_tgt=16)   # every j, cutoff applied
        assert normalised_error([g[begin:begin + 16] for g in got], want) < TOL
        part = [torch.zeros(16, dtype=torch.float64, device="cuda") for _ in range(3)]
        lib.accumulate_device(p, x, y, z, mm, *part, tgt_begin=begin, n_tgt=16)
        torch.cuda.synchronize()
        assert lib.stats()["pairs_accepted"] == oracle.count_within_cutoff(p, *pos, tgt_begin=begin, n_tgt=16)
        counted += lib.stats()["pairs_accepted"]
    assert 55 * n < total_accepted < 75 * n and 55 * 96 < counted < 75 * 96
    # the mesh force on the same particles: no net force either
    mesh = ffi.load_mesh(device=0)
    macc = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(3)]
    mesh.accelerations_device(config.mesh_params(), x, y, z, mm, *macc)
    torch.cuda.synchronize()
    for a in macc:
        s_ = float((a * mm).sum().cpu())
        assert abs(s_) < 1e-9 * float((a.abs() * mm).sum().cpu())


@pytest.mark.parametrize("n_side", [407, 512])
def test_p3m_beyond_2_to_26_sources(lib, oracle, n_side):
    """407^3 = 67.4 M particles: more than 2^26, past what a packed index-plus-image-code entry
    could address.  512^3 = 2^27 particles: with their ghosts more than 2^27 records, which the
    kernel addresses through 64-bit pointers instead of 32-bit byte offsets (its WIDE variant).
    Sampled target slices against the oracle, counts exactly."""
    import torch
    n = n_side ** 3
    assert n > 1 << 26
    pos = datasets.jittered_lattice(n_side)
    config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True)
    m = np.full(n, config.DM_PARTICLE_MASS)
    p = config.pp_params()
    x, y, z, mm = _t(pos + [m])
    rng = np.random.default_rng(12)
    for begin in [0, n - 8] + [int(b) for b in rng.choice(n - 8, 2, replace=False)]:
        part = [torch.zeros(8, dtype=torch.float64, device="cuda") for _ in range(3)]
        lib.accumulate_device(p, x, y, z, mm, *part, tgt_begin=begin, n_tgt=8)
        torch.cuda.synchronize()
        assert lib.stats()["kernel_variant"] == ffi.KERNEL_P3M_CELLS
        want = oracle.accumulate(p, *pos, m, tgt_begin=begin, n_tgt=8)
        assert normalised_error([t.cpu().numpy() for t in part], want) < TOL
        assert lib.stats()["pairs_accepted"] == oracle.count_within_cutoff(p, *pos, tgt_begin=begin, n_tgt=8)
    acc = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(3)]
    lib.accumulate_device(p, x, y, z, mm, *acc)
    torch.cuda.synchronize()
    assert 55 * n < lib.stats()["pairs_accepted"] < 75 * n
    for a in acc:
        assert abs(float((a * mm).sum().cpu())) < 1e-9 * float((a.abs() * mm).sum().cpu())


# --------------------------------------------------------------------------------------
# Zel'dovich initial conditions (the particle sets the reference actually starts from)
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_side", [16, 32])
@pytest.mark.parametrize("use_pm", [False, True])
def test_zeldovich_ics(lib, oracle, n_side, use_pm):
    pos = datasets.zeldovich(n_side)
    config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=use_pm, OMEGA_BARYON=0.048)
    m = np.full(n_side ** 3, config.DM_PARTICLE_MASS)
    p = config.pp_params()
    # P3M: the reference's own CPU-branch traversal (exact at mesh >= 7, <= 1500 neighbours);
    # PP-only: the exact sum (what the reference's offload branch gives on an odd mesh)
    want = oracle.accumulate(p, *pos, m, traversal=O.REF_CPU if use_pm else O.EXACT)
    got = gpu_acc(lib, p, pos, m)
    assert normalised_error(got, want) < TOL


def test_default_ini_total_acceleration_matches_reference_output(lib, oracle):
    """GPU short-range force + restated mesh force on the shipped simulation.ini against the
    values a build of the reference printed (tests/golden/reference_kat.json)."""
    from oracle.pm_mesh import initial_mesh_accelerations
    from oracle.zeldovich import zeldovich_particles
    kat = json.load(open(os.path.join(GOLDEN, "reference_kat.json")))["default_ini_anchor"]
    cfg = json.load(open(os.path.join(GOLDEN, "reference_config.json")))["simulation_ini"]
    pos, _, _ = zeldovich_particles(oracle=oracle)
    m = np.full(32768, cfg["dm_particle_mass"])
    acc = initial_mesh_accelerations(pos, m, 32, 1.0, cfg["G"], cfg["gas_total_mass"])
    config = Config(OMEGA_BARYON=cfg["omega_baryon"])
    lib.accumulate(config.pp_params(), *pos, m, *acc)   # acc += PP, as src/ics.cpp:319-322
    for c in range(3):
        assert acc[c][0] == pytest.approx(kat["p0_acc_initial_pm_plus_pp"][c], rel=1e-12)
    assert float(np.max(acc[0] ** 2 + acc[1] ** 2 + acc[2] ** 2)) == \
        pytest.approx(kat["max_accel_sq"], rel=1e-12)


# --------------------------------------------------------------------------------------
# stress: separations at and around half a box, extreme clustering
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("variant", [0, ffi.KERNEL_DIRECT, ffi.KERNEL_DIRECT_SYMMETRIC])
def test_half_box_separated_clusters(lib, oracle, seed, variant):
    """Tight clusters whose mutual separations sit at L/2 +- a few cluster radii on random axes,
    in boxes of assorted lengths: every tile pair is a uniform-shift, straddle or generic case
    decided by a few ulps of the bounding boxes."""
    rng = np.random.default_rng(100 + seed)
    L = [1.0, 2.0, 0.75, 10.0, 64.0, 3.3][seed]
    n_clusters, per = 6, 700
    centres = rng.random((n_clusters, 3)) * L
    for c in range(1, n_clusters):       # place each centre half a box from the previous one
        axes = rng.random(3) < 0.6
        centres[c] = np.where(axes, centres[c - 1] + 0.5 * L + rng.normal(0, 2e-3 * L, 3), centres[c])
    pts = (centres[:, None, :] + rng.normal(0, 1.5e-3 * L, (n_clusters, per, 3))).reshape(-1, 3)
    pts[:50] = np.round(pts[:50] / (L / 64)) * (L / 64)      # some exactly on a 1/64 lattice
    pts = np.mod(pts, L)
    pos = [np.ascontiguousarray(pts[:, c]) for c in range(3)]
    n = pts.shape[0]
    m = datasets.masses(n, equal=bool(seed % 2))
    p = Config(DOMAIN_SIZE=L, MESH_SIZE=16, N_PER_SIDE=16, USE_PM=False).pp_params()
    want = oracle.accumulate(p, *pos, m)
    lib.set_option(ffi.OPT_FORCE_KERNEL, variant)
    try:
        got = gpu_acc(lib, p, pos, m)
    finally:
        lib.set_option(ffi.OPT_FORCE_KERNEL, 0)
    assert normalised_error(got, want) < TOL


def test_p3m_extreme_clustering(lib, oracle):
    """Thousands of particles inside one search cell (late-time haloes): long runs, queue drains
    in the middle of a run, heavy per-cell ordering."""
    rng = np.random.default_rng(5)
    blob = 0.37 + 2e-3 * rng.random((3000, 3))
    edge = np.mod(0.999 + 4e-3 * rng.random((1500, 3)), 1.0)     # a halo across the box corner
    bg = rng.random((6000, 3))
    pts = np.concatenate([blob, edge, bg])
    pos = [np.ascontiguousarray(pts[:, c]) for c in range(3)]
    m = datasets.masses(pts.shape[0], equal=False)
    p = Config(MESH_SIZE=32, N_PER_SIDE=22, USE_PM=True).pp_params()
    want = oracle.accumulate(p, *pos, m, traversal=O.EXACT_CELLS)
    got = gpu_acc(lib, p, pos, m)
    assert lib.stats()["kernel_variant"] == ffi.KERNEL_P3M_CELLS
    assert normalised_error(got, want) < TOL
    assert lib.stats()["pairs_accepted"] == oracle.count_within_cutoff(p, *pos)


@pytest.mark.parametrize("blobs", [(5000, 2500), (12000, 300)])
@pytest.mark.parametrize("use_pm", [False, True])
def test_overfull_cells_are_ordered_deterministically(lib, oracle, use_pm, blobs):
    """Cells with thousands of members are sorted by a workgroup (in LDS up to 8192 members, by
    rank counting beyond) instead of by one thread or one wavefront; the sorted order, and with
    it every sum of the ordered kernels, must still be a pure function of the input: two
    evaluations agree bit for bit, and with the oracle to the usual bar."""
    rng = np.random.default_rng(77)
    pts = np.concatenate([0.21 + 1e-3 * rng.random((blobs[0], 3)), 0.8 + 3e-3 * rng.random((blobs[1], 3)),
                          rng.random((9000, 3))])
    pos = [np.ascontiguousarray(pts[:, c]) for c in range(3)]
    m = datasets.masses(pts.shape[0], equal=False)
    p = Config(MESH_SIZE=32, N_PER_SIDE=25, USE_PM=use_pm).pp_params()
    lib.set_option(ffi.OPT_DETERMINISTIC, 1)
    try:
        a = gpu_acc(lib, p, pos, m)
        b = gpu_acc(lib, p, pos, m)
    finally:
        lib.set_option(ffi.OPT_DETERMINISTIC, 0)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    want = oracle.accumulate(p, *pos, m, traversal=O.EXACT_CELLS if use_pm else O.EXACT)
    assert normalised_error(a, want) < TOL
    if use_pm:
        assert lib.stats()["pairs_accepted"] == oracle.count_within_cutoff(p, *pos)


@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16])
def test_p3m_lanes_per_target(lib, oracle, lanes):
    """The cell kernel with 1, 2, 4, 8 or 16 lanes sharing a target (automatic choice: more lanes the
    fewer targets): same accepted pairs, same forces, and a sub-range of targets."""
    pos = datasets.clustered(20000, seed=lanes)
    m = datasets.masses(20000, equal=False)
    p = Config(MESH_SIZE=32, N_PER_SIDE=27, USE_PM=True).pp_params()
    want = oracle.accumulate(p, *pos, m, traversal=O.EXACT_CELLS)
    lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, lanes)
    try:
        got = gpu_acc(lib, p, pos, m)
        assert lib.stats()["kernel_variant"] == ffi.KERNEL_P3M_CELLS
        assert lib.stats()["pairs_accepted"] == oracle.count_within_cutoff(p, *pos)
        assert normalised_error(got, want) < TOL
        import torch
        d = [torch.as_tensor(a, device="cuda:0") for a in pos + [m]]
        part = [torch.zeros(5001, dtype=torch.float64, device="cuda:0") for _ in range(3)]
        lib.accumulate_device(p, *d, *part, tgt_begin=7003, n_tgt=5001)
        torch.cuda.synchronize()
        sub = [w[7003:7003 + 5001] for w in want]
        assert normalised_error([t.cpu().numpy() for t in part], sub) < TOL
    finally:
        lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, 0)


@pytest.mark.parametrize("seed", range(64))
def test_randomised_configurations(lib, oracle, seed):
    """Seeded sweep over what a caller can vary at once: particle count (from 1 up), box length,
    mesh size, cutoff radius and taper width, zero or default softening, equal or unequal masses,
    particle distribution, direct or P3M mode, and a random target slice through the device
    entry point.  Every case is compared with the oracle's exact sum and, in P3M mode, its
    accepted-pair count."""
    import torch
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 3, 63, 64, 65, 255, 257, 1000, 2047, 2048, 2049, 4100, 6000]))
    L = float(rng.choice([1.0, 0.5, 7.0, 100.0]))
    mesh = int(rng.integers(4, 41))
    use_pm = bool(rng.integers(0, 2))
    config = Config(DOMAIN_SIZE=L, MESH_SIZE=mesh, N_PER_SIDE=max(1, round(n ** (1 / 3))), USE_PM=use_pm,
                    CUTOFF_RADIUS_CELLS=float(rng.uniform(1.0, 4.0)),
                    CUTOFF_TRANSITION_WIDTH_FACTOR=float(rng.uniform(0.05, 0.9)))
    kind = int(rng.integers(0, 3))
    if rng.random() < 0.3 and kind != 2:   # coincident particles without softening: NaN in the reference
        config.SOFTENING_SQUARED = 0.0
    if kind == 0:
        pos = [c * L for c in datasets.uniform_random(n, seed=seed)]
    elif kind == 1:
        pos = datasets.clustered(n, L=L, seed=seed)
    else:
        pts = (rng.integers(0, 8, (n, 3)) + 0.5) / 8 * L           # many coincident positions
        pos = [np.ascontiguousarray(pts[:, c]) for c in range(3)]
    m = datasets.masses(n, equal=bool(rng.integers(0, 2)))
    p = config.pp_params()
    traversal = O.EXACT
    want = oracle.accumulate(p, *pos, m, traversal=traversal)
    got = gpu_acc(lib, p, pos, m)
    floor = pair_force_scale(p, m, max(n, 8)) * 1e-3
    assert normalised_error(got, want, floor=floor) < TOL
    if lib.stats()["kernel_variant"] == ffi.KERNEL_P3M_CELLS:   # the all-pairs fallback does not count
        assert lib.stats()["pairs_accepted"] == oracle.count_within_cutoff(p, *pos)
    begin = int(rng.integers(0, n))
    count = int(rng.integers(1, n - begin + 1))
    d = [torch.as_tensor(np.ascontiguousarray(a), device="cuda:0") for a in pos + [m]]
    part = [torch.zeros(count, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    lib.accumulate_device(p, *d, *part, tgt_begin=begin, n_tgt=count)
    torch.cuda.synchronize()
    sub = [w[begin:begin + count] for w in want]
    assert normalised_error([t.cpu().numpy() for t in part], sub, floor=floor) < TOL


# --------------------------------------------------------------------------------------
# where the outputs live: the pair-once kernel's fp64 atomics need ordinary device memory
# --------------------------------------------------------------------------------------
def _raw_call(lib, fn, p, n, x, y, z, m, extra, outs):
    """An entry point called with bare addresses (page-locked host tensors are not CUDA tensors)."""
    import ctypes as C
    DP = C.POINTER(C.c_double)
    ptr = lambda t: C.cast(C.c_void_p(t.data_ptr()), DP)          # noqa: E731
    return fn(C.byref(p), n, ptr(x), ptr(y), ptr(z), ptr(m), *extra, *[ptr(o) for o in outs], C.c_void_p(0))


def test_host_mapped_outputs_never_get_the_atomic_kernel(lib, oracle):
    import torch
    n_side = 26                                                  # 17 576 >= 2^14: the pair-once kernel's range
    n = n_side ** 3
    pos = datasets.jittered_lattice(n_side)
    m = datasets.masses(n)
    p = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=False).pp_params()
    x, y, z, mm = (torch.as_tensor(a, device="cuda:0") for a in (*pos, m))
    outs = [torch.zeros(n, dtype=torch.float64).pin_memory() for _ in range(3)]    # hipHostMalloc
    rc = _raw_call(lib, lib.lib.nbpp_accumulate_device, p, n, x, y, z, mm, (0, n), outs)
    assert rc == 0, lib.lib.nbpp_last_error()
    torch.cuda.synchronize()
    assert lib.stats()["kernel_variant"] == ffi.KERNEL_DIRECT    # not KERNEL_DIRECT_SYMMETRIC
    dev_outs = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    lib.accumulate_device(p, x, y, z, mm, *dev_outs)
    assert lib.stats()["kernel_variant"] == ffi.KERNEL_DIRECT_SYMMETRIC
    want = oracle.accumulate(p, *pos, m, tgt_begin=0, n_tgt=512)
    assert normalised_error([o.numpy()[:512] for o in outs], want) < TOL
    assert normalised_error([o.cpu().numpy()[:512] for o in dev_outs], want) < TOL
    # the entry point that is nothing but that kernel refuses such outputs
    rc = _raw_call(lib, lib.lib.nbpp_accumulate_pairs_device, p, n, x, y, z, mm, (0, 1), outs)
    assert rc == -1 and b"ordinary device memory" in lib.lib.nbpp_last_error()
    lib.set_option(ffi.OPT_FORCE_KERNEL, ffi.KERNEL_DIRECT_SYMMETRIC)
    try:
        rc = _raw_call(lib, lib.lib.nbpp_accumulate_device, p, n, x, y, z, mm, (0, n), outs)
        assert rc == -1
    finally:
        lib.set_option(ffi.OPT_FORCE_KERNEL, 0)


@pytest.mark.parametrize("use_pm", [False, True])
def test_page_locked_caller_arrays_give_the_same_sums(lib, use_pm):
    """nbpp_accumulate from page-locked arrays (nbpp_host_alloc: DMA in, sums added in place by
    the device) against the same call on pageable arrays (staged, added by host threads): the
    same kernels and one rounding per accumulator either way, so bit for bit (with reproducible
    sums asked for: by default the order of a cell's members, and with it the last bits of the
    short-range sums, is whatever the binning's atomics left)."""
    n_side = 20
    n = n_side ** 3
    pos = datasets.jittered_lattice(n_side, seed=4)
    m = datasets.masses(n, equal=False)
    p = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=use_pm).pp_params()
    start = [np.full(n, 0.25), np.full(n, -1.0), np.zeros(n)]
    lib.set_option(ffi.OPT_DETERMINISTIC, 1)
    arrays = [lib.host_alloc(n) for _ in range(7)]
    try:
        plain = gpu_acc(lib, p, pos, m, acc0=start)
        for dst, src in zip(arrays, (*pos, m, *start)):
            dst[:] = src
        lib.accumulate(p, *arrays)
        for got, want in zip(arrays[4:], plain):
            assert np.array_equal(got, want)
    finally:
        lib.set_option(ffi.OPT_DETERMINISTIC, 0)
        for a in arrays:
            lib.host_free(a)


def test_zero_transition_width_is_accepted(lib, oracle):
    """cutoff_transition_width_factor = 0: r_switch equals the cutoff radius, so no accepted pair
    reaches the taper (src/particles.cpp:331-335); the reference accepts the configuration."""
    n_side = 16
    pos = datasets.jittered_lattice(n_side, jitter=0.9)
    m = datasets.masses(n_side ** 3)
    p = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True, CUTOFF_TRANSITION_WIDTH_FACTOR=0.0).pp_params()
    assert p.trans_width == 0.0 and p.r_switch_sq == p.cutoff_sq
    want = oracle.accumulate(p, *pos, m, traversal=O.EXACT_CELLS)
    got = gpu_acc(lib, p, pos, m)
    assert lib.stats()["kernel_variant"] == ffi.KERNEL_P3M_CELLS
    assert all(np.all(np.isfinite(g)) for g in got)
    assert normalised_error(got, want) < TOL


# --------------------------------------------------------------------------------------
# P3M cell kernel: survivors pooled per wavefront against every lane draining its own
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,n_side", [("lattice", 48), ("clustered", 40), ("lattice_soft0", 32)])
def test_p3m_pooled_drain_equals_per_lane_drain(lib, oracle, name, n_side):
    """k_p3m_pool against k_p3m_cells on the same inputs: the accepted-pair count is an exact
    integer and must be identical (and equal to the oracle's); the sums differ only in the order
    of additions."""
    n = n_side ** 3
    pos = datasets.clustered(n, sigma=0.05) if name == "clustered" else datasets.jittered_lattice(n_side, jitter=0.9)
    m = datasets.masses(n, equal=(name != "clustered"))
    config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True)
    if name == "lattice_soft0":
        config.SOFTENING_SQUARED = 0.0
    p = config.pp_params()
    out = {}
    try:
        lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, 1)
        lib.set_option(ffi.OPT_P3M_PAIR_ONCE, -1)
        for pooled in (-1, 1):
            lib.set_option(ffi.OPT_P3M_POOLED_DRAIN, pooled)
            out[pooled] = gpu_acc(lib, p, pos, m)
            st = lib.stats()
            assert st["kernel_variant"] == ffi.KERNEL_P3M_CELLS and st["p3m_pooled"] == (1 if pooled > 0 else 0)
            out[pooled, "accepted"] = st["pairs_accepted"]
    finally:
        lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, 0)
        lib.set_option(ffi.OPT_P3M_POOLED_DRAIN, 0)
        lib.set_option(ffi.OPT_P3M_PAIR_ONCE, 0)
    assert out[1, "accepted"] == out[-1, "accepted"] == oracle.count_within_cutoff(p, *pos)
    floor = pair_force_scale(p, m, n) * 1e-3
    assert normalised_error(out[1], out[-1], floor) < 1e-12
    n_tgt = 4096
    want = oracle.accumulate(p, *pos, m, tgt_begin=0, n_tgt=n_tgt, traversal=O.EXACT_CELLS)
    assert normalised_error([a[:n_tgt] for a in out[1]], want, floor) < TOL


@pytest.mark.parametrize("row_lanes", [4, 8, 16, 32, 64])
@pytest.mark.parametrize("name", ["clustered", "lattice_soft0"])
def test_p3m_lanes_walking_rows_together(lib, oracle, name, row_lanes):
    """k_p3m_rows (NBPP_OPT_P3M_ROW_LANES): 4 to 64 lanes of a target take every G-th candidate of
    each of its rows.  Exact accepted-pair count against the oracle, sums within the parity bar
    (ragged target count, unequal masses, blobs whose cells hold hundreds of particles; zero
    softening on a lattice), and - with the members of a cell in a fixed order
    (NBPP_OPT_DETERMINISTIC) - the same bits from one call to the next."""
    n_side = 28
    n = n_side ** 3 - 11
    if name == "clustered":
        pos = datasets.clustered(n, n_blobs=9, sigma=0.02, seed=5)
        m = datasets.masses(n, equal=False)
    else:
        pos = [c[:n].copy() for c in datasets.jittered_lattice(n_side, jitter=0.9)]
        m = datasets.masses(n, equal=True)
    config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True)
    if name == "lattice_soft0":
        config.SOFTENING_SQUARED = 0.0
    p = config.pp_params()
    try:
        lib.set_option(ffi.OPT_P3M_ROW_LANES, row_lanes)
        lib.set_option(ffi.OPT_DETERMINISTIC, 1)
        got = gpu_acc(lib, p, pos, m)
        st = lib.stats()
        again = gpu_acc(lib, p, pos, m)
        with pytest.raises(ffi.NbppError):
            lib.set_option(ffi.OPT_P3M_ROW_LANES, 12)
    finally:
        lib.set_option(ffi.OPT_P3M_ROW_LANES, 0)
        lib.set_option(ffi.OPT_DETERMINISTIC, 0)
    assert st["kernel_variant"] == ffi.KERNEL_P3M_CELLS and st["p3m_pooled"] == 3
    assert st["pairs_accepted"] == oracle.count_within_cutoff(p, *pos)
    for c in range(3):
        assert np.array_equal(got[c], again[c])
    n_tgt = 3000
    want = oracle.accumulate(p, *pos, m, tgt_begin=0, n_tgt=n_tgt, traversal=O.EXACT_CELLS)
    assert normalised_error([a[:n_tgt] for a in got], want, pair_force_scale(p, m, n) * 1e-3) < TOL


@pytest.mark.parametrize("row_lanes", [2, 4, 8, 16])
@pytest.mark.parametrize("name", ["clustered", "lattice_soft0"])
def test_p3m_row_walk_with_pooled_drain(lib, oracle, name, row_lanes):
    """k_p3m_rows_pool (NBPP_OPT_P3M_ROWS_POOLED = 1 with 2 to 16 lanes per target): the row walk
    with a single-precision test, survivors pooled per wavefront and evaluated 64 at a time.  Exact
    accepted-pair count against the oracle (every target also meets itself once, which is taken out
    again), sums within the parity bar; ragged target count, unequal masses, cells holding hundreds
    of particles; zero softening on a lattice."""
    n_side = 28
    n = n_side ** 3 - 11
    if name == "clustered":
        pos = datasets.clustered(n, n_blobs=9, sigma=0.02, seed=5)
        m = datasets.masses(n, equal=False)
    else:
        pos = [c[:n].copy() for c in datasets.jittered_lattice(n_side, jitter=0.9)]
        m = datasets.masses(n, equal=True)
    config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True)
    if name == "lattice_soft0":
        config.SOFTENING_SQUARED = 0.0
    p = config.pp_params()
    try:
        lib.set_option(ffi.OPT_P3M_ROW_LANES, row_lanes)
        lib.set_option(ffi.OPT_P3M_ROWS_POOLED, 1)
        got = gpu_acc(lib, p, pos, m)
        st = lib.stats()
    finally:
        lib.set_option(ffi.OPT_P3M_ROW_LANES, 0)
        lib.set_option(ffi.OPT_P3M_ROWS_POOLED, 0)
    assert st["kernel_variant"] == ffi.KERNEL_P3M_CELLS and st["p3m_pooled"] == 4
    assert st["pairs_accepted"] == oracle.count_within_cutoff(p, *pos)
    n_tgt = 3000
    want = oracle.accumulate(p, *pos, m, tgt_begin=0, n_tgt=n_tgt, traversal=O.EXACT_CELLS)
    assert normalised_error([a[:n_tgt] for a in got], want, pair_force_scale(p, m, n) * 1e-3) < TOL


def test_large_clustering_set_gets_the_row_walk_with_pooled_drain(lib, oracle):
    """A set that fills the GPU with one lane per target and has begun to cluster (clustering
    measure above 2.7, no brick of cells the pair-once kernel could hold) is given k_p3m_rows_pool
    with two lanes per target; NBPP_OPT_P3M_ROWS_POOLED = -1 keeps the pooled kernel.  Same exact
    count, sums equal to rounding, sampled targets against the oracle."""
    import torch
    n_side = 112
    n = n_side ** 3
    pos = datasets.clustered(n, n_blobs=1500, sigma=0.02, seed=8)
    config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True)
    m = np.full(n, config.DM_PARTICLE_MASS)
    p = config.pp_params()
    dev = _t(pos + [m])
    out = {}
    try:
        for label, value in (("auto", 0), ("never", -1)):
            lib.set_option(ffi.OPT_P3M_ROWS_POOLED, value)
            lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, 0 if label == "auto" else 1)
            acc = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(3)]
            lib.accumulate_device(p, *dev, *acc)
            torch.cuda.synchronize()
            st = lib.stats()
            out[label] = ([a.cpu().numpy() for a in acc], st["p3m_pooled"], st["pairs_accepted"], st["cell_occupancy"])
    finally:
        lib.set_option(ffi.OPT_P3M_ROWS_POOLED, 0)
        lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, 0)
    occupancy = out["auto"][3]
    assert 2.7 <= occupancy < 64.0 and n >= 65536 * min(occupancy, 16.0), occupancy   # what the data set is meant to be
    assert out["auto"][1] == 4
    assert out["never"][1] == 1
    assert out["auto"][2] == out["never"][2]
    floor = pair_force_scale(p, m, n) * 1e-3
    assert normalised_error(out["auto"][0], out["never"][0], floor) < 1e-11
    begin = n // 3
    want = oracle.accumulate(p, *pos, m, tgt_begin=begin, n_tgt=16, traversal=O.EXACT_CELLS)
    assert normalised_error([g[begin:begin + 16] for g in out["auto"][0]], want, floor) < TOL


def test_small_sets_get_the_row_walking_kernel_by_themselves(lib, oracle):
    """A set small enough to get several lanes per target is given k_p3m_rows; a forced lane count
    (NBPP_OPT_P3M_LANES_PER_TARGET) or NBPP_OPT_P3M_ROW_LANES = -1 keeps k_p3m_cells."""
    n_side = 24
    pos = datasets.jittered_lattice(n_side)
    m = datasets.masses(n_side ** 3, equal=True)
    p = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True).pp_params()
    kinds = {}
    try:
        for label, key, value in (("auto", ffi.OPT_P3M_ROW_LANES, 0), ("never", ffi.OPT_P3M_ROW_LANES, -1),
                                  ("lanes", ffi.OPT_P3M_LANES_PER_TARGET, 4)):
            lib.set_option(key, value)
            kinds[label] = (gpu_acc(lib, p, pos, m), lib.stats()["p3m_pooled"])
            lib.set_option(key, 0)
    finally:
        lib.set_option(ffi.OPT_P3M_ROW_LANES, 0)
        lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, 0)
    assert kinds["auto"][1] == 3 and kinds["never"][1] == 0 and kinds["lanes"][1] == 0
    floor = pair_force_scale(p, m, n_side ** 3) * 1e-3
    assert normalised_error(kinds["auto"][0], kinds["never"][0], floor) < 1e-12


def test_heavily_clustered_large_set_with_the_row_walking_kernels(lib, oracle):
    """128^3 particles in 400 tight blobs (clustering measure in the hundreds): enough targets for
    one lane each.  The library's own choice is the row walk with the pooled drain; k_p3m_rows with
    4 lanes per target, asked for, takes the set as well.  The same exact accepted-pair count as
    the pooled kernel, sums equal to it to rounding, and sampled targets against the oracle."""
    import torch
    n_side = 128
    n = n_side ** 3
    pos = datasets.clustered(n, n_blobs=400, sigma=0.004, seed=3)
    config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True)
    m = np.full(n, config.DM_PARTICLE_MASS)
    p = config.pp_params()
    dev = _t(pos + [m])
    out = {}
    try:
        for label, lanes, rows, rows_pooled in (("auto", 0, 0, 0), ("rows", 0, 4, -1), ("pooled", 1, 0, 0)):
            lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, lanes)
            lib.set_option(ffi.OPT_P3M_ROW_LANES, rows)
            lib.set_option(ffi.OPT_P3M_ROWS_POOLED, rows_pooled)
            acc = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(3)]
            lib.accumulate_device(p, *dev, *acc)
            torch.cuda.synchronize()
            st = lib.stats()
            out[label] = ([a.cpu().numpy() for a in acc], st["p3m_pooled"], st["pairs_accepted"], st["cell_occupancy"])
    finally:
        lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, 0)
        lib.set_option(ffi.OPT_P3M_ROW_LANES, 0)
        lib.set_option(ffi.OPT_P3M_ROWS_POOLED, 0)
    assert out["auto"][1] == 4 and out["rows"][1] == 3 and out["pooled"][1] == 1 and out["auto"][3] > 100.0
    assert out["rows"][2] == out["pooled"][2]
    assert out["auto"][2] == out["pooled"][2] > 1000 * n
    floor = pair_force_scale(p, m, n) * 1e-3
    assert normalised_error(out["auto"][0], out["pooled"][0], floor) < 1e-11
    assert normalised_error(out["rows"][0], out["pooled"][0], floor) < 1e-11
    rng = np.random.default_rng(2)
    for begin in rng.choice(n - 8, 4, replace=False):
        begin = int(begin)
        want = oracle.accumulate(p, *pos, m, tgt_begin=begin, n_tgt=8, traversal=O.EXACT_CELLS)
        for label in ("auto", "rows"):
            assert normalised_error([g[begin:begin + 8] for g in out[label][0]], want, floor) < TOL


@pytest.mark.parametrize("threads", [64, 128, 256])
def test_p3m_pooled_kernel_workgroup_sizes(lib, oracle, threads):
    """NBPP_OPT_P3M_POOL_WORKGROUP: the pooled kernel with one, two and four wavefronts per
    workgroup (ragged target count, so the last workgroup is partly padding): the same exact
    accepted-pair count as the oracle, sums within the parity bar; other values are refused."""
    n_side = 36
    n = n_side ** 3 - 37
    pos = [c[:n].copy() for c in datasets.jittered_lattice(n_side, jitter=0.9)]
    m = datasets.masses(n, equal=False)
    p = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True).pp_params()
    try:
        lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, 1)
        lib.set_option(ffi.OPT_P3M_PAIR_ONCE, -1)
        lib.set_option(ffi.OPT_P3M_POOLED_DRAIN, 1)
        lib.set_option(ffi.OPT_P3M_POOL_WORKGROUP, threads)
        got = gpu_acc(lib, p, pos, m)
        st = lib.stats()
        with pytest.raises(ffi.NbppError):
            lib.set_option(ffi.OPT_P3M_POOL_WORKGROUP, 96)
    finally:
        for k in (ffi.OPT_P3M_LANES_PER_TARGET, ffi.OPT_P3M_PAIR_ONCE, ffi.OPT_P3M_POOLED_DRAIN,
                  ffi.OPT_P3M_POOL_WORKGROUP):
            lib.set_option(k, 0)
    assert st["p3m_pooled"] == 1
    assert st["pairs_accepted"] == oracle.count_within_cutoff(p, *pos)
    n_tgt = 2048
    want = oracle.accumulate(p, *pos, m, tgt_begin=0, n_tgt=n_tgt, traversal=O.EXACT_CELLS)
    assert normalised_error([a[:n_tgt] for a in got], want, pair_force_scale(p, m, n) * 1e-3) < TOL


@pytest.mark.parametrize("seed", range(40))
def test_p3m_kernels_agree_on_random_configurations(lib, oracle, seed):
    """The three P3M cell kernels (per-lane rings, reproducible; pooled survivors, eight
    candidates per trip, rows trimmed to the cutoff sphere; every pair once per brick) on seeded
    random set-ups: box size, mesh, cutoff radius in cells, search radius 2 or 3, transition
    width, softening (zero included), masses, and particle sets that put many particles exactly
    on cell faces and box faces, where the row trimming and the ghost cells have their edge
    cases.  The accepted-pair count must be the oracle's exact integer for every kernel; sums
    within the path's 1e-10 of the oracle on sampled targets and of each other everywhere."""
    rng = np.random.Generator(np.random.MT19937(1000 + seed))
    L = float(rng.choice([1.0, 3.7, 100.0]))
    mesh = int(rng.integers(10, 33))
    n = int(rng.integers(3000, 26000))
    config = Config(MESH_SIZE=mesh, N_PER_SIDE=max(2, round(n ** (1 / 3))), USE_PM=True, DOMAIN_SIZE=L,
                    CUTOFF_RADIUS_CELLS=float(rng.uniform(1.3, 3.2)),
                    CUTOFF_TRANSITION_WIDTH_FACTOR=float(rng.choice([0.0, 0.2, 0.5])))
    kind = seed % 3
    if seed % 4 == 0 and kind != 2:   # (kind 2 holds coincident particles: NaN in the reference without softening)
        config.SOFTENING_SQUARED = 0.0
    p = config.pp_params()
    if kind == 0:
        pos = [rng.random(n) * L for _ in range(3)]
    elif kind == 1:
        pos = [np.mod(c * L, L) for c in datasets.clustered(n, n_blobs=int(rng.integers(2, 9)), sigma=0.04, seed=seed)]
    else:
        # a lattice commensurate with a fine grid: many coordinates exactly on faces of search cells,
        # 0 included, plus a jittered remainder
        k = int(rng.integers(8, 25))
        grid = rng.integers(0, k, size=(3, n)) * (L / k)
        jitter = (rng.random((3, n)) < 0.5) * rng.random((3, n)) * (L / k)
        pos = [np.ascontiguousarray(np.mod(grid[c] + jitter[c], L)) for c in range(3)]
    m = datasets.masses(n, equal=bool(seed % 2))
    radius = 2 if seed % 5 else 3
    want_pairs = oracle.count_within_cutoff(p, *pos)
    n_tgt = min(n, 1500)
    want = oracle.accumulate(p, *pos, m, tgt_begin=0, n_tgt=n_tgt, traversal=O.EXACT_CELLS)
    floor = pair_force_scale(p, m, n) * 1e-3
    out = {}
    try:
        lib.set_option(ffi.OPT_FORCE_KERNEL, ffi.KERNEL_P3M_CELLS)
        lib.set_option(ffi.OPT_P3M_SEARCH_RADIUS, radius)
        lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, 1)
        for name, pooled, once in (("cells", -1, -1), ("pool", 1, -1), ("sym", -1, 1)):
            lib.set_option(ffi.OPT_P3M_POOLED_DRAIN, pooled)
            lib.set_option(ffi.OPT_P3M_PAIR_ONCE, once)
            try:
                got = gpu_acc(lib, p, pos, m)
            except ffi.NbppError as e:       # a box too small for the cell search at this radius
                assert "box too small" in str(e)
                pytest.skip(str(e))
            st = lib.stats()
            if st["kernel_variant"] != ffi.KERNEL_P3M_CELLS:
                pytest.skip("box too small for the cell search at this cutoff: the all-pairs kernel took the call")
            if name == "sym" and st["p3m_pooled"] != 2:
                continue                      # a brick region did not fit: the call fell back
            assert st["p3m_pooled"] == {"cells": 0, "pool": 1, "sym": 2}[name]
            assert st["pairs_accepted"] == want_pairs, name
            assert normalised_error([a[:n_tgt] for a in got], want, floor) < TOL, name
            out[name] = got
    finally:
        for k_ in (ffi.OPT_FORCE_KERNEL, ffi.OPT_P3M_SEARCH_RADIUS, ffi.OPT_P3M_LANES_PER_TARGET,
                   ffi.OPT_P3M_POOLED_DRAIN, ffi.OPT_P3M_PAIR_ONCE):
            lib.set_option(k_, 0)
    for name in out:
        assert normalised_error(out[name], out["cells"], floor) < 1e-11, name


# --------------------------------------------------------------------------------------
# P3M cell kernel: every unordered pair once (reactions gathered per brick in LDS)
# --------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,n_side,mesh", [("lattice", 48, 48), ("lattice_r2", 64, 64), ("uneven_grid", 37, 41),
                                              ("lattice_soft0", 32, 32), ("unequal_masses", 40, 40),
                                              ("radius_3", 64, 64)])
def test_p3m_pair_once_equals_ordered_pairs(lib, oracle, name, n_side, mesh):
    """k_p3m_sym against k_p3m_cells on the same inputs: twice the unordered pairs it evaluates
    must be the ordered-pair count (an exact integer, equal to the oracle's), and the sums must
    agree to rounding.  Grids that are not a multiple of the brick size, both search radii,
    zero softening and unequal masses (the reaction carries the target's mass); a search radius of
    three cells on a set that gets two by itself (the brick's extent in cells follows the radius)."""
    n = n_side ** 3
    pos = datasets.jittered_lattice(n_side, jitter=0.9)
    m = datasets.masses(n, equal=(name != "unequal_masses"))
    config = Config(MESH_SIZE=mesh, N_PER_SIDE=n_side, USE_PM=True)
    if name == "lattice_soft0":
        config.SOFTENING_SQUARED = 0.0
    p = config.pp_params()
    out = {}
    try:
        lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, 1)
        lib.set_option(ffi.OPT_P3M_POOLED_DRAIN, -1)
        lib.set_option(ffi.OPT_P3M_SEARCH_RADIUS, 3 if name == "radius_3" else 0)
        for once in (-1, 1):
            lib.set_option(ffi.OPT_P3M_PAIR_ONCE, once)
            out[once] = gpu_acc(lib, p, pos, m, acc0=[np.full(n, 0.5), np.zeros(n), np.full(n, -2.0)])
            st = lib.stats()
            assert st["kernel_variant"] == ffi.KERNEL_P3M_CELLS and st["p3m_pooled"] == (2 if once > 0 else 0)
            out[once, "accepted"] = st["pairs_accepted"]
    finally:
        lib.set_option(ffi.OPT_P3M_SEARCH_RADIUS, 0)
        lib.set_option(ffi.OPT_P3M_LANES_PER_TARGET, 0)
        lib.set_option(ffi.OPT_P3M_POOLED_DRAIN, 0)
        lib.set_option(ffi.OPT_P3M_PAIR_ONCE, 0)
    assert out[1, "accepted"] == out[-1, "accepted"] == oracle.count_within_cutoff(p, *pos)
    floor = pair_force_scale(p, m, n) * 1e-3
    start = [0.5, 0.0, -2.0]
    a = [o - s for o, s in zip(out[1], start)]
    b = [o - s for o, s in zip(out[-1], start)]
    assert normalised_error(a, b, floor) < 1e-11
    n_tgt = 2048
    want = oracle.accumulate(p, *pos, m, tgt_begin=0, n_tgt=n_tgt, traversal=O.EXACT_CELLS)
    assert normalised_error([c[:n_tgt] for c in a], want, floor) < TOL


@pytest.mark.parametrize("use_pm,n_side,pair_once", [(True, 96, 0), (True, 64, -1), (False, 40, 0)])
def test_forces_do_not_depend_on_the_order_of_the_particles_in_memory(lib, use_pm, n_side, pair_once):
    """f(P x) = P f(x): the same particles handed over in a random order get the same forces, particle by
    particle, to the rounding of the sums, and exactly the same number of accepted pairs.  The binning
    gathers records and scatters sums by original index, so this walks a path the lattice-ordered
    sets never take: neighbours in space that are far apart in memory.  P3M through the pair-once
    kernel (the library's choice for 96^3) and through the pooled kernel, direct summation through
    the pair-once all-pairs kernel."""
    n = n_side ** 3
    pos = datasets.jittered_lattice(n_side, jitter=0.9)
    m = datasets.masses(n, equal=False)
    p = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=use_pm).pp_params()
    order = np.random.default_rng(11).permutation(n)
    lib.set_option(ffi.OPT_P3M_PAIR_ONCE, pair_once)
    try:
        straight = gpu_acc(lib, p, pos, m)
        accepted = lib.stats()["pairs_accepted"]
        shuffled = gpu_acc(lib, p, [np.ascontiguousarray(c[order]) for c in pos], np.ascontiguousarray(m[order]))
        assert lib.stats()["pairs_accepted"] == accepted
    finally:
        lib.set_option(ffi.OPT_P3M_PAIR_ONCE, 0)
    floor = pair_force_scale(p, m, n) * 1e-3
    assert normalised_error(shuffled, [c[order] for c in straight], floor) < 1e-10


@pytest.mark.parametrize("pair_once", [0, -1])
def test_p3m_forces_are_invariant_under_a_shift_of_the_box(lib, pair_once):
    """The box is periodic: moving every particle by the same vector (and folding back into [0, L))
    changes which particles sit near the faces, hence every ghost image, every brick and every row
    window, but no force.  128^3 particles, a shift of a third of the box and an odd fraction of a
    cell; the folded positions are rounded anew, which moves a pair's separation by ~1e-16 L, so the
    sums agree to 1e-10 of the pair-force scale and a handful of pairs may cross the cutoff."""
    n_side = 128
    n = n_side ** 3
    pos = datasets.jittered_lattice(n_side, jitter=0.9)
    m = datasets.masses(n, equal=True)
    config = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True)
    p = config.pp_params()
    L = config.DOMAIN_SIZE
    shift = np.array([1.0 / 3.0, 0.5 + 0.37 / n_side, 0.9991]) * L
    moved = [np.ascontiguousarray(np.mod(c + s, L)) for c, s in zip(pos, shift)]
    lib.set_option(ffi.OPT_P3M_PAIR_ONCE, pair_once)
    try:
        here = gpu_acc(lib, p, pos, m)
        accepted = lib.stats()["pairs_accepted"]
        there = gpu_acc(lib, p, moved, m)
        assert abs(lib.stats()["pairs_accepted"] - accepted) <= 8
    finally:
        lib.set_option(ffi.OPT_P3M_PAIR_ONCE, 0)
    floor = pair_force_scale(p, m, n) * 1e-3
    assert normalised_error(there, here, floor) < 1e-9


def test_p3m_pair_once_falls_back_on_a_clustered_set(lib):
    """A brick whose neighbourhood does not fit in LDS: the ordinary kernels take the call."""
    n = 64 ** 3
    pos = datasets.clustered(n, n_blobs=3, sigma=0.01)
    m = datasets.masses(n)
    p = Config(MESH_SIZE=64, N_PER_SIDE=64, USE_PM=True).pp_params()
    lib.set_option(ffi.OPT_P3M_PAIR_ONCE, 1)
    try:
        gpu_acc(lib, p, pos, m)
        assert lib.stats()["p3m_pooled"] != 2
    finally:
        lib.set_option(ffi.OPT_P3M_PAIR_ONCE, 0)


def test_p3m_needs_a_positive_mesh_softening(lib):
    """(r^2 + pm_soft_sq)^(-3/2) is evaluated at r = 0 as well (a target's own record passes the pooled
    kernels' filter): pm_soft_sq = 0 would put NaN into every sum, so it is refused as an argument
    error before anything is added.  The reference's (CELL_SIZE / 2)^2 is always positive."""
    n_side = 16
    n = n_side ** 3
    pos = datasets.jittered_lattice(n_side)
    m = datasets.masses(n, equal=True)
    p = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=True).pp_params()
    p.pm_soft_sq = 0.0
    with pytest.raises(ffi.NbppError) as err:
        gpu_acc(lib, p, pos, m)
    assert "pm_soft_sq" in str(err.value)


@pytest.mark.parametrize("use_pm", [True, False])
def test_compute_calls_on_two_streams_do_not_disturb_each_other(lib, use_pm):
    """The library's working buffers are one set per process.  Two compute calls queued back to
    back on two different non-blocking streams, with nothing in the caller's code ordering them, must
    both give the result of a call made alone: the library makes the second stream wait for the
    first call's kernels."""
    import torch
    n_side = 64 if use_pm else 40
    n = n_side ** 3
    p = Config(MESH_SIZE=n_side, N_PER_SIDE=n_side, USE_PM=use_pm).pp_params()
    sets = []
    for seed in (5, 6):
        pos = datasets.jittered_lattice(n_side, seed=seed)
        sets.append([torch.as_tensor(a, device="cuda:0") for a in (*pos, datasets.masses(n, equal=True))])
    lib.set_option(ffi.OPT_DETERMINISTIC, 1)     # reproducible kernels: equality can be exact
    try:
        alone = []
        for d in sets:
            acc = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
            lib.accumulate_device(p, *d, *acc)
            torch.cuda.synchronize()
            alone.append(acc)
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        for _ in range(3):
            got = [[torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)] for _ in sets]
            torch.cuda.synchronize()
            for d, acc, st in zip(sets, got, streams):
                lib.accumulate_device(p, *d, *acc, stream=st.cuda_stream)
            torch.cuda.synchronize()
            for acc, want in zip(got, alone):
                for a, w in zip(acc, want):
                    assert torch.equal(a, w)
    finally:
        lib.set_option(ffi.OPT_DETERMINISTIC, 0)

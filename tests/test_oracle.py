"""The CPU oracle against the reference's own known answers and compiled config/math code."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, normalised_error, pair_force_scale
from oracle import oracle as O
import datasets


def _golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def _two_body_params(o, kat, section):
    c = kat["config"]
    p = o.params_from_config(0, mesh_size=c["MESH_SIZE"], n_per_side=2)
    # the reference test overwrites these Config fields by hand (tests/test_gravity.cpp:100-108)
    p.domain_size = c["DOMAIN_SIZE"]
    p.cell_size = c["CELL_SIZE"]
    p.G = c["G"]
    p.soft_sq = c["SOFTENING_SQUARED"]
    p.pm_soft_sq = (0.5 * c["CELL_SIZE"]) ** 2
    s = kat[section]
    p.use_pm = 1 if s["USE_PM"] else 0
    p.cutoff_sq = s["CUTOFF_RADIUS_SQUARED"]
    if s["USE_PM"]:
        p.r_switch = s["R_SWITCH_START"]
        p.r_switch_sq = s["R_SWITCH_START_SQ"]
        p.trans_width = s["CUTOFF_TRANSITION_WIDTH"]
    return p


@pytest.mark.parametrize("traversal", [O.EXACT, O.REF_OFFLOAD, O.REF_CPU])
def test_two_body_newtonian_kat(oracle, traversal):
    """tests/test_gravity.cpp:115-134: acc_x = +-1/9, y = z = 0."""
    kat = _golden("reference_kat.json")["two_body"]
    p = _two_body_params(oracle, kat, "newtonian")
    pts = np.array(kat["particles"])
    ax, ay, az = oracle.accumulate(p, pts[:, 0], pts[:, 1], pts[:, 2], pts[:, 3],
                                   traversal=traversal)
    want = kat["newtonian"]["acc_x"]
    assert ax[0] == pytest.approx(want[0], rel=kat["approx_rel"])
    assert ax[1] == pytest.approx(want[1], rel=kat["approx_rel"])
    assert ax[0] == 1.0 / 9.0 and ax[1] == -1.0 / 9.0
    assert ay[0] == 0.0 and az[0] == 0.0


@pytest.mark.parametrize("traversal", [O.EXACT, O.REF_OFFLOAD, O.REF_CPU, O.EXACT_CELLS])
def test_two_body_p3m_cutoff_kat(oracle, traversal):
    """tests/test_gravity.cpp:136-154: separation 3 > cutoff 2 gives exactly 0."""
    kat = _golden("reference_kat.json")["two_body"]
    p = _two_body_params(oracle, kat, "p3m_cutoff")
    pts = np.array(kat["particles"])
    ax, ay, az = oracle.accumulate(p, pts[:, 0], pts[:, 1], pts[:, 2], pts[:, 3],
                                   traversal=traversal)
    assert ax[0] == 0.0 and ax[1] == 0.0


def test_periodic_displacement_kat(oracle):
    """tests/test_math.cpp:14-47."""
    kat = _golden("reference_kat.json")["periodic_displacement"]
    for case in kat["cases"]:
        d = oracle.periodic_displacement(case["dx"], kat["box_size"])
        if "expect" in case:
            assert d == case["expect"]
        else:
            assert abs(d) == case["expect_abs"]


def test_periodic_displacement_matches_reference_build(oracle):
    """Bit-for-bit against the reference's compiled math_utils.h (committed outputs)."""
    for row in _golden("reference_wrap.json"):
        dx, L, want = (float.fromhex(row[k]) for k in ("dx", "L", "out"))
        got = oracle.periodic_displacement(dx, L)
        assert got == want and np.signbit(got) == np.signbit(want)


def test_config_derive_matches_reference_build(oracle):
    """Bit-for-bit against the reference's compiled config.cpp (committed outputs)."""
    gold = _golden("reference_config.json")
    assert len(gold) >= 6
    for name, ref in gold.items():
        got = oracle.config_derive(
            domain_size=ref["domain_size"], mesh_size=ref["mesh_size"],
            n_per_side=ref["n_per_side"], omega_baryon=ref["omega_baryon"],
            omega_m=ref["omega_m"], cutoff_radius_cells=ref["cutoff_radius_cells"],
            cutoff_transition_width_factor=ref["cutoff_transition_width_factor"],
            dt_factor=ref["dt_factor"])
        for k, v in got.items():
            assert v == ref[k], (name, k, v, ref[k])


def test_default_ini_constants(oracle):
    """The derived constants of the shipped simulation.ini (32^3, mesh 32)."""
    ref = _golden("reference_config.json")["simulation_ini"]
    assert ref["G"] == 0.053051647697298449
    assert ref["softening_squared"] == 3.9062500000000002e-07
    assert ref["cutoff_radius"] == 0.078125 and ref["r_switch_start"] == 0.0625
    assert ref["cutoff_transition_width"] == 0.015625
    assert ref["dm_particle_mass"] == 2.5634765624999999e-05
    assert ref["fixed_dt"] == 0.0043416075273496066


@pytest.mark.skipif(not O.Reference.available(), reason="oracle/_ref not built here")
def test_live_reference_build_agrees(oracle):
    ref = O.Reference()
    rng = np.random.default_rng(5)
    for L in (1.0, 64.0, 7.3):
        for d in (rng.random(200) * 2 - 1) * 1.45 * L:
            assert oracle.periodic_displacement(d, L) == ref.periodic_displacement(d, L)
    for mesh, nps in ((32, 32), (31, 32), (128, 128), (7, 5)):
        a = oracle.config_derive(mesh_size=mesh, n_per_side=nps)
        b = ref.config_derive(mesh_size=mesh, n_per_side=nps)
        for k, v in a.items():
            assert v == b[k]


@pytest.mark.parametrize("name", ["lattice8", "lattice12_even_exact", "uniform1000",
                                  "clustered1500"])
@pytest.mark.parametrize("mode", [0, 1])
def test_restatement_vs_long_double_golden(oracle, name, mode):
    """Reference-order double arithmetic vs the independent long-double sum: <= 1e-12."""
    g = np.load(os.path.join(GOLDEN, "pp_small.npz"))
    pos, m, want = g[f"{name}_pos"], g[f"{name}_mass"], g[f"{name}_acc_mode{mode}"]
    pv = g[f"{name}_params_mode{mode}"]
    p = O.Params(*pv[:9], int(pv[9]), int(pv[10]))
    got = oracle.accumulate(p, pos[0], pos[1], pos[2], m)
    floor = pair_force_scale(p, m, m.size) if "exact" in name else 0.0
    assert normalised_error(got, want, floor) < 1e-12


def test_traversals_agree_where_reference_is_exact(oracle):
    """Odd mesh, PP-only, n <= 1500: the reference's sweep visits every cell once, so both of
    its branches equal the exact sum; P3M with mesh >= 7 likewise."""
    pos = datasets.jittered_lattice(11)
    m = datasets.masses(pos[0].size)
    p = oracle.params_from_config(0, mesh_size=11, n_per_side=11)
    # different summation orders of the same terms: rounding only (observed ~2e-12)
    base = oracle.accumulate(p, *pos, m, traversal=O.EXACT)
    for t in (O.REF_OFFLOAD, O.REF_CPU):
        assert normalised_error(oracle.accumulate(p, *pos, m, traversal=t), base) < 1e-10
    p = oracle.params_from_config(1, mesh_size=11, n_per_side=11)
    base = oracle.accumulate(p, *pos, m, traversal=O.EXACT)
    for t in (O.REF_OFFLOAD, O.REF_CPU, O.EXACT_CELLS):
        assert normalised_error(oracle.accumulate(p, *pos, m, traversal=t), base) < 1e-10


def test_reference_quirks_are_reproduced(oracle):
    """Even mesh, PP-only: the +-N/2 cell planes alias, so a pair exactly N/2 cells apart is
    counted twice by the reference's sweep (force ratio 2); and the CPU branch stops gathering
    at 1500 neighbours."""
    p = oracle.params_from_config(0, mesh_size=4, n_per_side=2)
    x = np.array([0.1, 0.6]); y = np.array([0.1, 0.1]); z = np.array([0.1, 0.1])
    m = np.array([0.5, 0.5])
    exact = oracle.accumulate(p, x, y, z, m, traversal=O.EXACT)
    quirk = oracle.accumulate(p, x, y, z, m, traversal=O.REF_OFFLOAD)
    assert quirk[0][0] == pytest.approx(2.0 * exact[0][0], rel=1e-15)
    pos = datasets.uniform_random(1700)
    mm = datasets.masses(1700)
    p = oracle.params_from_config(0, mesh_size=5, n_per_side=12)
    full = oracle.accumulate(p, *pos, mm, traversal=O.REF_OFFLOAD)
    capped = oracle.accumulate(p, *pos, mm, traversal=O.REF_CPU)
    assert normalised_error(full, oracle.accumulate(p, *pos, mm, traversal=O.EXACT)) < 1e-10
    assert normalised_error(capped, full) > 1e-3


def test_empty_and_single(oracle):
    p = oracle.params_from_config(0)
    e = np.zeros(0)
    ax, ay, az = oracle.accumulate(p, e, e, e, e)
    assert ax.size == 0
    one = np.array([0.3])
    acc = [np.array([1.5]), np.array([-2.0]), np.array([0.25])]
    oracle.accumulate(p, one, one, one, one, acc=acc)
    assert acc[0][0] == 1.5 and acc[1][0] == -2.0 and acc[2][0] == 0.25


def test_accumulate_semantics_and_target_slice(oracle):
    pos = datasets.uniform_random(300)
    m = datasets.masses(300, equal=False)
    p = oracle.params_from_config(1, mesh_size=8, n_per_side=7)
    full = oracle.accumulate(p, *pos, m)
    acc = [np.full(100, 2.0), np.full(100, -1.0), np.full(100, 0.5)]
    oracle.accumulate(p, *pos, m, acc=acc, tgt_begin=150, n_tgt=100)
    for c, off in zip(range(3), (2.0, -1.0, 0.5)):
        assert np.array_equal(acc[c], off + full[c][150:250])


def test_cutoff_count(oracle):
    pos = datasets.jittered_lattice(16)
    p = oracle.params_from_config(1, mesh_size=16, n_per_side=16)
    cnt = oracle.count_within_cutoff(p, *pos)
    per = cnt / pos[0].size
    assert 55 < per < 80  # sphere of 2.5 cells at one particle per cell: 4/3 pi 2.5^3 = 65.4


def test_caller_contract_pieces(oracle):
    rng = np.random.default_rng(0)
    n = 1000
    v = [rng.standard_normal(n) for _ in range(3)]
    a = [rng.standard_normal(n) for _ in range(3)]
    v0 = [c.copy() for c in v]
    dt, sc, H = 1e-3, 0.5, 3.0
    oracle.kick(v, a, dt, sc, H)
    want = v0[1] + (a[1] * (dt / (sc * sc * sc)) - v0[1] * (dt * 2.0 * H))
    assert np.array_equal(v[1], want)
    x = [rng.random(n) for _ in range(3)]
    x0 = [c.copy() for c in x]
    oracle.drift(x, v, 0.7, 1.0)
    nx = x0[2] + v[2] * 0.7
    assert np.array_equal(x[2], nx - 1.0 * np.floor(nx * 1.0))
    assert all((c >= 0).all() and (c <= 1.0).all() for c in x)
    assert oracle.max_accel_sq(a) == max(1e-9, float(np.max(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])))
    tiny = [np.full(4, 1e-7)] * 3
    assert oracle.max_accel_sq(tiny) == 1e-9
    assert oracle.gravity_timestep(10, 4e-6, 1e-4, 0.2) == np.sqrt(4e-6) / np.sqrt(1e-4) * 0.2
    assert oracle.gravity_timestep(0, 4e-6, 1e-4, 0.2) == np.inf


def test_zeldovich_ics_and_total_acceleration_match_reference_output(oracle):
    """The shipped simulation.ini through the restated pipeline (white noise -> Zel'dovich
    displacements -> particles -> mesh force + short-range force) against the numbers a build
    of the reference printed: particle 0's position bit for bit, its total initial acceleration
    and the global max |a|^2 to 1e-13."""
    from oracle.zeldovich import zeldovich_particles
    from oracle.pm_mesh import initial_mesh_accelerations
    anchor = _golden("reference_kat.json")["default_ini_anchor"]
    cfg = _golden("reference_config.json")["simulation_ini"]
    pos, vel, info = zeldovich_particles(oracle=oracle)
    assert info["scale_factor"] == anchor["scale_factor"]
    assert info["hubble_param"] == anchor["hubble_param"]
    for c in range(3):
        assert pos[c][0] == anchor["p0_pos"][c]
        assert vel[c][0] == pytest.approx(anchor["p0_vel"][c], rel=1e-13)
    m = np.full(pos[0].size, cfg["dm_particle_mass"])
    acc = initial_mesh_accelerations(pos, m, 32, 1.0, cfg["G"], cfg["gas_total_mass"])
    p = oracle.params_from_config(1, omega_baryon=cfg["omega_baryon"])
    oracle.accumulate(p, *pos, m, acc=acc, traversal=O.REF_CPU)
    for c in range(3):
        assert acc[c][0] == pytest.approx(anchor["p0_acc_initial_pm_plus_pp"][c], rel=1e-13)
    assert oracle.max_accel_sq(acc) == pytest.approx(anchor["max_accel_sq"], rel=1e-13)

"""GPU parity of libnbpm (the mesh side of the gravity step, through its C ABI) against the numpy
restatement in oracle/pm_mesh.py and against values a build of the reference printed.

Tolerances: the interpolation and the gradient follow the reference's operation order and are
compared exactly; the density is summed with atomics (order varies) and the two FFTs are
hipFFT's instead of pocketfft's, so density, potential and everything downstream are compared
to 1e-12 of the field's largest magnitude (observed ~1e-15).
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from nbody_amd import Config, ParticleSystem, ffi
from oracle import pm_mesh
import datasets

pytestmark = pytest.mark.gpu
TOL = 1e-12


@pytest.fixture(scope="module")
def mesh():
    L = ffi.load_mesh(device=0)
    L.set_timing(True)
    yield L


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")


def params(M, L=1.0, G=1.0 / (6.0 * np.pi)):
    p = ffi.MeshParams()
    p.domain_size, p.G, p.mesh_size = L, G, M
    return p


def rel_to_max(got, ref):
    return float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))


@pytest.mark.parametrize("M,n_side,L", [(16, 16, 1.0), (32, 20, 2.5), (7, 9, 1.0), (64, 48, 1.0)])
def test_deposit_matches_oracle(mesh, M, n_side, L):
    pos = datasets.jittered_lattice(n_side, L=L, jitter=0.9, seed=5)
    m = datasets.masses(n_side ** 3, equal=False)
    rho_ref, _, _ = pm_mesh.cic_deposit(pos, m, M, L)
    rho = torch.full((M ** 3,), 7.0, dtype=torch.float64, device="cuda:0")  # must be overwritten
    mesh.deposit_device(params(M, L), *[dev(a) for a in pos], dev(m), rho)
    torch.cuda.synchronize()
    got = rho.cpu().numpy().reshape(M, M, M)
    assert rel_to_max(got, rho_ref) < TOL
    # total mass is conserved by the assignment
    assert got.sum() * (L / M) ** 3 == pytest.approx(m.sum(), rel=1e-13)


def test_deposit_particles_on_the_faces(mesh):
    """x == 0 and x == L (what the drift can produce, src/integrator.cpp:194) land in cell 0
    through the reference's (ix + N) % N."""
    M, L = 8, 2.0
    x = np.array([0.0, L, L, 0.3, L - 1e-16])
    y = np.array([L, 0.0, L, L, 0.5])
    z = np.array([0.0, 0.0, L, 1.999999, L])
    m = np.array([1.0, 2.0, 3.0, 4.0, 5.0])
    rho_ref, _, _ = pm_mesh.cic_deposit([x, y, z], m, M, L)
    rho = torch.zeros(M ** 3, dtype=torch.float64, device="cuda:0")
    mesh.deposit_device(params(M, L), dev(x), dev(y), dev(z), dev(m), rho)
    torch.cuda.synchronize()
    assert rel_to_max(rho.cpu().numpy().reshape(M, M, M), rho_ref) < TOL


@pytest.mark.parametrize("M,L,G", [(16, 1.0, 1.0 / (6 * np.pi)), (32, 3.0, 2.0), (9, 1.0, 1.0), (64, 1.0, 0.05)])
def test_solve_matches_oracle(mesh, M, L, G):
    rng = np.random.default_rng(M)
    rho = rng.random((M, M, M)) + 0.1
    ref = pm_mesh.poisson_accelerations(rho, L, G)
    d_rho = dev(rho.ravel())
    grids = [torch.zeros(M ** 3, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    mesh.solve_device(params(M, L, G), d_rho, *grids)
    torch.cuda.synchronize()
    assert np.array_equal(d_rho.cpu().numpy(), rho.ravel())          # the density is preserved
    for g, r in zip(grids, ref):
        assert rel_to_max(g.cpu().numpy().reshape(M, M, M), r) < TOL


def test_solve_of_a_single_mode_is_analytic(mesh):
    """rho = cos(2 pi x / L): phi = -4 pi G rho / k^2 and the central difference of phi along x
    is known in closed form; y and z components vanish."""
    M, L, G = 32, 2.0, 0.7
    x = (np.arange(M) * (L / M))[:, None, None] * np.ones((1, M, M))
    k = 2 * np.pi / L
    rho = np.cos(k * x)
    grids = [torch.zeros(M ** 3, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    mesh.solve_device(params(M, L, G), dev(rho.ravel()), *grids)
    torch.cuda.synchronize()
    h = L / M
    expected = -(-4 * np.pi * G / k ** 2) * (np.cos(k * (x + h)) - np.cos(k * (x - h))) / (2 * h)
    gx = grids[0].cpu().numpy().reshape(M, M, M)
    assert np.max(np.abs(gx - expected)) < 1e-12 * np.max(np.abs(expected))
    assert float(grids[1].abs().max()) < 1e-12 * np.max(np.abs(expected))
    assert float(grids[2].abs().max()) < 1e-12 * np.max(np.abs(expected))


@pytest.mark.parametrize("M,n_side,L", [(16, 16, 1.0), (12, 21, 4.0)])
def test_interpolation_is_bit_exact(mesh, M, n_side, L):
    rng = np.random.default_rng(3)
    pos = datasets.jittered_lattice(n_side, L=L, jitter=0.9, seed=8)
    pos[0][0], pos[1][0], pos[2][0] = L, 0.0, L          # faces
    grids = [rng.standard_normal((M, M, M)) for _ in range(3)]
    _, idx, weights = pm_mesh.cic_deposit(pos, np.ones(n_side ** 3), M, L)
    ref = pm_mesh.cic_interpolate(grids, idx, weights)
    acc = [torch.full((n_side ** 3,), 9.0, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    mesh.interpolate_device(params(M, L), *[dev(a) for a in pos], *[dev(g.ravel()) for g in grids], *acc)
    torch.cuda.synchronize()
    for a, r in zip(acc, ref):
        assert np.array_equal(a.cpu().numpy(), r)


@pytest.mark.parametrize("n_side,M", [(16, 16), (32, 32), (24, 16)])
def test_mesh_accelerations_match_oracle(mesh, n_side, M):
    pos = datasets.jittered_lattice(n_side, jitter=0.8, seed=21)
    m = datasets.masses(n_side ** 3)
    G = 1.0 / (6.0 * np.pi)
    ref = pm_mesh.initial_mesh_accelerations(pos, m, M, 1.0, G)
    acc = [np.full(m.size, 3.0) for _ in range(3)]                   # must be overwritten
    mesh.accelerations(params(M), *pos, m, *acc)
    scale = max(float(np.max(np.abs(r))) for r in ref)
    for a, r in zip(acc, ref):
        assert float(np.max(np.abs(a - r))) < TOL * scale
    st = mesh.stats()
    assert st["total_ms"] > 0.0
    # the assignment and the interpolation use the same cloud, so the mesh force exerts no net
    # force on the system
    for a in acc:
        assert abs(float(np.sum(m * a))) < 1e-12 * float(np.sum(m * np.abs(a)))


def test_device_and_host_entry_points_agree(mesh):
    n_side, M = 20, 16
    pos = datasets.jittered_lattice(n_side, jitter=0.8, seed=2)
    m = datasets.masses(n_side ** 3)
    host = [np.zeros(m.size) for _ in range(3)]
    mesh.accelerations(params(M), *pos, m, *host)
    d = [torch.zeros(m.size, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    mesh.accelerations_device(params(M), *[dev(a) for a in pos], dev(m), *d)
    torch.cuda.synchronize()
    scale = max(float(np.max(np.abs(h))) for h in host)
    for a, h in zip(d, host):   # the deposit's atomics make the two runs differ in the last bits
        assert float(np.max(np.abs(a.cpu().numpy() - h))) < TOL * scale


def test_empty_and_bad_arguments(mesh):
    e = np.zeros(0)
    mesh.accelerations(params(8), e, e, e, e, e.copy(), e.copy(), e.copy())
    rho = torch.ones(8 ** 3, dtype=torch.float64, device="cuda:0")
    t = torch.zeros(0, dtype=torch.float64, device="cuda:0")
    mesh.deposit_device(params(8), t, t, t, t, rho)                  # no particles: all zero
    torch.cuda.synchronize()
    assert float(rho.abs().max()) == 0.0
    with pytest.raises(ffi.NbppError):
        mesh.accelerations(params(1), e, e, e, e, e, e, e)
    bad = params(8)
    bad.domain_size = 0.0
    with pytest.raises(ffi.NbppError):
        mesh.accelerations(bad, e, e, e, e, e, e, e)


def test_mirrored_step_order_pm_then_pp(lib, mesh, oracle):
    """The force phase of the reference's step (src/integrator.cpp:239-258): mesh force
    overwrites acc, the short-range force is accumulated on top."""
    n_side = 16
    config = Config(N_PER_SIDE=n_side, MESH_SIZE=n_side, OMEGA_BARYON=0.0)
    pos = datasets.jittered_lattice(n_side, jitter=0.8, seed=4)
    m = np.full(n_side ** 3, config.DM_PARTICLE_MASS)
    ps = ParticleSystem(config, lib)
    ps.set_particles(*pos, m)
    ps.bin_and_assign_mass(config)
    ps.compute_mesh_forces(config, mesh)
    ps.compute_pp_forces(config)
    ref = pm_mesh.initial_mesh_accelerations(pos, m, n_side, 1.0, config.G)
    pp = oracle.accumulate(config.pp_params(), *pos, m)
    for got, a, b in zip((ps.acc_x, ps.acc_y, ps.acc_z), ref, pp):
        assert float(np.max(np.abs(got - (a + b)))) < 1e-11 * float(np.max(np.abs(a + b)))


def test_default_run_initial_accelerations_match_reference_output_on_the_gpu(lib, mesh, oracle):
    """Mesh + short range entirely on the GPU for the shipped simulation.ini (32^3 Zel'dovich
    particles, hydro on so the gas density joins the source, src/ics.cpp:249-256, :300-322)
    against the values a build of the reference printed (tests/golden/reference_kat.json)."""
    from oracle.zeldovich import zeldovich_particles
    kat = json.load(open(os.path.join(GOLDEN, "reference_kat.json")))["default_ini_anchor"]
    cfg = json.load(open(os.path.join(GOLDEN, "reference_config.json")))["simulation_ini"]
    pos, _, _ = zeldovich_particles(oracle=oracle)
    n, M = 32768, 32
    m = np.full(n, cfg["dm_particle_mass"])
    p = params(M, 1.0, cfg["G"])
    d_pos, d_m = [dev(a) for a in pos], dev(m)
    rho = torch.zeros(M ** 3, dtype=torch.float64, device="cuda:0")
    mesh.deposit_device(p, *d_pos, d_m, rho)
    gas = rho * (cfg["gas_total_mass"] / (n * cfg["dm_particle_mass"]))      # src/ics.cpp:249-256
    gas = torch.where(gas < 1e-12, torch.full_like(gas, 1e-12), gas)
    total = rho + gas                                                          # src/integrator.cpp:57-59
    grids = [torch.zeros(M ** 3, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    mesh.solve_device(p, total, *grids)
    acc = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    mesh.interpolate_device(p, *d_pos, *grids, *acc)
    config = Config(OMEGA_BARYON=cfg["omega_baryon"])
    lib.accumulate_device(config.pp_params(), *d_pos, d_m, *acc)            # src/ics.cpp:319-322
    torch.cuda.synchronize()
    a = [t.cpu().numpy() for t in acc]
    for c in range(3):
        assert a[c][0] == pytest.approx(kat["p0_acc_initial_pm_plus_pp"][c], rel=1e-12)
    assert float(np.max(a[0] ** 2 + a[1] ** 2 + a[2] ** 2)) == pytest.approx(kat["max_accel_sq"], rel=1e-12)


def test_potential_of_the_last_solve(mesh):
    """nbpm_potential_device hands out the unnormalised inverse transform the reference keeps as
    state.phi (src/integrator.cpp:105-106)."""
    M, L, G = 16, 1.0, 0.3
    rng = np.random.default_rng(1)
    rho = rng.random((M, M, M))
    grids = [torch.zeros(M ** 3, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    p = params(M, L, G)
    mesh.solve_device(p, dev(rho.ravel()), *grids)
    d_phi = torch.zeros(M ** 3, dtype=torch.float64, device="cuda:0")
    mesh.potential_device(p, d_phi)
    torch.cuda.synchronize()
    phi = d_phi.cpu().numpy().reshape(M, M, M)
    rho_k = np.fft.rfftn(rho)
    f = np.where(np.arange(M) < M // 2, np.arange(M), np.arange(M) - M) * 2.0 * np.pi / L
    k2 = f[:, None, None] ** 2 + f[None, :, None] ** 2 + (np.arange(M // 2 + 1) * 2.0 * np.pi / L)[None, None, :] ** 2
    k2[0, 0, 0] = 1.0
    phi_k = rho_k * (-4.0 * np.pi * G / k2)
    phi_k[0, 0, 0] = 0.0
    ref = np.fft.irfftn(phi_k, s=(M, M, M)) * M ** 3
    assert rel_to_max(phi, ref) < TOL
    with pytest.raises(ffi.NbppError):
        mesh.potential_device(params(M + 1, L, G), torch.zeros((M + 1) ** 3, dtype=torch.float64, device="cuda:0"))


@pytest.mark.parametrize("n_side,mesh,seed,standing", [(20, 20, 3, False), (12, 18, 5, False), (16, 16, 42, True)])
def test_initial_conditions_on_the_device_match_the_oracle(n_side, mesh, seed, standing):
    """nbpm_zeldovich_field_device + nbpm_zeldovich_particles_device through the C ABI, fed with the
    white noise of libstdc++'s stream (the oracle's C restatement of it), against the numpy pipeline
    of oracle/zeldovich.py: displacement meshes, positions and velocities.  Mesh and lattice of
    different sizes exercise the cell look-up; `standing` the particles at rest (src/ics.cpp:200-208)."""
    import ctypes as C
    import nbody_amd
    from nbody_amd import ffi
    from oracle.oracle import Oracle, _dp
    from oracle.zeldovich import zeldovich_particles
    lib = nbody_amd.load_mesh(device=0)
    pos, vel, info = zeldovich_particles(n_per_side=n_side, mesh_size=mesh, seed=seed)
    orc = Oracle()
    orc.lib.orc_libstdcxx_normal_field.argtypes = [C.c_uint32, C.c_size_t, C.POINTER(C.c_double)]
    white = np.empty(mesh ** 3)
    orc.lib.orc_libstdcxx_normal_field(seed, white.size, _dp(white))
    ip = ffi.IcParams(mesh_size=mesh, n_per_side=n_side, domain_size=1.0, box_size_mpc=16.0, hubble_param=0.7,
                      omega_m=0.3, spectral_index=0.96, scale_factor=info["scale_factor"],
                      amplitude=lib.sigma8_amplitude(0.81, 0.3, 0.7, 0.96))
    assert ip.amplitude == pytest.approx(info["A"], rel=1e-12)
    noise = torch.as_tensor(white, device="cuda:0")
    disp = [torch.empty(mesh ** 3, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    lib.zeldovich_field_device(ip, noise, *disp)
    torch.cuda.synchronize()
    assert np.array_equal(noise.cpu().numpy(), white)                       # the input is preserved
    scale = max(float(np.max(np.abs(d))) for d in info["displacement"])
    for got, want in zip(disp, info["displacement"]):
        assert np.max(np.abs(got.cpu().numpy().reshape(mesh, mesh, mesh) - want)) < 1e-12 * scale
    n = n_side ** 3
    out = [torch.full((n,), -7.0, dtype=torch.float64, device="cuda:0") for _ in range(7)]
    factor = 0.0 if standing else info["hubble_param"] * info["growth_f"]
    lib.zeldovich_particles_device(ip, disp, factor, 0.125, out[:3], out[3:6], m=out[6])
    torch.cuda.synchronize()
    for c in range(3):
        d = np.abs(out[c].cpu().numpy() - pos[c])
        assert np.max(np.minimum(d, 1.0 - d)) < 1e-13                       # either side of the wrap
        if standing:
            assert not out[3 + c].any()
        else:
            assert np.max(np.abs(out[3 + c].cpu().numpy() - vel[c])) < 1e-11 * np.max(np.abs(vel[c]))
    assert torch.all(out[6] == 0.125)
    assert all(0.0 <= float(o.min()) and float(o.max()) < 1.0 for o in out[:3])

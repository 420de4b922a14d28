"""The gas side of the reference's step, restated in numpy: primitive variables, first-order
HLL fluxes with directional splitting, CFL time step, gravity kick and initial conditions
(src/gas.cpp, src/integrator.cpp:134-166, src/ics.cpp:218-288).

TEST INFRASTRUCTURE ONLY (see pp_oracle.h).  Every array operation keeps the reference's
expression structure (Eigen evaluates these coefficient-wise, left to right).  Pinned by the
solver's known-answer cases of the reference's tests/test_hydro.cpp:29-67 and, through
oracle/full_step.py, by the accelerations a build of the reference printed after one whole step of
the shipped simulation.ini (mesh + short range + gas).
"""
from __future__ import annotations

import math

import numpy as np

FLOOR = 1e-12


class Gas:
    """GasGrid's nine meshes (src/gas.h:46-48), each (M, M, M) with z fastest."""

    FIELDS = ("density", "momentum_x", "momentum_y", "momentum_z", "energy",
              "pressure", "velocity_x", "velocity_y", "velocity_z")

    def __init__(self, mesh_size):
        for f in self.FIELDS:
            setattr(self, f, np.zeros((mesh_size,) * 3))

    def copy(self):
        g = Gas(self.density.shape[0])
        for f in self.FIELDS:
            setattr(g, f, getattr(self, f).copy())
        return g


def update_primitive_variables(g: Gas, gamma):
    """GasGrid::update_primitive_variables (src/gas.cpp:186-248): velocities and pressure from the
    conserved fields, pressure floor, energy re-synchronised with the floor, vacuum cells."""
    gm1 = gamma - 1.0
    inv_gm1 = 1.0 / gm1
    rho = g.density
    ok = rho > FLOOR
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv_rho = 1.0 / rho
        vx = g.momentum_x * inv_rho
        vy = g.momentum_y * inv_rho
        vz = g.momentum_z * inv_rho
        ke = 0.5 * rho * (vx * vx + vy * vy + vz * vz)
        p = gm1 * (g.energy - ke)
        p = np.where(p < FLOOR, FLOOR, p)
        en = p * inv_gm1 + ke
    g.velocity_x = np.where(ok, vx, 0.0)
    g.velocity_y = np.where(ok, vy, 0.0)
    g.velocity_z = np.where(ok, vz, 0.0)
    g.pressure = np.where(ok, p, FLOOR)
    g.energy = np.where(ok, en, FLOOR * inv_gm1)


def solve_hll(FL, FR, UL, UR, S_L, S_R, S_R_minus_S_L):
    """RiemannSolver::solve_hll (src/gas.cpp:53-64)."""
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        mixed = (S_R * FL - S_L * FR + S_L * S_R * (UR - UL)) / S_R_minus_S_L
    return np.where(S_L >= 0, FL, np.where(S_R <= 0, FR, mixed))


_AXES = {0: ("x", "y", "z"), 1: ("y", "z", "x"), 2: ("z", "x", "y")}   # normal, t1, t2 (:74-98)


def compute_fluxes(g: Gas, axis, gamma):
    """RiemannSolver::compute_fluxes (src/gas.cpp:66-169): the five face fluxes at the +axis face
    of every cell, in the sweep's (normal, t1, t2) frame."""
    n, t1, t2 = _AXES[axis]
    R = lambda a: np.roll(a, -1, axis)                     # noqa: E731  neighbour in +axis (:103-123)
    rho_L, p_L, E_L = g.density, g.pressure, g.energy
    vn_L, vt1_L, vt2_L = (getattr(g, "velocity_" + c) for c in (n, t1, t2))
    mn_L, mt1_L, mt2_L = (getattr(g, "momentum_" + c) for c in (n, t1, t2))
    rho_R, p_R, E_R = R(rho_L), R(p_L), R(E_L)
    vn_R, vt1_R, vt2_R = R(vn_L), R(vt1_L), R(vt2_L)
    mn_R, mt1_R, mt2_R = R(mn_L), R(mt1_L), R(mt2_L)
    with np.errstate(divide="ignore", invalid="ignore"):
        cs_L = np.sqrt(gamma * p_L / rho_L)                # :126-127
        cs_R = np.sqrt(gamma * p_R / rho_R)
    cs_L = np.where(rho_L > FLOOR, cs_L, 0.0)              # :130-133
    cs_R = np.where(rho_R > FLOOR, cs_R, 0.0)
    cs_L = np.where(cs_L == cs_L, cs_L, 0.0)
    cs_R = np.where(cs_R == cs_R, cs_R, 0.0)
    S_L = np.minimum(vn_L - cs_L, vn_R - cs_R)             # :136-139
    S_R = np.maximum(vn_L + cs_L, vn_R + cs_R)
    F = {
        "density": (rho_L * vn_L, rho_R * vn_R, rho_L, rho_R),                                  # :142-143
        "mom_n": (rho_L * np.square(vn_L) + p_L, rho_R * np.square(vn_R) + p_R, mn_L, mn_R),    # :145-146
        "mom_t1": (rho_L * vn_L * vt1_L, rho_R * vn_R * vt1_R, mt1_L, mt1_R),                   # :148-149
        "mom_t2": (rho_L * vn_L * vt2_L, rho_R * vn_R * vt2_R, mt2_L, mt2_R),                   # :151-152
        "energy": ((E_L + p_L) * vn_L, (E_R + p_R) * vn_R, E_L, E_R),                           # :154-155
    }
    d = S_R - S_L                                          # :158-160
    d = np.where(np.abs(d) < 1e-9, 1e-9, d)
    return {k: solve_hll(fl, fr, ul, ur, S_L, S_R, d) for k, (fl, fr, ul, ur) in F.items()}


def hydro_step(g: Gas, dt, cell_size, gamma):
    """GasGrid::hydro_step (src/gas.cpp:250-299): x, y, z sweeps, then primitives."""
    factor = dt / cell_size
    for axis in range(3):
        update_primitive_variables(g, gamma)
        flux = compute_fluxes(g, axis, gamma)
        n, t1, t2 = _AXES[axis]
        sh = {k: np.roll(v, 1, axis) for k, v in flux.items()}                                  # :163-168
        g.density = g.density - factor * (flux["density"] - sh["density"])                      # :277-278
        g.density = np.where(g.density < FLOOR, FLOOR, g.density)                               # :281
        for key, c in (("mom_n", n), ("mom_t1", t1), ("mom_t2", t2)):                           # :283-290
            name = "momentum_" + c
            setattr(g, name, getattr(g, name) - factor * (flux[key] - sh[key]))
        g.energy = g.energy - factor * (flux["energy"] - sh["energy"])                          # :292-293
    update_primitive_variables(g, gamma)


def cfl_timestep(g: Gas, cell_size, gamma, cfl_safety_factor):
    """GasGrid::get_cfl_timestep (src/gas.cpp:301-323)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        cs_sq = gamma * g.pressure / g.density
    cs_sq = np.where(g.density > FLOOR, cs_sq, 0.0)
    cs_sq = np.where(cs_sq == cs_sq, cs_sq, 0.0)
    v_mag = np.sqrt(np.square(g.velocity_x) + np.square(g.velocity_y) + np.square(g.velocity_z))
    signal = v_mag + np.sqrt(cs_sq)
    m = float(np.max(signal))
    if m < 1e-9:
        return math.inf
    return (cell_size / m) * cfl_safety_factor


def apply_gas_kick(g: Gas, grav, dt, a, H, gamma):
    """apply_gas_kick (src/integrator.cpp:134-166); `grav` = the three mesh accelerations."""
    update_primitive_variables(g, gamma)
    a3 = a * a * a
    vel = (g.velocity_x, g.velocity_y, g.velocity_z)
    total = [(gr / a3) - (2 * H * v) for gr, v in zip(grav, vel)]                               # :144-149
    src = [g.density * t for t in total]                                                        # :153-155
    power = vel[0] * src[0] + vel[1] * src[1] + vel[2] * src[2]                                 # :158-160
    g.momentum_x = g.momentum_x + src[0] * dt                                                   # :162-165
    g.momentum_y = g.momentum_y + src[1] * dt
    g.momentum_z = g.momentum_z + src[2] * dt
    g.energy = g.energy + power * dt


def internal_energy_from_temperature(T_kelvin, hubble_param, box_size_mpc, domain_size, gamma):
    """get_internal_energy_from_temp_k (src/ics.cpp:218-239)."""
    k_B, m_p, mu = 1.380649e-23, 1.67262192e-27, 1.22
    u = (k_B * T_kelvin) / ((gamma - 1.0) * mu * m_p)
    H0 = hubble_param * 100.0
    code_unit_mpc = box_size_mpc / domain_size
    v_unit_m_s = 1.5 * H0 * code_unit_mpc * 1000.0
    return u / (v_unit_m_s * v_unit_m_s)


def initialize_gas(dm_rho, disp, growth_f, hubble_rate, gas_total_mass, total_dm_mass, gamma,
                   initial_temperature_k=50.0, hubble_param=0.7, box_size_mpc=16.0, domain_size=1.0,
                   standing=False):
    """initialize_gas (src/ics.cpp:241-288): gas density = dark-matter density scaled to the gas
    mass (floored), velocities from the displacement field on the mesh, uniform temperature."""
    M = dm_rho.shape[0]
    g = Gas(M)
    rho = dm_rho * (gas_total_mass / total_dm_mass)
    g.density = np.where(rho < FLOOR, FLOOR, rho)
    u0 = internal_energy_from_temperature(initial_temperature_k, hubble_param, box_size_mpc,
                                          domain_size, gamma)
    v = [np.zeros((M,) * 3) if standing else hubble_rate * d * growth_f for d in disp]
    g.velocity_x, g.velocity_y, g.velocity_z = v
    g.momentum_x, g.momentum_y, g.momentum_z = (g.density * c for c in v)
    kin = 0.5 * g.density * (v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
    g.energy = (g.density * u0) + kin
    update_primitive_variables(g, gamma)
    return g

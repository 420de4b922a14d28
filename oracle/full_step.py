"""One whole step of the reference (dark matter + gas, mesh force + short range), composed from
the oracle's pieces in the order of initialize_state (src/ics.cpp:290-349), KDK_step
(src/integrator.cpp:215-275) and the engine's time-step choice (src/engine.cpp:16-22, :47-51).

TEST INFRASTRUCTURE ONLY (see pp_oracle.h).  Checked against the accelerations a build of the
reference printed for the shipped simulation.ini before and after its first step
(tests/golden/reference_kat.json).
"""
from __future__ import annotations

import math

import numpy as np

from . import gas as G
from . import pm_mesh
from .oracle import EXACT_CELLS, Oracle
from .zeldovich import cosmology, zeldovich_particles


class Settings:
    """The switches and scalars of the reference's Config that a step reads (src/config.h)."""

    def __init__(self, oracle: Oracle, mesh_size=32, n_per_side=32, omega_baryon=0.045, omega_m=0.3,
                 omega_lambda=0.7, domain_size=1.0, start_a=0.02, gamma=5.0 / 3.0, use_pm=True,
                 use_pp=True, use_hydro=True, expanding=True, cfl_safety_factor=0.5,
                 gravity_dt_factor=0.1, seed=42, initial_gas_temperature_k=50.0, hubble_param=0.7,
                 box_size_mpc=16.0, dt_factor=1e-3):
        self.__dict__.update(locals())
        del self.__dict__["self"]
        kw = dict(domain_size=domain_size, mesh_size=mesh_size, n_per_side=n_per_side,
                  omega_baryon=omega_baryon, omega_m=omega_m, dt_factor=dt_factor)
        self.derived = oracle.config_derive(**kw)
        self.pp = oracle.params_from_config(use_pm, **kw)


class State:
    pass


def _forces(s: Settings, st: State):
    """bin_and_assign_mass .. compute_pp_forces of src/integrator.cpp:236-269 / src/ics.cpp:300-346."""
    d = s.derived
    rho, idx, weights = pm_mesh.cic_deposit(st.pos, st.mass, s.mesh_size, s.domain_size)
    st.dm_rho = rho
    total = rho + st.gas.density if s.use_hydro else rho                       # integrator.cpp:57-59
    st.grav = pm_mesh.poisson_accelerations(total, s.domain_size, d["G"])
    if s.use_pm:
        st.acc = [np.ascontiguousarray(a) for a in pm_mesh.cic_interpolate(st.grav, idx, weights)]
    else:
        st.acc = [np.zeros(st.mass.size) for _ in range(3)]
    if s.use_pp:
        s.oracle.accumulate(s.pp, *st.pos, st.mass, acc=st.acc,
                            traversal=EXACT_CELLS if s.use_pm else 0)
    st.max_accel_sq = s.oracle.max_accel_sq(st.acc)


def initialize_state(s: Settings) -> State:
    st = State()
    st.total_time = 0.0
    st.a, st.H = cosmology(0.0, s.start_a, s.omega_m, s.omega_lambda, s.expanding)
    pos, vel, info = zeldovich_particles(n_per_side=s.n_per_side, mesh_size=s.mesh_size, seed=s.seed,
                                         domain_size=s.domain_size, box_size_mpc=s.box_size_mpc,
                                         omega_m=s.omega_m, omega_lambda=s.omega_lambda,
                                         hubble_param=s.hubble_param, start_a=s.start_a, oracle=s.oracle)
    st.pos, st.vel = pos, vel
    n = pos[0].size
    st.mass = np.full(n, s.derived["dm_particle_mass"])
    st.gas = G.Gas(s.mesh_size)
    if s.use_hydro:
        rho, _, _ = pm_mesh.cic_deposit(pos, st.mass, s.mesh_size, s.domain_size)
        st.gas = G.initialize_gas(rho, info["displacement"], info["growth_f"], st.H,
                                  s.derived["gas_total_mass"], n * s.derived["dm_particle_mass"], s.gamma,
                                  s.initial_gas_temperature_k, s.hubble_param, s.box_size_mpc, s.domain_size)
    _forces(s, st)
    return st


def timestep(s: Settings, st: State):
    """src/engine.cpp:16-22: min(CFL, gravity, FIXED_DT)."""
    cell = s.derived["cell_size"]
    dt_cfl = G.cfl_timestep(st.gas, cell, s.gamma, s.cfl_safety_factor) if s.use_hydro else math.inf
    dt_grav = s.oracle.gravity_timestep(st.mass.size, s.derived["softening_squared"], st.max_accel_sq,
                                        s.gravity_dt_factor)
    return min(dt_cfl, dt_grav, s.derived["fixed_dt"])


def kdk_step(s: Settings, st: State, dt):
    """src/integrator.cpp:215-275."""
    st.a, st.H = cosmology(st.total_time, s.start_a, s.omega_m, s.omega_lambda, s.expanding)
    if s.use_hydro:
        G.apply_gas_kick(st.gas, st.grav, dt / 2.0, st.a, st.H, s.gamma)
    s.oracle.kick(st.vel, st.acc, dt / 2.0, st.a, st.H)
    s.oracle.drift(st.pos, st.vel, dt, s.domain_size)
    if s.use_hydro:
        G.hydro_step(st.gas, dt, s.derived["cell_size"], s.gamma)
    st.total_time += dt
    st.a, st.H = cosmology(st.total_time, s.start_a, s.omega_m, s.omega_lambda, s.expanding)
    _forces(s, st)
    if s.use_hydro:
        G.apply_gas_kick(st.gas, st.grav, dt / 2.0, st.a, st.H, s.gamma)
    s.oracle.kick(st.vel, st.acc, dt / 2.0, st.a, st.H)

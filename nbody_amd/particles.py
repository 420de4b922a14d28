"""Python mirror of the reference's Config / ParticleSystem seam for the PP path.

Same names and argument meaning as the reference (src/config.h, src/particles.h), so the
parity tests read like tests/test_gravity.cpp.  All force work goes through libnbpp; this
module only holds host arrays and scalars.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

from . import ffi


@dataclass
class Config:
    """Fields and defaults of reference src/config.h:5-50 that the gravity step reads."""
    DOMAIN_SIZE: float = 1.0
    MESH_SIZE: int = 32
    OMEGA_BARYON: float = 0.045
    OMEGA_M: float = 0.3
    N_PER_SIDE: int = 32
    STANDING_PARTICLES: bool = False
    USE_PM: bool = True
    USE_PP: bool = True
    CUTOFF_RADIUS_CELLS: float = 2.5
    CUTOFF_TRANSITION_WIDTH_FACTOR: float = 0.2
    DT_FACTOR: float = 1e-3
    GRAVITY_DT_FACTOR: float = 0.2
    USE_HYDRO: bool = True
    GAMMA: float = 5.0 / 3.0
    CFL_SAFETY_FACTOR: float = 0.5
    # derived (src/config.h:43-50)
    G: float = 0.0
    CELL_SIZE: float = 0.0
    CELL_VOLUME: float = 0.0
    OMEGA_DM: float = 0.0
    CUTOFF_RADIUS: float = 0.0
    CUTOFF_RADIUS_SQUARED: float = 0.0
    CUTOFF_TRANSITION_WIDTH: float = 0.0
    R_SWITCH_START: float = 0.0
    R_SWITCH_START_SQ: float = 0.0
    NUM_DM_PARTICLES: int = 0
    DM_PARTICLE_MASS: float = 0.0
    GAS_TOTAL_MASS: float = 0.0
    SOFTENING_SQUARED: float = 0.0
    FIXED_DT: float = 0.0

    def __post_init__(self):
        self.compute_derived_data()  # the reference's constructor does (src/config.cpp:8)

    def compute_derived_data(self):
        """src/config.cpp:73-99, statement by statement."""
        self.CELL_SIZE = self.DOMAIN_SIZE / self.MESH_SIZE
        self.CELL_VOLUME = self.CELL_SIZE * self.CELL_SIZE * self.CELL_SIZE
        self.OMEGA_DM = self.OMEGA_M - self.OMEGA_BARYON
        self.CUTOFF_RADIUS = self.CUTOFF_RADIUS_CELLS * self.CELL_SIZE
        self.CUTOFF_RADIUS_SQUARED = self.CUTOFF_RADIUS * self.CUTOFF_RADIUS
        self.CUTOFF_TRANSITION_WIDTH = self.CUTOFF_TRANSITION_WIDTH_FACTOR * self.CUTOFF_RADIUS
        self.R_SWITCH_START = self.CUTOFF_RADIUS - self.CUTOFF_TRANSITION_WIDTH
        self.R_SWITCH_START_SQ = self.R_SWITCH_START * self.R_SWITCH_START
        self.NUM_DM_PARTICLES = self.N_PER_SIDE * self.N_PER_SIDE * self.N_PER_SIDE
        total_mass = 1.0
        self.G = math.pow(self.DOMAIN_SIZE, 3) / (6.0 * math.pi * total_mass)
        baryon_fraction = self.OMEGA_BARYON / self.OMEGA_M
        dm_total_mass = total_mass * (1.0 - baryon_fraction)
        self.DM_PARTICLE_MASS = dm_total_mass / self.NUM_DM_PARTICLES
        self.GAS_TOTAL_MASS = total_mass * baryon_fraction
        spacing = self.DOMAIN_SIZE / np.cbrt(float(self.NUM_DM_PARTICLES))
        self.SOFTENING_SQUARED = math.pow(spacing / 50.0, 2)
        dynamical_time = 1.0 / math.sqrt(self.G)
        self.FIXED_DT = self.DT_FACTOR * dynamical_time

    def pp_params(self) -> ffi.Params:
        """The scalars compute_pp_forces hoists (src/particles.cpp:143-152)."""
        p = ffi.Params()
        p.domain_size = self.DOMAIN_SIZE
        p.G = self.G
        p.soft_sq = self.SOFTENING_SQUARED
        p.pm_soft_sq = (0.5 * self.CELL_SIZE) * (0.5 * self.CELL_SIZE)
        p.r_switch = self.R_SWITCH_START
        p.r_switch_sq = self.R_SWITCH_START_SQ
        p.cutoff_sq = self.CUTOFF_RADIUS_SQUARED
        p.trans_width = self.CUTOFF_TRANSITION_WIDTH
        p.cell_size = self.CELL_SIZE
        p.mesh_size = self.MESH_SIZE
        p.use_pm = 1 if self.USE_PM else 0
        return p

    def gas_params(self) -> ffi.GasParams:
        """What the gas solver reads from the configuration (src/gas.cpp:191, :251, :322)."""
        p = ffi.GasParams()
        p.cell_size = self.CELL_SIZE
        p.gamma = self.GAMMA
        p.cfl_safety_factor = self.CFL_SAFETY_FACTOR
        p.mesh_size = self.MESH_SIZE
        return p

    def mesh_params(self) -> ffi.MeshParams:
        """What the mesh solve reads from the configuration (src/integrator.cpp:48, :91, :100)."""
        p = ffi.MeshParams()
        p.domain_size = self.DOMAIN_SIZE
        p.G = self.G
        p.mesh_size = self.MESH_SIZE
        return p


class ParticleSystem:
    """Host SoA particle state (src/particles.h:26-72) whose compute_pp_forces runs on the GPU."""

    def __init__(self, config: Config, library: ffi.Library | None = None):
        self._cap = max(int(config.NUM_DM_PARTICLES), 16)
        self.num_particles = 0
        self.max_accel_sq = 0.0
        for name in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "acc_x", "acc_y",
                     "acc_z", "mass"):
            setattr(self, "_" + name, np.zeros(self._cap))
        self._lib = library

    def _view(self, name):
        return getattr(self, "_" + name)[: self.num_particles]

    pos_x = property(lambda s: s._view("pos_x"))
    pos_y = property(lambda s: s._view("pos_y"))
    pos_z = property(lambda s: s._view("pos_z"))
    vel_x = property(lambda s: s._view("vel_x"))
    vel_y = property(lambda s: s._view("vel_y"))
    vel_z = property(lambda s: s._view("vel_z"))
    acc_x = property(lambda s: s._view("acc_x"))
    acc_y = property(lambda s: s._view("acc_y"))
    acc_z = property(lambda s: s._view("acc_z"))
    mass = property(lambda s: s._view("mass"))

    def add_particle(self, px, py, pz, vx, vy, vz, m):
        """src/particles.cpp:32-45 (accelerations start at zero)."""
        if self.num_particles == self._cap:
            self._cap *= 2
            for name in ("pos_x", "pos_y", "pos_z", "vel_x", "vel_y", "vel_z", "acc_x", "acc_y",
                         "acc_z", "mass"):
                a = getattr(self, "_" + name)
                setattr(self, "_" + name, np.concatenate([a, np.zeros(self._cap - a.size)]))
        i = self.num_particles
        for name, v in (("pos_x", px), ("pos_y", py), ("pos_z", pz), ("vel_x", vx), ("vel_y", vy),
                        ("vel_z", vz), ("acc_x", 0.0), ("acc_y", 0.0), ("acc_z", 0.0),
                        ("mass", m)):
            getattr(self, "_" + name)[i] = v
        self.num_particles += 1

    def set_particles(self, x, y, z, m):
        n = len(x)
        self._cap = max(n, 16)
        self.num_particles = n
        for name, v in (("pos_x", x), ("pos_y", y), ("pos_z", z), ("mass", m)):
            setattr(self, "_" + name, np.ascontiguousarray(v, dtype=np.float64).copy())
        for name in ("vel_x", "vel_y", "vel_z", "acc_x", "acc_y", "acc_z"):
            setattr(self, "_" + name, np.zeros(n))

    def bin_and_assign_mass(self, config: Config):
        """The reference builds its linked list here (src/particles.cpp:94-96); libnbpp bins on
        the GPU inside compute_pp_forces, so this is a no-op kept for call-site parity."""

    def compute_pp_forces(self, config: Config):
        """acc += PP / P3M short-range force (src/particles.cpp:133-367), on the GPU."""
        lib = self._lib or ffi.load(device=-1)
        n = self.num_particles
        arrs = [np.ascontiguousarray(self._view(k)) for k in ("pos_x", "pos_y", "pos_z", "mass")]
        acc = [np.ascontiguousarray(self._view(k)).copy() for k in ("acc_x", "acc_y", "acc_z")]
        lib.accumulate(config.pp_params(), *arrs, *acc)
        for k, a in zip(("acc_x", "acc_y", "acc_z"), acc):
            getattr(self, "_" + k)[:n] = a

    def compute_mesh_forces(self, config: Config, mesh_library: ffi.MeshLibrary | None = None):
        """acc = mesh (PM) acceleration, on the GPU: the sequence bin_and_assign_mass ->
        compute_gravitational_acceleration -> interpolate_cic_forces of the reference's step
        (src/integrator.cpp:239-245) for dark matter alone (USE_HYDRO off)."""
        lib = mesh_library or ffi.load_mesh(device=0)
        n = self.num_particles
        arrs = [np.ascontiguousarray(self._view(k)) for k in ("pos_x", "pos_y", "pos_z", "mass")]
        acc = [np.zeros(n) for _ in range(3)]
        lib.accelerations(config.mesh_params(), *arrs, *acc)
        for k, a in zip(("acc_x", "acc_y", "acc_z"), acc):
            getattr(self, "_" + k)[:n] = a

    def get_gravity_timestep(self, config: Config):
        """src/particles.cpp:386-391, as written."""
        if self.num_particles == 0:
            return math.inf
        return math.sqrt(config.SOFTENING_SQUARED) / math.sqrt(self.max_accel_sq) * \
            config.GRAVITY_DT_FACTOR

"""The mesh (PM) side of the reference's force, restated in numpy: CIC mass assignment,
FFT Poisson solve, central differences, CIC interpolation (src/particles.cpp:47-131,
src/integrator.cpp:47-132).

TEST INFRASTRUCTURE ONLY (see pp_oracle.h).  Its one use is to reproduce the total initial
acceleration (mesh + short range) that a build of the reference printed for the shipped
simulation.ini, so that the short-range restatement is checked against reference OUTPUT and not
only against hand-computable cases.
"""
from __future__ import annotations

import math

import numpy as np


def cic_deposit(pos, mass, mesh_size, domain_size):
    """dm_rho of bin_and_assign_mass (src/particles.cpp:47-100); grid index (ix, iy, iz) with
    z fastest (src/types.h:17-19).  Returns (rho, (ix, iy, iz), weights[8])."""
    M = mesh_size
    cell = domain_size / M
    s = [p / cell for p in pos]
    idx = [np.trunc(v).astype(np.int64) for v in s]                 # :60-62
    fr = [v - i for v, i in zip(s, idx)]                            # :64-66
    rho = np.zeros((M, M, M))
    weights = []
    for cz in (0, 1):                                               # w000,w100,w010,w110,w001,...
        for cy in (0, 1):
            for cx in (0, 1):
                w = (fr[0] if cx else 1 - fr[0]) * (fr[1] if cy else 1 - fr[1]) * \
                    (fr[2] if cz else 1 - fr[2])                    # :69-76
                weights.append(((cx, cy, cz), w))
                np.add.at(rho, ((idx[0] + cx) % M, (idx[1] + cy) % M, (idx[2] + cz) % M),
                          mass * w)                                 # :85-92
    return rho / cell ** 3, idx, weights                            # :99


def poisson_accelerations(total_rho, domain_size, G):
    """gravity_x/y/z of compute_gravitational_acceleration (src/integrator.cpp:61-131)."""
    M = total_rho.shape[0]
    rho_k = np.fft.rfftn(total_rho, axes=(0, 1, 2))                 # :73-74
    f = np.where(np.arange(M) < M // 2, np.arange(M), np.arange(M) - M).astype(np.float64)
    kx = f[:, None, None] * 2.0 * math.pi / domain_size             # :87-93
    ky = f[None, :, None] * 2.0 * math.pi / domain_size
    kz = np.arange(M // 2 + 1, dtype=np.float64)[None, None, :] * 2.0 * math.pi / domain_size
    k2 = kx * kx + ky * ky + kz * kz
    k2[0, 0, 0] = 1.0
    phi_k = rho_k * (-4.0 * math.pi * G / k2)                       # :100
    phi_k[0, 0, 0] = 0.0                                            # :82-85
    phi = np.fft.irfftn(phi_k, s=(M, M, M), axes=(0, 1, 2)) * float(M) ** 3   # :105-106
    cell = domain_size / M
    factor = -(1.0 / float(M) ** 3) / (2.0 * cell)                  # :108-109
    return [(np.roll(phi, -1, axis=a) - np.roll(phi, 1, axis=a)) * factor for a in range(3)]  # :126-128


def cic_interpolate(grids, idx, weights):
    """interpolate_cic_forces (src/particles.cpp:102-131): returns acc_x, acc_y, acc_z."""
    M = grids[0].shape[0]
    out = []
    for g in grids:
        a = np.zeros(idx[0].shape)
        for (cx, cy, cz), w in weights:
            a = a + g[(idx[0] + cx) % M, (idx[1] + cy) % M, (idx[2] + cz) % M] * w
        out.append(a)
    return out


def initial_mesh_accelerations(pos, mass, mesh_size, domain_size, G, gas_total_mass=0.0):
    """The PM acceleration initialize_state leaves in acc before the PP call
    (src/ics.cpp:300-311).  With hydro on, the initial gas density is the dark-matter density
    scaled to the gas mass (src/ics.cpp:249-256) and adds to the source (src/integrator.cpp:57-59)."""
    rho, idx, weights = cic_deposit(pos, mass, mesh_size, domain_size)
    total = rho
    if gas_total_mass > 0.0:
        ratio = gas_total_mass / (mass.size * mass[0])
        gas = rho * ratio
        gas = np.where(gas < 1e-12, 1e-12, gas)
        total = rho + gas
    grids = poisson_accelerations(total, domain_size, G)
    return cic_interpolate(grids, idx, weights)

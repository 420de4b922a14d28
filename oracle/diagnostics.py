"""Diagnostics::update_physics (src/diagnostics.cpp:47-129) restated in numpy, for a state of
oracle/full_step.py.

TEST INFRASTRUCTURE ONLY (see pp_oracle.h).  Parity unpinned: the reference's files hold no
output of these quantities, so this restatement is checked only against what follows from the
configuration (total masses) and is otherwise the yardstick for the product's own diagnostics.
"""
from __future__ import annotations

import numpy as np


def update_physics(s, st, enable_energy=True):
    """s: full_step.Settings, st: full_step.State after a force evaluation.  Returns the columns
    of the reference's diagnostics.csv that depend on the physical state."""
    d = s.derived
    a = st.a
    out = {"sim_time": st.total_time, "scale_factor": a}
    m, v = st.mass, st.vel
    out["total_mass_dm"] = float(np.sum(m))                                             # :69-79
    out["total_momentum_dm"] = [float(np.sum(m * c)) for c in v]
    out["ke_dm"] = out["pe_total"] = 0.0
    if enable_energy:                                                                   # :87-100
        out["ke_dm"] = float(np.sum(0.5 * m * ((a * v[0]) ** 2 + (a * v[1]) ** 2 + (a * v[2]) ** 2)))
        total_rho = st.dm_rho + st.gas.density if s.use_hydro else st.dm_rho
        phi = potential(total_rho, s.domain_size, d["G"])
        out["pe_total"] = float(np.sum(0.5 * total_rho * phi * d["cell_volume"])) / a
    if s.use_hydro:                                                                     # :58-66, :103-128
        g = st.gas
        out["max_gas_density"] = float(np.max(g.density))
        out["max_gas_pressure"] = float(np.max(g.pressure))
        out["max_gas_velocity"] = float(np.max(np.sqrt(g.velocity_x ** 2 + g.velocity_y ** 2 + g.velocity_z ** 2)))
        out["total_mass_gas"] = float(np.sum(g.density)) * d["cell_volume"]
        out["total_momentum_gas"] = [float(np.sum(c)) for c in (g.momentum_x, g.momentum_y, g.momentum_z)]
        ke = 0.5 * (g.momentum_x ** 2 + g.momentum_y ** 2 + g.momentum_z ** 2)
        ke = np.where(g.density > 1e-12, ke / g.density, 0.0)
        out["ke_gas"] = float(np.sum(ke)) * d["cell_volume"]
        out["ie_gas"] = float(np.sum(g.pressure / (s.gamma - 1.0))) * d["cell_volume"]
    return out


def potential(total_rho, domain_size, G):
    """state.phi: the unnormalised inverse transform of -4 pi G rho_k / k^2 (src/integrator.cpp:73-106)."""
    import math
    M = total_rho.shape[0]
    rho_k = np.fft.rfftn(total_rho, axes=(0, 1, 2))
    f = np.where(np.arange(M) < M // 2, np.arange(M), np.arange(M) - M).astype(np.float64)
    kx = f[:, None, None] * 2.0 * math.pi / domain_size
    ky = f[None, :, None] * 2.0 * math.pi / domain_size
    kz = np.arange(M // 2 + 1, dtype=np.float64)[None, None, :] * 2.0 * math.pi / domain_size
    k2 = kx * kx + ky * ky + kz * kz
    k2[0, 0, 0] = 1.0
    phi_k = rho_k * (-4.0 * math.pi * G / k2)
    phi_k[0, 0, 0] = 0.0
    return np.fft.irfftn(phi_k, s=(M, M, M), axes=(0, 1, 2)) * float(M) ** 3

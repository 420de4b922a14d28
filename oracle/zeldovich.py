"""Zel'dovich initial conditions as the reference generates them (src/ics.cpp:36-216,
src/integrator.cpp:9-45): the particle sets on which parity is judged.

TEST INFRASTRUCTURE ONLY (see pp_oracle.h).  White noise comes from the C restatement of
libstdc++'s minstd_rand0 + normal_distribution stream; the FFTs are numpy's (pocketfft, like
the reference).  Pinned by the particle-0 anchor values that a build of the reference produced
for the shipped simulation.ini (tests/golden/reference_kat.json: "default_ini_anchor").
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from .oracle import Oracle, _dp


def cosmology(total_time, start_a=0.02, omega_m=0.3, omega_lambda=0.7, expanding=True):
    """(scale factor, Hubble rate) in code units: src/integrator.cpp:9-45."""
    if not expanding:
        return 1.0, 0.0
    if omega_lambda == 0.0:
        t = math.pow(start_a, 1.5) + total_time
        return math.pow(t, 2.0 / 3.0), (2.0 / 3.0) / t
    Om, Ol = omega_m, omega_lambda
    H0 = 2.0 / (3.0 * math.sqrt(Om))
    factor = math.sqrt(Ol / Om)
    t_start = (2.0 / (3.0 * H0 * math.sqrt(Ol))) * math.asinh(factor * math.pow(start_a, 1.5))
    t = t_start + total_time
    sinh_term = math.sinh(1.5 * H0 * math.sqrt(Ol) * t)
    a = math.pow(Om / Ol, 1.0 / 3.0) * math.pow(sinh_term, 2.0 / 3.0)
    return a, H0 * math.sqrt(Om / math.pow(a, 3.0) + Ol)


def zeldovich_particles(n_per_side=32, mesh_size=32, seed=42, domain_size=1.0, box_size_mpc=16.0,
                        omega_m=0.3, omega_lambda=0.7, hubble_param=0.7, sigma_8=0.81,
                        spectral_index=0.96, start_a=0.02, oracle: Oracle | None = None):
    """Positions [x, y, z] and velocities [vx, vy, vz] of initialize_dm (src/ics.cpp:167-216) on
    the displacement field of compute_zeldovich_field (src/ics.cpp:36-165).  Particle index is
    (i*N + j)*N + k with i the x lattice index, as the reference's loops."""
    orc = oracle or Oracle()
    L = orc.lib
    L.orc_libstdcxx_normal_field.argtypes = [C.c_uint32, C.c_size_t, C.POINTER(C.c_double)]
    L.orc_sigma8_amplitude.restype = C.c_double
    L.orc_sigma8_amplitude.argtypes = [C.c_double] * 4
    L.orc_bbks_pk_array.argtypes = [C.c_size_t, C.POINTER(C.c_double)] + [C.c_double] * 3 + [
        C.POINTER(C.c_double)]

    a, H = cosmology(0.0, start_a, omega_m, omega_lambda)
    M = mesh_size
    A = L.orc_sigma8_amplitude(sigma_8, omega_m, hubble_param, spectral_index)      # :38-51
    white = np.empty(M * M * M)
    L.orc_libstdcxx_normal_field(seed, white.size, _dp(white))                       # :61-66
    random_k = np.fft.rfftn(white.reshape(M, M, M), axes=(0, 1, 2))                  # :77-79

    freq = np.where(np.arange(M) < M // 2, np.arange(M), np.arange(M) - M).astype(np.float64)
    kxf = freq[:, None, None]
    kyf = freq[None, :, None]
    kzf = np.arange(M // 2 + 1, dtype=np.float64)[None, None, :]
    two_pi = 2.0 * math.pi
    kx, ky, kz = (f * two_pi / box_size_mpc for f in (kxf, kyf, kzf))                # :103-105
    k_mag = np.sqrt(kx * kx + ky * ky + kz * kz)                                     # :106
    k_h = k_mag / hubble_param                                                       # :107
    amplitude_scaling = math.sqrt(float(M) ** 3 / math.pow(box_size_mpc, 3.0))       # :57-59
    flat = np.ascontiguousarray(k_h.ravel())
    pk = np.zeros_like(flat)
    L.orc_bbks_pk_array(flat.size, _dp(flat), omega_m, hubble_param, spectral_index, _dp(pk))
    ps_sqrt = (np.sqrt(A * pk) * amplitude_scaling).reshape(k_h.shape)               # :113-114
    delta_k = random_k * ps_sqrt                                                     # :118-119
    ckx, cky, ckz = (f * two_pi / domain_size for f in (kxf, kyf, kzf))              # :123-125
    code_k2 = ckx * ckx + cky * cky + ckz * ckz                                       # :126-127
    code_k2[0, 0, 0] = 1.0
    phi_k = -delta_k / code_k2                                                       # :129
    norm_ic = a / float(M) ** 3                                                      # :150
    disp = []
    for ck in (ckx, cky, ckz):
        d_k = (0 - 1j) * ck * phi_k                                                  # :131-133
        d_k[0, 0, 0] = 0.0                                                           # :88-92
        # unnormalised inverse transform (pocketfft c2r with fct = 1), then norm_ic
        d = np.fft.irfftn(d_k, s=(M, M, M), axes=(0, 1, 2)) * float(M) ** 3          # :143-148
        disp.append(d * norm_ic)                                                     # :152-156
    a3 = math.pow(a, 3.0)
    f_growth = math.pow(omega_m / (omega_m + omega_lambda * a3), 0.55)               # :159-162

    N = n_per_side
    cell_size = domain_size / M
    spacing = domain_size / N
    q = (np.arange(N) + 0.5) * spacing                                               # :177-179
    cell = (q / cell_size).astype(np.int64) % M                                      # :181-183
    ix, iy, iz = np.meshgrid(cell, cell, cell, indexing="ij")
    qx, qy, qz = np.meshgrid(q, q, q, indexing="ij")
    pos, vel = [], []
    for qq, d in zip((qx, qy, qz), disp):
        dd = d[ix, iy, iz]                                                           # :185-191
        pos.append(np.ascontiguousarray(np.fmod(qq + dd + domain_size, domain_size).ravel()))
        vel.append(np.ascontiguousarray((H * dd * f_growth).ravel()))                # :200-208
    return pos, vel, {"scale_factor": a, "hubble_param": H, "growth_f": f_growth, "A": A,
                      "displacement": disp}

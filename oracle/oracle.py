"""ctypes doorway onto the CPU oracle (oracle/pp_oracle.c) and, where it was built, onto
the reference's own config.cpp / math_utils.h (oracle/_ref/libnbref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never by nbody_amd.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

EXACT, REF_OFFLOAD, REF_CPU, EXACT_CELLS = 0, 1, 2, 3


class Params(C.Structure):
    """Field-for-field twin of orc_params / nbpp_params."""
    _fields_ = [(k, C.c_double) for k in (
        "domain_size", "G", "soft_sq", "pm_soft_sq", "r_switch", "r_switch_sq",
        "cutoff_sq", "trans_width", "cell_size")] + [("mesh_size", C.c_int32),
                                                     ("use_pm", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class ConfigIn(C.Structure):
    _fields_ = [("domain_size", C.c_double), ("mesh_size", C.c_int32),
                ("n_per_side", C.c_int32), ("omega_baryon", C.c_double),
                ("omega_m", C.c_double), ("cutoff_radius_cells", C.c_double),
                ("cutoff_transition_width_factor", C.c_double), ("dt_factor", C.c_double)]


class ConfigOut(C.Structure):
    _fields_ = [(k, C.c_double) for k in (
        "cell_size", "cell_volume", "omega_dm", "cutoff_radius", "cutoff_radius_squared",
        "cutoff_transition_width", "r_switch_start", "r_switch_start_sq")] + [
        ("num_dm_particles", C.c_int32), ("pad", C.c_int32)] + [(k, C.c_double) for k in (
            "G", "dm_particle_mass", "gas_total_mass", "softening_squared", "fixed_dt")]


class RefConfigView(C.Structure):
    """Twin of ref_config_view in ref_shim.cpp."""
    _fields_ = [("domain_size", C.c_double), ("mesh_size", C.c_int), ("n_per_side", C.c_int)] + [
        (k, C.c_double) for k in ("omega_baryon", "omega_m", "cutoff_radius_cells",
                                  "cutoff_transition_width_factor", "dt_factor",
                                  "gravity_dt_factor")] + [
        (k, C.c_int) for k in ("use_pm", "use_pp", "standing_particles", "seed")] + [
        (k, C.c_double) for k in ("cell_size", "cell_volume", "omega_dm", "cutoff_radius",
                                  "cutoff_radius_squared", "cutoff_transition_width",
                                  "r_switch_start", "r_switch_start_sq")] + [
        ("num_dm_particles", C.c_int), ("pad", C.c_int)] + [
        (k, C.c_double) for k in ("G", "dm_particle_mass", "gas_total_mass",
                                  "softening_squared", "fixed_dt")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "pad"}


_DP = C.POINTER(C.c_double)


def _dp(a):
    return a.ctypes.data_as(_DP)


def _params(p):
    """Accept any ctypes struct with the nbpp_params layout (e.g. nbody_amd.ffi.Params)."""
    if isinstance(p, Params):
        return p
    if C.sizeof(p) != C.sizeof(Params):
        raise TypeError("parameter struct has the wrong size")
    return Params.from_buffer_copy(bytes(p))


def build(quiet=True):
    """Compile the oracle (and _ref where the reference tree exists) with oracle/Makefile."""
    subprocess.run(["make", "-C", HERE], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _cpu_has(*flags):
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    have = set(line.split(":", 1)[1].split())
                    return all(fl in have for fl in flags)
    except OSError:
        pass
    return False


class Oracle:
    def __init__(self, fast=False):
        name = "libpp_oracle_fast.so" if fast else "libpp_oracle.so"
        if fast and not _cpu_has("avx2", "fma", "bmi2"):
            name = "libpp_oracle.so"
        path = os.path.join(HERE, "_build", name)
        if not os.path.exists(path):
            build()
        self.path = path
        self.fast = name.endswith("_fast.so")
        L = self.lib = C.CDLL(path)
        L.orc_periodic_displacement.restype = C.c_double
        L.orc_periodic_displacement.argtypes = [C.c_double, C.c_double]
        L.orc_config_derive.argtypes = [C.POINTER(ConfigIn), C.POINTER(ConfigOut)]
        L.orc_params_from_config.argtypes = [C.POINTER(ConfigIn), C.c_int, C.POINTER(Params)]
        sig = [C.POINTER(Params), C.c_size_t, _DP, _DP, _DP, _DP, C.c_size_t, C.c_size_t,
               _DP, _DP, _DP]
        L.orc_pp_accumulate.argtypes = sig + [C.c_int, C.c_int]
        L.orc_pp_accumulate_ld.argtypes = sig + [C.c_int]
        L.orc_count_within_cutoff.restype = C.c_uint64
        L.orc_count_within_cutoff.argtypes = [C.POINTER(Params), C.c_size_t, _DP, _DP, _DP,
                                              C.c_size_t, C.c_size_t, C.c_int]
        L.orc_kick.argtypes = [C.c_size_t] + [_DP] * 6 + [C.c_double] * 3
        L.orc_drift.argtypes = [C.c_size_t] + [_DP] * 6 + [C.c_double] * 2
        L.orc_max_accel_sq.restype = C.c_double
        L.orc_max_accel_sq.argtypes = [C.c_size_t, _DP, _DP, _DP]
        L.orc_gravity_timestep.restype = C.c_double
        L.orc_gravity_timestep.argtypes = [C.c_size_t, C.c_double, C.c_double, C.c_double]
        L.orc_max_threads.restype = C.c_int

    # -- scalars ---------------------------------------------------------------------
    def periodic_displacement(self, dx, L):
        return self.lib.orc_periodic_displacement(dx, L)

    def config_derive(self, **kw):
        cin = default_config_in(**kw)
        out = ConfigOut()
        self.lib.orc_config_derive(C.byref(cin), C.byref(out))
        return {k: getattr(out, k) for k, _ in ConfigOut._fields_ if k != "pad"}

    def params_from_config(self, use_pm, **kw):
        cin = default_config_in(**kw)
        p = Params()
        self.lib.orc_params_from_config(C.byref(cin), int(use_pm), C.byref(p))
        return p

    def max_threads(self):
        return self.lib.orc_max_threads()

    # -- forces ----------------------------------------------------------------------
    def accumulate(self, p, x, y, z, m, acc=None, tgt_begin=0, n_tgt=None,
                   traversal=EXACT, nthreads=0, long_double=False):
        x, y, z, m = (np.ascontiguousarray(a, dtype=np.float64) for a in (x, y, z, m))
        p = _params(p)
        n = x.size
        if n_tgt is None:
            n_tgt = n - tgt_begin
        if acc is None:
            acc = [np.zeros(n_tgt) for _ in range(3)]
        ax, ay, az = acc
        if long_double:
            rc = self.lib.orc_pp_accumulate_ld(C.byref(p), n, _dp(x), _dp(y), _dp(z), _dp(m),
                                               tgt_begin, n_tgt, _dp(ax), _dp(ay), _dp(az),
                                               nthreads)
        else:
            rc = self.lib.orc_pp_accumulate(C.byref(p), n, _dp(x), _dp(y), _dp(z), _dp(m),
                                            tgt_begin, n_tgt, _dp(ax), _dp(ay), _dp(az),
                                            traversal, nthreads)
        if rc != 0:
            raise ValueError("oracle rejected its arguments")
        return ax, ay, az

    def count_within_cutoff(self, p, x, y, z, tgt_begin=0, n_tgt=None, nthreads=0):
        x, y, z = (np.ascontiguousarray(a, dtype=np.float64) for a in (x, y, z))
        p = _params(p)
        if n_tgt is None:
            n_tgt = x.size - tgt_begin
        return int(self.lib.orc_count_within_cutoff(C.byref(p), x.size, _dp(x), _dp(y), _dp(z),
                                                    tgt_begin, n_tgt, nthreads))

    # -- caller contract ---------------------------------------------------------------
    def kick(self, v, a, dt, scale, hubble):
        self.lib.orc_kick(v[0].size, _dp(v[0]), _dp(v[1]), _dp(v[2]), _dp(a[0]), _dp(a[1]),
                          _dp(a[2]), dt, scale, hubble)

    def drift(self, x, v, dt, L):
        self.lib.orc_drift(x[0].size, _dp(x[0]), _dp(x[1]), _dp(x[2]), _dp(v[0]), _dp(v[1]),
                           _dp(v[2]), dt, L)

    def max_accel_sq(self, a):
        return self.lib.orc_max_accel_sq(a[0].size, _dp(a[0]), _dp(a[1]), _dp(a[2]))

    def gravity_timestep(self, n, soft_sq, max_acc_sq, factor):
        return self.lib.orc_gravity_timestep(n, soft_sq, max_acc_sq, factor)


def default_config_in(domain_size=1.0, mesh_size=32, n_per_side=32, omega_baryon=0.045,
                      omega_m=0.3, cutoff_radius_cells=2.5,
                      cutoff_transition_width_factor=0.2, dt_factor=1e-3):
    """Defaults are the struct defaults of the reference's Config (src/config.h:5-40)."""
    return ConfigIn(domain_size, mesh_size, n_per_side, omega_baryon, omega_m,
                    cutoff_radius_cells, cutoff_transition_width_factor, dt_factor)


class Reference:
    """The reference's own compiled config.cpp + math_utils.h (only where oracle/_ref exists)."""

    def __init__(self):
        path = os.path.join(HERE, "_ref", "libnbref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        L = self.lib = C.CDLL(path)
        L.ref_periodic_displacement.restype = C.c_double
        L.ref_periodic_displacement.argtypes = [C.c_double, C.c_double]
        L.ref_config_defaults.argtypes = [C.POINTER(RefConfigView)]
        L.ref_config_load.argtypes = [C.c_char_p, C.POINTER(RefConfigView)]
        L.ref_config_derive.argtypes = [C.c_double, C.c_int, C.c_int] + [C.c_double] * 5 + [
            C.POINTER(RefConfigView)]

    @staticmethod
    def available():
        return os.path.exists(os.path.join(HERE, "_ref", "libnbref.so"))

    def periodic_displacement(self, dx, L):
        return self.lib.ref_periodic_displacement(dx, L)

    def config_defaults(self):
        v = RefConfigView()
        self.lib.ref_config_defaults(C.byref(v))
        return v.as_dict()

    def config_load(self, path):
        v = RefConfigView()
        if self.lib.ref_config_load(path.encode(), C.byref(v)) != 0:
            raise OSError("reference Config::load failed for " + path)
        return v.as_dict()

    def config_derive(self, domain_size=1.0, mesh_size=32, n_per_side=32, omega_baryon=0.045,
                      omega_m=0.3, cutoff_radius_cells=2.5,
                      cutoff_transition_width_factor=0.2, dt_factor=1e-3):
        v = RefConfigView()
        self.lib.ref_config_derive(domain_size, mesh_size, n_per_side, omega_baryon, omega_m,
                                   cutoff_radius_cells, cutoff_transition_width_factor,
                                   dt_factor, C.byref(v))
        return v.as_dict()

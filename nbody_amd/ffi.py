"""ctypes binding of libnbpp.so, one Python method per entry point of include/nbpp.h."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# NBPP_LIB points at another build of the library (kernel experiments)
LIB_PATH = os.environ.get("NBPP_LIB") or os.path.join(HERE, "lib", "libnbpp.so")

KERNEL_NONE, KERNEL_DIRECT, KERNEL_DIRECT_WRAP, KERNEL_P3M_CELLS, KERNEL_P3M_ALLPAIRS, \
    KERNEL_DIRECT_SYMMETRIC = range(6)
NOALLOC_MAX_N = 262144                 # NBPP_NOALLOC_MAX_N
NOALLOC_SCRATCH_BYTES = 48 * 1024 * 1024  # NBPP_NOALLOC_SCRATCH_BYTES
OPT_FORCE_KERNEL, OPT_TIMING, OPT_TARGETS_PER_LANE, OPT_SOURCE_SPLIT, OPT_P3M_SEARCH_RADIUS, \
    OPT_NO_EQUAL_MASS_PATH, OPT_DETERMINISTIC, OPT_P3M_LANES_PER_TARGET, OPT_P3M_POOLED_DRAIN, OPT_P3M_PAIR_ONCE = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
OPT_P3M_ROW_LANES = 12        # lanes of one target walking every row together: 0 automatic, -1 never, 4 / 8 / 16 / 32 / 64
OPT_P3M_ROWS_POOLED = 13      # the row walk with the pooled drain: 0 automatic, 1 with OPT_P3M_ROW_LANES 2..16, -1 never
OPT_P3M_POOL_WORKGROUP = 11   # threads per workgroup of the pooled P3M kernel: 0 = 64, else 64, 128 or 256

EXPORTS = (
    "nbpp_init", "nbpp_shutdown", "nbpp_last_error", "nbpp_set_option", "nbpp_get_stats",
    "nbpp_accumulate", "nbpp_host_alloc", "nbpp_host_free", "nbpp_accumulate_device",
    "nbpp_accumulate_device_noalloc",
    "nbpp_accumulate_pairs_device", "nbpp_kick_device", "nbpp_drift_device", "nbpp_add_device",
    "nbpp_add_clear_device", "nbpp_kick_drift_device", "nbpp_add_clear_kick_device",
    "nbpp_add_clear_kick_max_device",
    "nbpp_max_accel_sq_device",
)


class NbppError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libnbpp error {code}: {message}")
        self.code = code


class Params(C.Structure):
    """nbpp_params: the scalars hoisted at reference src/particles.cpp:134-152."""
    _fields_ = [(k, C.c_double) for k in (
        "domain_size", "G", "soft_sq", "pm_soft_sq", "r_switch", "r_switch_sq",
        "cutoff_sq", "trans_width", "cell_size")] + [("mesh_size", C.c_int32),
                                                     ("use_pm", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("pair_kernel_ms", "prepare_ms", "total_ms",
                                          "h2d_ms", "d2h_ms")] + [
        (k, C.c_uint64) for k in ("n_sources", "n_targets", "pairs_evaluated",
                                  "pairs_accepted", "pair_launches", "subtiles_total",
                                  "subtiles_wrapped")] + [
        ("kernel_variant", C.c_int32), ("p3m_pooled", C.c_int32), ("cell_occupancy", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_DP = C.POINTER(C.c_double)


def _host(a):
    if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous):
        raise TypeError("host arrays must be C-contiguous float64 numpy arrays")
    return a.ctypes.data_as(_DP)


def _dev(t):
    """Device pointer of a torch CUDA float64 tensor (torch is only plumbing here)."""
    if t.dtype.__str__() != "torch.float64" or not t.is_cuda or not t.is_contiguous():
        raise TypeError("device arrays must be contiguous float64 CUDA tensors")
    return C.cast(C.c_void_p(t.data_ptr()), _DP)


class Library:
    """Loaded libnbpp.so.  Construction fails loudly when the shared object is missing."""

    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found: build it with `make -C nbody_amd/csrc` "
                "(or __graft_entry__.build()); there is no fallback path")
        # torch ships its own libamdhip64.so.7; two HIP runtimes in one process cannot both
        # open the GPU.  Importing torch first makes the dynamic linker resolve libnbpp's
        # DT_NEEDED libamdhip64.so.7 to the copy torch already loaded.  Without torch the
        # system ROCm runtime is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = self.lib = C.CDLL(path)
        self.path = path
        L.nbpp_init.argtypes = [C.c_int]
        L.nbpp_shutdown.restype = None
        L.nbpp_last_error.restype = C.c_char_p
        L.nbpp_set_option.argtypes = [C.c_int, C.c_int64]
        L.nbpp_get_stats.argtypes = [C.POINTER(Stats)]
        L.nbpp_accumulate.argtypes = [C.POINTER(Params), C.c_size_t] + [_DP] * 7
        L.nbpp_host_alloc.argtypes = [C.c_size_t]
        L.nbpp_host_alloc.restype = C.c_void_p
        L.nbpp_host_free.argtypes = [C.c_void_p]
        L.nbpp_host_free.restype = None
        L.nbpp_accumulate_device.argtypes = [C.POINTER(Params), C.c_size_t] + [_DP] * 4 + [
            C.c_size_t, C.c_size_t] + [_DP] * 3 + [C.c_void_p]
        L.nbpp_accumulate_device_noalloc.argtypes = [C.POINTER(Params), C.c_size_t] + [_DP] * 7 + [
            C.c_void_p]
        L.nbpp_accumulate_pairs_device.argtypes = [C.POINTER(Params), C.c_size_t] + [_DP] * 4 + [
            C.c_int, C.c_int] + [_DP] * 3 + [C.c_void_p]
        L.nbpp_kick_device.argtypes = [C.c_size_t] + [_DP] * 6 + [C.c_double] * 3 + [C.c_void_p]
        L.nbpp_add_device.argtypes = [C.c_size_t] + [_DP] * 6 + [C.c_void_p]
        L.nbpp_add_clear_device.argtypes = [C.c_size_t] + [_DP] * 6 + [C.c_void_p]
        L.nbpp_kick_drift_device.argtypes = [C.c_size_t] + [_DP] * 9 + [C.c_double] * 5 + [C.c_void_p]
        L.nbpp_add_clear_kick_device.argtypes = [C.c_size_t] + [_DP] * 9 + [C.c_double] * 3 + [C.c_void_p]
        L.nbpp_add_clear_kick_max_device.argtypes = [C.c_size_t] + [_DP] * 9 + [C.c_double] * 3 + \
            [C.POINTER(C.c_double), C.c_void_p]
        L.nbpp_drift_device.argtypes = [C.c_size_t] + [_DP] * 6 + [C.c_double] * 2 + [C.c_void_p]
        L.nbpp_max_accel_sq_device.argtypes = [C.c_size_t] + [_DP] * 3 + [_DP, C.c_void_p]
        self._inited = False

    def _check(self, rc):
        if rc != 0:
            raise NbppError(rc, self.lib.nbpp_last_error().decode(errors="replace"))

    # lifecycle -------------------------------------------------------------------
    def init(self, device=-1):
        self._check(self.lib.nbpp_init(device))
        self._inited = True
        return self

    def shutdown(self):
        self.lib.nbpp_shutdown()
        self._inited = False

    def set_option(self, key, value):
        self._check(self.lib.nbpp_set_option(key, int(value)))

    def stats(self):
        s = Stats()
        self._check(self.lib.nbpp_get_stats(C.byref(s)))
        return s.as_dict()

    # hot path --------------------------------------------------------------------
    def accumulate(self, p, x, y, z, m, ax, ay, az):
        """Host numpy arrays; ax/ay/az are accumulated into (nbpp_accumulate)."""
        n = x.size
        self._check(self.lib.nbpp_accumulate(C.byref(p), n, _host(x), _host(y), _host(z),
                                             _host(m), _host(ax), _host(ay), _host(az)))

    def host_alloc(self, n):
        """float64[n] numpy view of page-locked host memory (nbpp_host_alloc); release it with
        host_free(array) once nothing refers to it any more."""
        ptr = self.lib.nbpp_host_alloc(max(n, 1) * 8)
        if not ptr:
            raise NbppError(-5, self.lib.nbpp_last_error().decode(errors="replace"))
        return np.ctypeslib.as_array(C.cast(C.c_void_p(ptr), _DP), shape=(n,))

    def host_free(self, a):
        self.lib.nbpp_host_free(C.c_void_p(a.ctypes.data))

    def accumulate_device(self, p, x, y, z, m, ax, ay, az, tgt_begin=0, n_tgt=None,
                          stream=None):
        """torch CUDA tensors (nbpp_accumulate_device)."""
        n_src = x.numel()
        if n_tgt is None:
            n_tgt = n_src - tgt_begin
        if ax.numel() < n_tgt:
            raise ValueError("acceleration arrays shorter than the target count")
        self._check(self.lib.nbpp_accumulate_device(
            C.byref(p), n_src, _dev(x), _dev(y), _dev(z), _dev(m), tgt_begin, n_tgt,
            _dev(ax), _dev(ay), _dev(az), C.c_void_p(stream or 0)))

    def accumulate_device_noalloc(self, p, x, y, z, m, ax, ay, az, stream=None):
        """All targets, working memory from the library's fixed array only
        (nbpp_accumulate_device_noalloc)."""
        n = x.numel()
        if min(ax.numel(), ay.numel(), az.numel()) < n:
            raise ValueError("acceleration arrays must hold all n particles")
        self._check(self.lib.nbpp_accumulate_device_noalloc(
            C.byref(p), n, _dev(x), _dev(y), _dev(z), _dev(m), _dev(ax), _dev(ay), _dev(az),
            C.c_void_p(stream or 0)))

    def accumulate_pairs_device(self, p, x, y, z, m, ax, ay, az, part=0, n_parts=1, stream=None):
        """Share `part` of `n_parts` of all unordered pairs into full-length ax/ay/az
        (nbpp_accumulate_pairs_device)."""
        n = x.numel()
        if min(ax.numel(), ay.numel(), az.numel()) < n:
            raise ValueError("acceleration arrays must hold all n particles")
        self._check(self.lib.nbpp_accumulate_pairs_device(
            C.byref(p), n, _dev(x), _dev(y), _dev(z), _dev(m), part, n_parts,
            _dev(ax), _dev(ay), _dev(az), C.c_void_p(stream or 0)))

    # caller contract ---------------------------------------------------------------
    def add_device(self, a, b, stream=None, clear=False):
        """a += b per component (nbpp_add_device); clear: b = 0 afterwards (nbpp_add_clear_device)."""
        fn = self.lib.nbpp_add_clear_device if clear else self.lib.nbpp_add_device
        self._check(fn(a[0].numel(), _dev(a[0]), _dev(a[1]), _dev(a[2]),
                       _dev(b[0]), _dev(b[1]), _dev(b[2]), C.c_void_p(stream or 0)))

    def kick_drift_device(self, x, v, a, dt_kick, dt_drift, scale_factor, hubble, domain_size, stream=None):
        """kick_device(dt_kick) then drift_device(dt_drift) in one pass (nbpp_kick_drift_device)."""
        self._check(self.lib.nbpp_kick_drift_device(
            x[0].numel(), *[_dev(t) for t in (*x, *v, *a)], dt_kick, dt_drift, scale_factor, hubble, domain_size,
            C.c_void_p(stream or 0)))

    def add_clear_kick_device(self, a, b, v, dt, scale_factor, hubble, stream=None):
        """add_device(clear=True) then kick_device(dt) in one pass (nbpp_add_clear_kick_device)."""
        self._check(self.lib.nbpp_add_clear_kick_device(
            a[0].numel(), *[_dev(t) for t in (*a, *b, *v)], dt, scale_factor, hubble, C.c_void_p(stream or 0)))

    def add_clear_kick_max_device(self, a, b, v, dt, scale_factor, hubble, stream=None):
        """add_clear_kick_device then max_accel_sq_device(a) in one pass (nbpp_add_clear_kick_max_device);
        blocking, returns the maximum."""
        out = C.c_double(0.0)
        self._check(self.lib.nbpp_add_clear_kick_max_device(
            a[0].numel(), *[_dev(t) for t in (*a, *b, *v)], dt, scale_factor, hubble, C.byref(out),
            C.c_void_p(stream or 0)))
        return out.value

    def kick_device(self, v, a, dt, scale_factor, hubble, stream=None):
        self._check(self.lib.nbpp_kick_device(v[0].numel(), _dev(v[0]), _dev(v[1]), _dev(v[2]),
                                              _dev(a[0]), _dev(a[1]), _dev(a[2]), dt,
                                              scale_factor, hubble, C.c_void_p(stream or 0)))

    def drift_device(self, x, v, dt, domain_size, stream=None):
        self._check(self.lib.nbpp_drift_device(x[0].numel(), _dev(x[0]), _dev(x[1]), _dev(x[2]),
                                               _dev(v[0]), _dev(v[1]), _dev(v[2]), dt,
                                               domain_size, C.c_void_p(stream or 0)))

    def max_accel_sq_device(self, a, stream=None):
        out = C.c_double(0.0)
        self._check(self.lib.nbpp_max_accel_sq_device(a[0].numel(), _dev(a[0]), _dev(a[1]),
                                                      _dev(a[2]), C.byref(out),
                                                      C.c_void_p(stream or 0)))
        return out.value


_LIB = None


def load(device=None):
    """Process-wide Library; `device` given -> also nbpp_init(device)."""
    global _LIB
    if _LIB is None:
        _LIB = Library()
    if device is not None and not _LIB._inited:
        _LIB.init(device)
    return _LIB


# ---------------------------------------------------------------------------------------
# libnbpm.so: the mesh (long-range) side, include/nbpm.h
# ---------------------------------------------------------------------------------------
MESH_LIB_PATH = os.environ.get("NBPM_LIB") or os.path.join(HERE, "lib", "libnbpm.so")

MESH_EXPORTS = (
    "nbpm_init", "nbpm_shutdown", "nbpm_last_error", "nbpm_set_timing", "nbpm_get_stats",
    "nbpm_deposit_device", "nbpm_solve_device", "nbpm_potential_device", "nbpm_interpolate_device",
    "nbpm_accelerations_device", "nbpm_accelerations",
    "nbpm_sigma8_amplitude", "nbpm_zeldovich_field_device", "nbpm_zeldovich_particles_device",
)


class MeshParams(C.Structure):
    """nbpm_params."""
    _fields_ = [("domain_size", C.c_double), ("G", C.c_double), ("mesh_size", C.c_int32),
                ("reserved", C.c_int32)]


class IcParams(C.Structure):
    """nbpm_ic_params."""
    _fields_ = [("mesh_size", C.c_int32), ("n_per_side", C.c_int32)] + [
        (k, C.c_double) for k in ("domain_size", "box_size_mpc", "hubble_param", "omega_m", "spectral_index",
                                  "amplitude", "scale_factor")]


class MeshStats(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("deposit_ms", "solve_ms", "interpolate_ms", "total_ms")]


class MeshLibrary:
    """Loaded libnbpm.so.  Construction fails loudly when the shared object is missing."""

    def __init__(self, path=MESH_LIB_PATH):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found: build it with `make -C nbody_amd/csrc` "
                "(or __graft_entry__.build()); there is no fallback path")
        try:  # same reason as in Library: one HIP runtime per process
            import torch  # noqa: F401
        except ImportError:
            pass
        L = self.lib = C.CDLL(path)
        self.path = path
        L.nbpm_init.argtypes = [C.c_int]
        L.nbpm_shutdown.restype = None
        L.nbpm_last_error.restype = C.c_char_p
        L.nbpm_set_timing.argtypes = [C.c_int]
        L.nbpm_get_stats.argtypes = [C.POINTER(MeshStats)]
        P = C.POINTER(MeshParams)
        L.nbpm_deposit_device.argtypes = [P, C.c_size_t] + [_DP] * 5 + [C.c_void_p]
        L.nbpm_solve_device.argtypes = [P] + [_DP] * 4 + [C.c_void_p]
        L.nbpm_potential_device.argtypes = [P, _DP, C.c_void_p]
        L.nbpm_interpolate_device.argtypes = [P, C.c_size_t] + [_DP] * 9 + [C.c_void_p]
        L.nbpm_accelerations_device.argtypes = [P, C.c_size_t] + [_DP] * 7 + [C.c_void_p]
        L.nbpm_accelerations.argtypes = [P, C.c_size_t] + [_DP] * 7
        L.nbpm_sigma8_amplitude.argtypes = [C.c_double] * 4 + [C.POINTER(C.c_double)]
        IP = C.POINTER(IcParams)
        L.nbpm_zeldovich_field_device.argtypes = [IP] + [_DP] * 4 + [C.c_void_p]
        L.nbpm_zeldovich_particles_device.argtypes = [IP] + [_DP] * 3 + [C.c_double] * 2 + [_DP] * 7 + [C.c_void_p]
        self._inited = False

    def _check(self, rc):
        if rc != 0:
            raise NbppError(rc, self.lib.nbpm_last_error().decode(errors="replace"))

    def init(self, device=0):
        self._check(self.lib.nbpm_init(device))
        self._inited = True
        return self

    def shutdown(self):
        self.lib.nbpm_shutdown()
        self._inited = False

    def set_timing(self, on=True):
        self._check(self.lib.nbpm_set_timing(1 if on else 0))

    def stats(self):
        s = MeshStats()
        self._check(self.lib.nbpm_get_stats(C.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_}

    def deposit_device(self, p, x, y, z, m, rho, stream=None):
        if rho.numel() != p.mesh_size ** 3:
            raise ValueError("rho must hold mesh_size^3 values")
        self._check(self.lib.nbpm_deposit_device(C.byref(p), x.numel(), _dev(x), _dev(y), _dev(z),
                                                 _dev(m), _dev(rho), C.c_void_p(stream or 0)))

    def solve_device(self, p, rho, gx, gy, gz, stream=None):
        if min(t.numel() for t in (rho, gx, gy, gz)) != p.mesh_size ** 3:
            raise ValueError("meshes must hold mesh_size^3 values")
        self._check(self.lib.nbpm_solve_device(C.byref(p), _dev(rho), _dev(gx), _dev(gy), _dev(gz),
                                               C.c_void_p(stream or 0)))

    def potential_device(self, p, phi, stream=None):
        """Copy of the potential mesh of the last solve into the CUDA tensor `phi`."""
        if phi.numel() != p.mesh_size ** 3:
            raise ValueError("phi must hold mesh_size^3 values")
        self._check(self.lib.nbpm_potential_device(C.byref(p), _dev(phi), C.c_void_p(stream or 0)))

    def interpolate_device(self, p, x, y, z, gx, gy, gz, ax, ay, az, stream=None):
        if min(t.numel() for t in (gx, gy, gz)) != p.mesh_size ** 3:
            raise ValueError("meshes must hold mesh_size^3 values")
        if min(ax.numel(), ay.numel(), az.numel()) < x.numel():
            raise ValueError("acceleration arrays shorter than the particle count")
        self._check(self.lib.nbpm_interpolate_device(
            C.byref(p), x.numel(), _dev(x), _dev(y), _dev(z), _dev(gx), _dev(gy), _dev(gz),
            _dev(ax), _dev(ay), _dev(az), C.c_void_p(stream or 0)))

    def accelerations_device(self, p, x, y, z, m, ax, ay, az, stream=None):
        if min(ax.numel(), ay.numel(), az.numel()) < x.numel():
            raise ValueError("acceleration arrays shorter than the particle count")
        self._check(self.lib.nbpm_accelerations_device(
            C.byref(p), x.numel(), _dev(x), _dev(y), _dev(z), _dev(m), _dev(ax), _dev(ay),
            _dev(az), C.c_void_p(stream or 0)))

    # initial conditions ------------------------------------------------------------
    def sigma8_amplitude(self, sigma_8, omega_m, hubble_param, spectral_index):
        """A of P(k) = A k^n T(k)^2 (nbpm_sigma8_amplitude; host arithmetic, no GPU needed)."""
        a = C.c_double()
        self._check(self.lib.nbpm_sigma8_amplitude(sigma_8, omega_m, hubble_param, spectral_index, C.byref(a)))
        return a.value

    def zeldovich_field_device(self, ip, white_noise, dx, dy, dz, stream=None):
        """white noise (mesh^3 device tensor) -> the three displacement meshes (nbpm_zeldovich_field_device)."""
        self._check(self.lib.nbpm_zeldovich_field_device(C.byref(ip), _dev(white_noise), _dev(dx), _dev(dy), _dev(dz),
                                                         C.c_void_p(stream or 0)))

    def zeldovich_particles_device(self, ip, disp, velocity_factor, mass, pos, vel, m=None, stream=None):
        """The lattice particles displaced by the field (nbpm_zeldovich_particles_device)."""
        self._check(self.lib.nbpm_zeldovich_particles_device(
            C.byref(ip), *[_dev(d) for d in disp], velocity_factor, mass, *[_dev(a) for a in pos],
            *[_dev(a) for a in vel], _dev(m) if m is not None else None, C.c_void_p(stream or 0)))

    def accelerations(self, p, x, y, z, m, ax, ay, az):
        """Host numpy arrays; ax/ay/az are overwritten (nbpm_accelerations)."""
        self._check(self.lib.nbpm_accelerations(C.byref(p), x.size, _host(x), _host(y), _host(z),
                                                _host(m), _host(ax), _host(ay), _host(az)))


_MESH_LIB = None


def load_mesh(device=None):
    """Process-wide MeshLibrary; `device` given -> also nbpm_init(device)."""
    global _MESH_LIB
    if _MESH_LIB is None:
        _MESH_LIB = MeshLibrary()
    if device is not None and not _MESH_LIB._inited:
        _MESH_LIB.init(device)
    return _MESH_LIB


# ---------------------------------------------------------------------------------------
# libnbgas.so: the gas (HLL hydro) side, include/nbgas.h
# ---------------------------------------------------------------------------------------
GAS_LIB_PATH = os.environ.get("NBGAS_LIB") or os.path.join(HERE, "lib", "libnbgas.so")

GAS_EXPORTS = (
    "nbgas_init", "nbgas_shutdown", "nbgas_last_error", "nbgas_update_primitives_device",
    "nbgas_hydro_step_device", "nbgas_cfl_timestep_device", "nbgas_cfl_timestep_enqueue_device",
    "nbgas_cfl_timestep_finish", "nbgas_kick_device", "nbgas_kick_cfl_enqueue_device",
    "nbgas_total_density_device", "nbgas_initialize_device",
)

GAS_FIELDS = ("density", "momentum_x", "momentum_y", "momentum_z", "energy",
              "pressure", "velocity_x", "velocity_y", "velocity_z")


class GasParams(C.Structure):
    """nbgas_params."""
    _fields_ = [("cell_size", C.c_double), ("gamma", C.c_double), ("cfl_safety_factor", C.c_double),
                ("mesh_size", C.c_int32), ("reserved", C.c_int32)]


class GasState(C.Structure):
    """nbgas_state: nine device pointers."""
    _fields_ = [(k, _DP) for k in GAS_FIELDS]


class GasLibrary:
    """Loaded libnbgas.so.  Construction fails loudly when the shared object is missing."""

    def __init__(self, path=GAS_LIB_PATH):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found: build it with `make -C nbody_amd/csrc` "
                "(or __graft_entry__.build()); there is no fallback path")
        try:  # same reason as in Library: one HIP runtime per process
            import torch  # noqa: F401
        except ImportError:
            pass
        L = self.lib = C.CDLL(path)
        self.path = path
        P, S = C.POINTER(GasParams), C.POINTER(GasState)
        L.nbgas_init.argtypes = [C.c_int]
        L.nbgas_shutdown.restype = None
        L.nbgas_last_error.restype = C.c_char_p
        L.nbgas_update_primitives_device.argtypes = [P, S, C.c_void_p]
        L.nbgas_hydro_step_device.argtypes = [P, S, C.c_double, C.c_void_p]
        L.nbgas_cfl_timestep_device.argtypes = [P, S, C.POINTER(C.c_double), C.c_void_p]
        L.nbgas_cfl_timestep_enqueue_device.argtypes = [P, S, C.c_void_p]
        L.nbgas_cfl_timestep_finish.argtypes = [P, C.POINTER(C.c_double)]
        L.nbgas_kick_device.argtypes = [P, S] + [_DP] * 3 + [C.c_double] * 3 + [C.c_void_p]
        L.nbgas_kick_cfl_enqueue_device.argtypes = [P, S] + [_DP] * 3 + [C.c_double] * 3 + [C.c_void_p]
        L.nbgas_total_density_device.argtypes = [P, S, _DP, _DP, C.c_void_p]
        L.nbgas_initialize_device.argtypes = [P, S] + [_DP] * 4 + [C.c_double] * 3 + [C.c_int, C.c_double,
                                                                                     C.c_void_p]
        self._inited = False

    def _check(self, rc):
        if rc != 0:
            raise NbppError(rc, self.lib.nbgas_last_error().decode(errors="replace"))

    def init(self, device=0):
        self._check(self.lib.nbgas_init(device))
        self._inited = True
        return self

    def shutdown(self):
        self.lib.nbgas_shutdown()
        self._inited = False

    @staticmethod
    def state(tensors):
        """nbgas_state over a dict of nine torch CUDA tensors keyed by GAS_FIELDS."""
        s = GasState()
        for k in GAS_FIELDS:
            setattr(s, k, _dev(tensors[k]))
        return s

    def update_primitives_device(self, p, tensors, stream=None):
        s = self.state(tensors)
        self._check(self.lib.nbgas_update_primitives_device(C.byref(p), C.byref(s), C.c_void_p(stream or 0)))

    def hydro_step_device(self, p, tensors, dt, stream=None):
        s = self.state(tensors)
        self._check(self.lib.nbgas_hydro_step_device(C.byref(p), C.byref(s), dt, C.c_void_p(stream or 0)))

    def cfl_timestep_device(self, p, tensors, stream=None):
        s = self.state(tensors)
        out = C.c_double(0.0)
        self._check(self.lib.nbgas_cfl_timestep_device(C.byref(p), C.byref(s), C.byref(out),
                                                       C.c_void_p(stream or 0)))
        return out.value

    def cfl_timestep_enqueue_device(self, p, tensors, stream=None):
        """First half of cfl_timestep_device: queues the reduction and its read-back."""
        s = self.state(tensors)
        self._check(self.lib.nbgas_cfl_timestep_enqueue_device(C.byref(p), C.byref(s), C.c_void_p(stream or 0)))

    def cfl_timestep_finish(self, p):
        out = C.c_double(0.0)
        self._check(self.lib.nbgas_cfl_timestep_finish(C.byref(p), C.byref(out)))
        return out.value

    def kick_device(self, p, tensors, grav, dt, scale_factor, hubble, stream=None):
        s = self.state(tensors)
        self._check(self.lib.nbgas_kick_device(C.byref(p), C.byref(s), _dev(grav[0]), _dev(grav[1]),
                                               _dev(grav[2]), dt, scale_factor, hubble,
                                               C.c_void_p(stream or 0)))

    def kick_cfl_enqueue_device(self, p, tensors, grav, dt, scale_factor, hubble, stream=None):
        """kick_device then cfl_timestep_enqueue_device in one pass (nbgas_kick_cfl_enqueue_device)."""
        s = self.state(tensors)
        self._check(self.lib.nbgas_kick_cfl_enqueue_device(C.byref(p), C.byref(s), _dev(grav[0]), _dev(grav[1]),
                                                           _dev(grav[2]), dt, scale_factor, hubble,
                                                           C.c_void_p(stream or 0)))

    def total_density_device(self, p, tensors, dm_density, total, stream=None):
        s = self.state(tensors)
        self._check(self.lib.nbgas_total_density_device(C.byref(p), C.byref(s), _dev(dm_density),
                                                        _dev(total), C.c_void_p(stream or 0)))

    def initialize_device(self, p, tensors, dm_density, disp, mass_ratio, hubble, growth_f,
                          internal_energy, standing=False, stream=None):
        s = self.state(tensors)
        self._check(self.lib.nbgas_initialize_device(
            C.byref(p), C.byref(s), _dev(dm_density), _dev(disp[0]), _dev(disp[1]), _dev(disp[2]),
            mass_ratio, hubble, growth_f, 1 if standing else 0, internal_energy,
            C.c_void_p(stream or 0)))


_GAS_LIB = None


def load_gas(device=None):
    """Process-wide GasLibrary; `device` given -> also nbgas_init(device)."""
    global _GAS_LIB
    if _GAS_LIB is None:
        _GAS_LIB = GasLibrary()
    if device is not None and not _GAS_LIB._inited:
        _GAS_LIB.init(device)
    return _GAS_LIB


# ---------------------------------------------------------------------------------------
# libnbmg.so: the multi-GPU layer (one process per GPU, RCCL), include/nbmg.h
# ---------------------------------------------------------------------------------------
MG_LIB_PATH = os.environ.get("NBMG_LIB") or os.path.join(HERE, "lib", "libnbmg.so")

MG_EXPORTS = (
    "nbmg_get_unique_id", "nbmg_init", "nbmg_init_from_env", "nbmg_shutdown", "nbmg_last_error",
    "nbmg_rank", "nbmg_nranks", "nbmg_set_option", "nbmg_get_stats", "nbmg_partition",
    "nbmg_padded_length", "nbmg_allgather_device", "nbmg_allreduce_sum_device",
    "nbmg_allreduce_max_device", "nbmg_accumulate_device", "nbmg_max_accel_sq_device",
    "nbmg_x_extent_device", "nbmg_halo_select_device", "nbmg_transport",
)
MG_UNIQUE_ID_BYTES = 128
MG_OPT_TIMING, MG_OPT_PAIR_SHARES, MG_OPT_HALO_EXCHANGE = 1, 2, 3


class MultiGpuStats(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("allgather_ms", "compute_ms", "reduce_ms", "total_ms")] + [
        ("allgather_bytes", C.c_uint64), ("reduce_bytes", C.c_uint64),
        ("pair_shares", C.c_int32), ("halo_exchange", C.c_int32), ("halo_particles", C.c_uint64)]


class MultiGpuLibrary:
    """Loaded libnbmg.so.  Construction fails loudly when the shared object is missing."""

    def __init__(self, path=MG_LIB_PATH):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found: build it with `make -C nbody_amd/csrc` "
                "(or __graft_entry__.build()); there is no fallback path")
        try:  # one HIP runtime and one RCCL per process: torch's, when torch is in the process
            import torch  # noqa: F401
        except ImportError:
            pass
        Library()   # libnbpp.so first, from the same directory
        L = self.lib = C.CDLL(path)
        self.path = path
        SZ = C.POINTER(C.c_size_t)
        L.nbmg_get_unique_id.argtypes = [C.c_void_p]
        L.nbmg_init.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.nbmg_shutdown.restype = None
        L.nbmg_last_error.restype = C.c_char_p
        L.nbmg_transport.restype = C.c_char_p
        L.nbmg_set_option.argtypes = [C.c_int, C.c_int64]
        L.nbmg_get_stats.argtypes = [C.POINTER(MultiGpuStats)]
        L.nbmg_partition.argtypes = [C.c_size_t, C.c_int, C.c_int, SZ, SZ]
        L.nbmg_padded_length.argtypes = [C.c_size_t, C.c_int]
        L.nbmg_padded_length.restype = C.c_size_t
        L.nbmg_allgather_device.argtypes = [C.c_size_t, _DP, _DP, C.c_void_p]
        L.nbmg_allreduce_sum_device.argtypes = [C.c_size_t, _DP, C.c_void_p]
        L.nbmg_allreduce_max_device.argtypes = [C.c_size_t, _DP, C.c_void_p]
        L.nbmg_accumulate_device.argtypes = [C.POINTER(Params), C.c_size_t] + [_DP] * 7 + [C.c_void_p]
        L.nbmg_max_accel_sq_device.argtypes = [C.c_size_t] + [_DP] * 3 + [_DP, C.c_void_p]
        L.nbmg_x_extent_device.argtypes = [C.c_size_t, _DP, C.c_double, _DP, _DP, C.c_void_p]
        L.nbmg_halo_select_device.argtypes = ([C.c_size_t] + [_DP] * 4 + [C.c_double] * 4
                                              + [_DP, C.c_size_t, SZ, C.c_void_p])
        self._inited = False

    def _check(self, rc):
        if rc != 0:
            raise NbppError(rc, self.lib.nbmg_last_error().decode(errors="replace"))

    # lifecycle -------------------------------------------------------------------
    def unique_id(self) -> bytes:
        buf = C.create_string_buffer(MG_UNIQUE_ID_BYTES)
        self._check(self.lib.nbmg_get_unique_id(buf))
        return buf.raw

    def init(self, rank=0, nranks=1, unique_id=None, device=0):
        buf = C.create_string_buffer(unique_id, MG_UNIQUE_ID_BYTES) if unique_id is not None else None
        self._check(self.lib.nbmg_init(rank, nranks, buf, device))
        self._inited = True
        return self

    def shutdown(self):
        self.lib.nbmg_shutdown()
        self._inited = False

    def set_option(self, key, value):
        self._check(self.lib.nbmg_set_option(key, int(value)))

    def stats(self):
        s = MultiGpuStats()
        self._check(self.lib.nbmg_get_stats(C.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_}

    def transport(self) -> str:
        """"rccl", "host" (collectives staged through shared host memory: ranks sharing one GPU), or ""."""
        return self.lib.nbmg_transport().decode()

    # partition (no GPU needed) ------------------------------------------------------
    def partition(self, n, rank, nranks):
        b, c = C.c_size_t(), C.c_size_t()
        self._check(self.lib.nbmg_partition(n, rank, nranks, C.byref(b), C.byref(c)))
        return b.value, c.value

    def padded_length(self, n, nranks):
        return self.lib.nbmg_padded_length(n, nranks)

    # collectives and the force ---------------------------------------------------------
    def allgather_device(self, n, own, full, stream=None):
        self._check(self.lib.nbmg_allgather_device(n, _dev(own), _dev(full), C.c_void_p(stream or 0)))

    def allreduce_sum_device(self, data, stream=None):
        self._check(self.lib.nbmg_allreduce_sum_device(data.numel(), _dev(data), C.c_void_p(stream or 0)))

    def allreduce_max_device(self, data, stream=None):
        self._check(self.lib.nbmg_allreduce_max_device(data.numel(), _dev(data), C.c_void_p(stream or 0)))

    def accumulate_device(self, p, n, own_x, own_y, own_z, m_full, ax, ay, az, stream=None):
        """a[own] += PP force of all n particles on this rank's block (nbmg_accumulate_device)."""
        self._check(self.lib.nbmg_accumulate_device(
            C.byref(p), n, _dev(own_x), _dev(own_y), _dev(own_z), _dev(m_full), _dev(ax), _dev(ay),
            _dev(az), C.c_void_p(stream or 0)))

    def x_extent_device(self, x, domain_size, stream=None):
        """Circular x interval (lo, hi) that holds every position of the device tensor x."""
        lo, hi = C.c_double(), C.c_double()
        self._check(self.lib.nbmg_x_extent_device(x.numel(), _dev(x), domain_size, C.byref(lo), C.byref(hi),
                                                  C.c_void_p(stream or 0)))
        return lo.value, hi.value

    def halo_select_device(self, x, y, z, m, domain_size, lo, hi, margin, out=None, stream=None):
        """Number of particles within `margin` of the circular interval [lo, hi]; with `out` (a device
        tensor of 4 * capacity doubles) they are packed into it, x | y | z | m, in index order."""
        n_sel = C.c_size_t()
        cap = 0 if out is None else out.numel() // 4
        self._check(self.lib.nbmg_halo_select_device(
            x.numel(), _dev(x), _dev(y), _dev(z), _dev(m), domain_size, lo, hi, margin,
            _dev(out) if out is not None else None, cap, C.byref(n_sel), C.c_void_p(stream or 0)))
        return n_sel.value

    def max_accel_sq_device(self, a, count=None, stream=None):
        out = C.c_double(0.0)
        n = a[0].numel() if count is None else count
        self._check(self.lib.nbmg_max_accel_sq_device(n, _dev(a[0]), _dev(a[1]), _dev(a[2]), C.byref(out),
                                                      C.c_void_p(stream or 0)))
        return out.value


_MG_LIB = None


def load_multi_gpu():
    """Process-wide MultiGpuLibrary (not initialised: call .init(rank, nranks, id, device))."""
    global _MG_LIB
    if _MG_LIB is None:
        _MG_LIB = MultiGpuLibrary()
    return _MG_LIB

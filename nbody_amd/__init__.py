"""nbody_amd: MI355X-native particle-particle gravity (direct PP and P3M short range).

The product is lib/libnbpp.so (hand-written gfx950 HIP kernels behind the C ABI of
include/nbpp.h) and, for the mesh side of the same gravity step, lib/libnbpm.so
(include/nbpm.h), lib/libnbgas.so for the gas (include/nbgas.h) and lib/libnbmg.so, the
multi-GPU layer over libnbpp (include/nbmg.h: one process per GPU, RCCL).  This package is the thin Python host side used by the tests and the
benchmark: a ctypes binding (`nbody_amd.ffi`) and a mirror of the reference's
Config / ParticleSystem seam (`nbody_amd.particles`).  There is no CPU fallback: every
compute call raises if the library or a gfx950 device is missing.
"""
from .ffi import (GasLibrary, GasParams, Library, MeshLibrary, MeshParams, MultiGpuLibrary, NbppError,  # noqa: F401
                  Params, Stats, load, load_gas, load_mesh, load_multi_gpu)
from .particles import Config, ParticleSystem  # noqa: F401

__all__ = ["GasLibrary", "GasParams", "Library", "MeshLibrary", "MeshParams", "MultiGpuLibrary", "NbppError",
           "Params", "Stats", "load", "load_gas", "load_mesh", "load_multi_gpu",
           "Config", "ParticleSystem"]

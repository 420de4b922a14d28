"""nbody_amd: MI355X-native particle-particle gravity (direct PP and P3M short range).

The product is lib/libnbpp.so (hand-written gfx950 HIP kernels behind the C ABI of
include/nbpp.h) and, for the mesh side of the same gravity step, lib/libnbpm.so
(include/nbpm.h).  This package is the thin Python host side used by the tests and the
benchmark: a ctypes binding (`nbody_amd.ffi`) and a mirror of the reference's
Config / ParticleSystem seam (`nbody_amd.particles`).  There is no CPU fallback: every
compute call raises if the library or a gfx950 device is missing.
"""
from .ffi import Library, MeshLibrary, MeshParams, NbppError, Params, Stats, load, load_mesh  # noqa: F401
from .particles import Config, ParticleSystem  # noqa: F401

__all__ = ["Library", "MeshLibrary", "MeshParams", "NbppError", "Params", "Stats", "load", "load_mesh",
           "Config", "ParticleSystem"]

"""Reader of the engine's HDF5 snapshots (the reference's layout, src/hdf5_writter.cpp:57-134:
root attributes, /particles/*, /gas/*), through the HDF5 C library bound with ctypes - the image
has no h5py.  `read_snapshot` returns what scripts/validate_run.py:62-100 and scripts/viewer.py
take from a file.

Library search order: $NBODY_HDF5_LIB, libhdf5.so on the loader path, /opt/conda/lib/libhdf5.so
(the same order as the C++ writer, nbody_amd/host/snapshot.cpp).
"""
import ctypes as C
import os

import numpy as np

_hid = C.c_int64
_lib = None

PARTICLE_FIELDS = ("position_x", "position_y", "position_z", "velocity_x", "velocity_y", "velocity_z",
                   "acceleration_x", "acceleration_y", "acceleration_z", "mass")
GAS_FIELDS = ("density", "momentum_x", "momentum_y", "momentum_z", "energy", "pressure")


def _load():
    global _lib
    if _lib is not None:
        return _lib
    names = [os.environ.get("NBODY_HDF5_LIB"), "libhdf5.so", "libhdf5_serial.so", "/opt/conda/lib/libhdf5.so"]
    lib = None
    for name in names:
        if not name:
            continue
        try:
            lib = C.CDLL(name, mode=os.RTLD_LOCAL)
            break
        except OSError:
            continue
    if lib is None:
        raise OSError("no HDF5 library found (set NBODY_HDF5_LIB to a libhdf5.so, version 1.10 or later)")
    sig = {
        "H5open": (C.c_int, []),
        "H5get_libversion": (C.c_int, [C.POINTER(C.c_uint)] * 3),
        "H5Fopen": (_hid, [C.c_char_p, C.c_uint, _hid]),
        "H5Fclose": (C.c_int, [_hid]),
        "H5Lexists": (C.c_int, [_hid, C.c_char_p, _hid]),
        "H5Gopen2": (_hid, [_hid, C.c_char_p, _hid]),
        "H5Gclose": (C.c_int, [_hid]),
        "H5Dopen2": (_hid, [_hid, C.c_char_p, _hid]),
        "H5Dget_space": (_hid, [_hid]),
        "H5Dget_type": (_hid, [_hid]),
        "H5Dread": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Dclose": (C.c_int, [_hid]),
        "H5Sget_simple_extent_ndims": (C.c_int, [_hid]),
        "H5Sget_simple_extent_dims": (C.c_int, [_hid, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
        "H5Sclose": (C.c_int, [_hid]),
        "H5Aget_num_attrs": (C.c_int, [_hid]),
        "H5Aopen_by_idx": (_hid, [_hid, C.c_char_p, C.c_int, C.c_int, C.c_ulonglong, _hid, _hid]),
        "H5Aget_name": (C.c_ssize_t, [_hid, C.c_size_t, C.c_char_p]),
        "H5Aget_type": (_hid, [_hid]),
        "H5Aread": (C.c_int, [_hid, _hid, C.c_void_p]),
        "H5Aclose": (C.c_int, [_hid]),
        "H5Tget_class": (C.c_int, [_hid]),
        "H5Tget_size": (C.c_size_t, [_hid]),
        "H5Tclose": (C.c_int, [_hid]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    if lib.H5open() < 0:
        raise OSError("H5open failed")
    v = [C.c_uint() for _ in range(3)]
    lib.H5get_libversion(*[C.byref(x) for x in v])
    if (v[0].value, v[1].value) < (1, 10):
        raise OSError(f"HDF5 {v[0].value}.{v[1].value} is older than 1.10")
    lib.native_double = _hid.in_dll(lib, "H5T_NATIVE_DOUBLE_g").value
    lib.native_int = _hid.in_dll(lib, "H5T_NATIVE_INT_g").value
    _lib = lib
    return lib


def hdf5_available():
    try:
        _load()
        return True
    except OSError:
        return False


def _check(rc, what, path):
    if rc < 0:
        raise OSError(f"HDF5: {what} failed for {path}")
    return rc


def _attributes(lib, obj, path):
    out = {}
    n = _check(lib.H5Aget_num_attrs(obj), "H5Aget_num_attrs", path)
    for k in range(n):
        a = _check(lib.H5Aopen_by_idx(obj, b".", 0, 0, k, 0, 0), "H5Aopen_by_idx", path)   # by name, increasing
        buf = C.create_string_buffer(256)
        _check(lib.H5Aget_name(a, 256, buf), "H5Aget_name", path)
        t = _check(lib.H5Aget_type(a), "H5Aget_type", path)
        cls = lib.H5Tget_class(t)                      # 0 integer, 1 floating point
        lib.H5Tclose(t)
        if cls == 0:
            v = C.c_int()
            _check(lib.H5Aread(a, lib.native_int, C.byref(v)), "H5Aread", path)
            out[buf.value.decode()] = int(v.value)
        elif cls == 1:
            v = C.c_double()
            _check(lib.H5Aread(a, lib.native_double, C.byref(v)), "H5Aread", path)
            out[buf.value.decode()] = float(v.value)
        lib.H5Aclose(a)
    return out


def _dataset(lib, f, name, path):
    d = _check(lib.H5Dopen2(f, name.encode(), 0), f"H5Dopen2({name})", path)
    space = _check(lib.H5Dget_space(d), "H5Dget_space", path)
    rank = _check(lib.H5Sget_simple_extent_ndims(space), "H5Sget_simple_extent_ndims", path)
    dims = (C.c_ulonglong * max(rank, 1))()
    _check(lib.H5Sget_simple_extent_dims(space, dims, None), "H5Sget_simple_extent_dims", path)
    lib.H5Sclose(space)
    t = lib.H5Dget_type(d)
    if lib.H5Tget_class(t) != 1 or lib.H5Tget_size(t) != 8:
        raise OSError(f"{path}:{name} is not a float64 dataset")
    lib.H5Tclose(t)
    out = np.empty(tuple(int(x) for x in dims[:rank]), dtype=np.float64)
    _check(lib.H5Dread(d, lib.native_double, 0, 0, 0, out.ctypes.data_as(C.c_void_p)), f"H5Dread({name})", path)
    lib.H5Dclose(d)
    return out


def read_snapshot(path, particle_fields=PARTICLE_FIELDS, gas_fields=GAS_FIELDS):
    """{'attrs': {name: int | float}, 'particles': {field: float64[n]}, 'gas': {field:
    float64[M, M, M]} or None when the file has no /gas group}."""
    lib = _load()
    path = os.fspath(path)
    f = _check(lib.H5Fopen(path.encode(), 0, 0), "H5Fopen", path)      # H5F_ACC_RDONLY
    try:
        root = _check(lib.H5Gopen2(f, b"/", 0), "H5Gopen2", path)
        attrs = _attributes(lib, root, path)
        lib.H5Gclose(root)
        snap = {"attrs": attrs,
                "particles": {k: _dataset(lib, f, "particles/" + k, path) for k in particle_fields},
                "gas": None}
        if _check(lib.H5Lexists(f, b"gas", 0), "H5Lexists", path) > 0:
            snap["gas"] = {k: _dataset(lib, f, "gas/" + k, path) for k in gas_fields}
        return snap
    finally:
        lib.H5Fclose(f)

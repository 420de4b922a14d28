"""The C-ABI library loads on a CPU-only machine and exports what include/nbpp.h declares."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

import nbody_amd
from nbody_amd import ffi
from conftest import ROOT


def _declared(header="nbpp.h", prefix="nbpp_"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_header_and_library_agree():
    names = _declared()
    assert len(names) >= 10
    lib = ctypes.CDLL(ffi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/nbpp.h but not exported"
    assert sorted(ffi.EXPORTS) == names


def test_mesh_header_and_library_agree():
    names = _declared("nbpm.h", "nbpm_")
    assert len(names) >= 10
    lib = ctypes.CDLL(ffi.MESH_LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/nbpm.h but not exported"
    assert sorted(ffi.MESH_EXPORTS) == names
    assert ctypes.sizeof(ffi.MeshParams) == 2 * 8 + 2 * 4
    assert ctypes.sizeof(ffi.MeshStats) == 4 * 8


def test_mesh_library_has_no_silent_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = ffi.MeshLibrary()
    with pytest.raises(ffi.NbppError):
        L.init(0)
    a = np.zeros(4)
    with pytest.raises(ffi.NbppError) as e:
        L.accelerations(nbody_amd.Config().mesh_params(), a, a, a, a, a.copy(), a.copy(), a.copy())
    assert e.value.code == -3  # NBPM_ERR_NOINIT


def test_gas_header_and_library_agree():
    names = _declared("nbgas.h", "nbgas_")
    assert len(names) >= 9
    lib = ctypes.CDLL(ffi.GAS_LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/nbgas.h but not exported"
    assert sorted(ffi.GAS_EXPORTS) == names
    assert ctypes.sizeof(ffi.GasParams) == 3 * 8 + 2 * 4
    assert ctypes.sizeof(ffi.GasState) == 9 * 8
    L = ffi.GasLibrary()
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(ffi.NbppError):
            L.init(0)                     # no device: fails loudly, no CPU path


def test_multi_gpu_header_and_library_agree():
    names = _declared("nbmg.h", "nbmg_")
    assert len(names) >= 14
    lib = ctypes.CDLL(ffi.LIB_PATH, mode=ctypes.RTLD_GLOBAL)   # libnbmg.so needs libnbpp.so
    lib = ctypes.CDLL(ffi.MG_LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/nbmg.h but not exported"
    assert sorted(ffi.MG_EXPORTS) == names
    assert ctypes.sizeof(ffi.MultiGpuStats) == 4 * 8 + 2 * 8 + 2 * 4 + 8
    L = ffi.MultiGpuLibrary()
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(ffi.NbppError):
            L.init(0, 1, None, 0)          # no device: fails loudly, no CPU path
        assert L.lib.nbmg_rank() == -1 and L.lib.nbmg_nranks() == 0


def test_multi_gpu_partition_is_blocks_of_the_padded_layout():
    """nbmg_partition: rank r owns [r * stride, min(n, (r + 1) * stride)), stride = ceil(n / nranks),
    so a padded array in global order is rank-major blocks and collectives need no repacking."""
    L = ffi.MultiGpuLibrary()
    for n, w in ((10, 3), (8, 8), (2097152, 8), (16777216, 8), (5, 8), (0, 2), (1001, 2)):
        stride = -(-n // w)
        assert L.padded_length(n, w) == stride * w
        covered = 0
        for r in range(w):
            b, c = L.partition(n, r, w)
            assert b == min(n, r * stride) and c == min(stride, n - b)
            assert b == covered or c == 0
            covered += c
        assert covered == n
    with pytest.raises(ffi.NbppError):
        L.partition(10, 3, 3)


def test_struct_layouts_match_header():
    assert ctypes.sizeof(ffi.Params) == 9 * 8 + 2 * 4
    assert ctypes.sizeof(ffi.Stats) == 5 * 8 + 7 * 8 + 2 * 4 + 8
    from oracle.oracle import Params as OrcParams
    assert [f[0] for f in OrcParams._fields_] == [f[0] for f in ffi.Params._fields_]


def test_no_silent_fallback_without_gpu():
    """Without a GPU every compute entry point must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = ffi.Library()
    with pytest.raises(ffi.NbppError):
        L.init(0)
    p = nbody_amd.Config().pp_params()
    a = np.zeros(4)
    with pytest.raises(ffi.NbppError) as e:
        L.accumulate(p, a, a, a, a, a.copy(), a.copy(), a.copy())
    assert e.value.code == -3  # NBPP_ERR_NOINIT


def test_product_does_not_import_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "nbody_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                text = open(os.path.join(root, f), errors="replace").read()
                assert "oracle" not in text.lower().replace("test_oracle", ""), f


def test_bench_parent_of_a_multi_gpu_run_never_loads_the_gpu_runtime(tmp_path):
    """`python bench.py --gpus 2` without a launcher: the parent only starts the ranks
    (torch.distributed.run) and relays their line.  It must not load torch or the HIP runtime - a
    process that has initialised the GPU has no business holding it beside its ranks.  The dynamic
    loader itself records what each process maps (LD_DEBUG=libs, one file per pid).  Here, without a
    GPU, the ranks fail, and the parent must pass that on as a non-zero exit code."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update({"LD_DEBUG": "libs", "LD_DEBUG_OUTPUT": str(tmp_path / "ld")})
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n-side", "8",
                             "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, text=True, env=env)
    out, err = proc.communicate(timeout=600)
    import torch
    if not torch.cuda.is_available():
        assert proc.returncode != 0 and '{"metric"' not in out
    parent = (tmp_path / f"ld.{proc.pid}").read_text()
    assert "libpython" in parent or "libc.so" in parent          # the record is there
    for name in ("libamdhip64", "libtorch", "libhsa-runtime", "librccl"):
        assert name not in parent, f"the bench parent loaded {name}"
    others = [f for f in tmp_path.iterdir() if f.name != f"ld.{proc.pid}"]
    assert any("libtorch" in f.read_text() for f in others)      # the ranks did start

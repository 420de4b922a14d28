import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def lib():
    """libnbpp bound to GPU 0.  Fails (does not skip) when the library or the GPU is missing."""
    import nbody_amd
    L = nbody_amd.load()
    L.init(0)
    L.set_option(nbody_amd.ffi.OPT_TIMING, 1)
    yield L
    L.shutdown()


def normalised_error(got, ref, floor=0.0):
    """max_i,c |a_got - a_ref|_c / max(||a_ref,i||_2, floor): the parity measure of the PP path.

    `floor` is for particle sets whose net force vanishes by symmetry (a perfect lattice):
    there the denominator is the force scale of one nearest-neighbour pair instead."""
    got = np.stack(got, axis=1)
    ref = np.stack(ref, axis=1)
    norm = np.maximum(np.linalg.norm(ref, axis=1), floor)
    norm = np.where(norm > 0, norm, 1.0)
    return float(np.max(np.abs(got - ref) / norm[:, None]))


def pair_force_scale(p, m, n):
    """G m / spacing^2 for n particles in the box of params p: one nearest-neighbour pair."""
    spacing = p.domain_size / round(n ** (1.0 / 3.0))
    return p.G * float(np.max(m)) / spacing ** 2

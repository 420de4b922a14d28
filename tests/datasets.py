"""Seeded synthetic particle sets shared by the tests, the golden generator and bench.py."""
import numpy as np


def jittered_lattice(n_side, L=1.0, jitter=0.3, seed=12345):
    """(i + 0.5)/n + jitter/n * (U - 0.5) per axis, x slowest: the near-uniform load of early
    cosmological ICs (lattice plus a small displacement)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    g = (np.arange(n_side) + 0.5) / n_side
    x, y, z = np.meshgrid(g, g, g, indexing="ij")
    pos = np.stack([x.ravel(), y.ravel(), z.ravel()])
    pos = pos + (jitter / n_side) * (rng.random(pos.shape) - 0.5)
    pos = np.mod(pos, 1.0) * L
    return [np.ascontiguousarray(pos[c]) for c in range(3)]


def uniform_random(n, L=1.0, seed=7):
    rng = np.random.Generator(np.random.MT19937(seed))
    return [np.ascontiguousarray(rng.random(n) * L) for _ in range(3)]


def clustered(n, L=1.0, n_blobs=6, sigma=0.03, seed=11):
    """Gaussian blobs wrapped into the box plus a uniform background: late-time-like clustering,
    including blobs that straddle the periodic boundary."""
    rng = np.random.Generator(np.random.MT19937(seed))
    centres = rng.random((n_blobs, 3))
    centres[0] = [0.001, 0.5, 0.999]  # on the box faces
    which = rng.integers(0, n_blobs, size=n)
    pos = centres[which] + sigma * rng.standard_normal((n, 3))
    bg = rng.random(n) < 0.3
    pos[bg] = rng.random((int(bg.sum()), 3))
    pos = np.mod(pos, 1.0) * L
    return [np.ascontiguousarray(pos[:, c]) for c in range(3)]


def masses(n, equal=True, total=0.84, seed=3):
    if equal:
        return np.full(n, total / n)
    rng = np.random.Generator(np.random.MT19937(seed))
    m = rng.random(n) + 0.5
    return m * (total / m.sum())


def zeldovich(n_side, mesh_size=None, seed=42):
    """Zel'dovich initial conditions exactly as the reference generates them (restated in
    oracle/zeldovich.py; test infrastructure).  Returns positions [x, y, z]."""
    from oracle.zeldovich import zeldovich_particles
    pos, _, _ = zeldovich_particles(n_per_side=n_side, mesh_size=mesh_size or n_side, seed=seed)
    return pos

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cornell():
    from nova_amd import scene as S
    return S.cornell_box()


@pytest.fixture(scope="session")
def soup50k():
    from nova_amd import scene as S
    return S.build_scene(S.soup(50000, seed=4, edge=0.05))


@pytest.fixture(scope="session")
def knot_small():
    """Small version of the dragon stand-in (with its light and ground plane)."""
    from nova_amd import scene as S
    return S.dragon_standin(96, 96)


def rel_rmse(a, b):
    """Relative RMSE over the rgb planes.  NaNs must sit in the same places in both (the reference
    renormalises an all-zero accumulated normal for sky pixels: kernels/kernels.cu:80-82)."""
    a = np.asarray(a)[..., :3].astype(np.float64)
    b = np.asarray(b)[..., :3].astype(np.float64)
    assert (np.isnan(a) == np.isnan(b)).all(), "NaN pattern differs"
    ok = ~np.isnan(b)
    return float(np.sqrt(((a[ok] - b[ok]) ** 2).sum() / max((b[ok] ** 2).sum(), 1e-300)))


def random_rays(n, seed, lo=-1.2, hi=1.2, anyhit_every=0, tmax_every=0):
    """n rays: points uniform in [lo,hi]^3, unit directions; every k-th ray any-hit / bounded."""
    from nova_amd import scene as S
    rng = np.random.default_rng(seed)
    pt = np.zeros((n, 4), np.float32)
    pt[:, :3] = rng.uniform(lo, hi, (n, 3))
    pt[:, 3] = S.FLT_MAX
    d = rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    df = np.zeros((n, 4), np.float32)
    df[:, :3] = d
    flags = np.zeros(n, np.uint32)
    if anyhit_every:
        flags[::anyhit_every] = 1
    if tmax_every:
        pt[::tmax_every, 3] = rng.uniform(0.05, 2.5, pt[::tmax_every].shape[0])
    df[:, 3] = flags.view(np.float32)
    return pt, df

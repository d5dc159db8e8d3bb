import hashlib
import os

import numpy as np

from nova_amd import scene as S

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SOUP_CAMERA = dict(position=[0, 0, 3.5], target=[0, 0, 0], up=[0, 1, 0], fovy=40.0)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def scene_hash(sc):
    h = hashlib.sha256()
    for a in (sc.bvh, sc.triangles, sc.meta, sc.lights):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def check_scene(sc, g):
    """The fixtures were rendered from a scene the in-tree builder must reproduce bit for bit."""
    assert scene_hash(sc) == str(g["scene_sha256"]), "scene builder output drifted from the fixtures"


def id_inputs(sc):
    """meta / sky that make kernel_raytrace (path_tracing = 0) report primary-hit triangle ids."""
    return S.encode_ids_in_kd(sc.meta), S.constant_sky((-1, -1, -1), w=sc.sky.shape[1], h=sc.sky.shape[0])


def lit_soup(n_tris=4000, n_emissive=6, seed=11, edge=0.25):
    """Random triangles in [-1, 1]^3 of which a few large ones emit: lights in the middle of the clutter, so that
    BRDF samples reach them from everywhere and other triangles often sit in between."""
    verts = S.soup(n_tris, seed=seed, edge=edge)
    em = np.zeros((n_tris, 3), np.float32)
    rng = np.random.default_rng(seed + 1)
    em[rng.choice(n_tris, n_emissive, replace=False)] = (40.0, 30.0, 20.0)
    return S.build_scene(verts, emissive=em)

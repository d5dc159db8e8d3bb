"""How much does ray order matter?  Random rays traced in random order, sorted by origin cell
(Morton), and sorted by (direction octant, origin cell)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nova_amd import scene as S, capi
from tests.conftest import random_rays

capi.init(0)


def part1by2(x):
    x = x.astype(np.uint64) & 0x3FF
    x = (x | (x << 16)) & 0x30000FF
    x = (x | (x << 8)) & 0x300F00F
    x = (x | (x << 4)) & 0x30C30C3
    x = (x | (x << 2)) & 0x9249249
    return x


def morton(p, lo, hi, bits=10):
    q = np.clip(((p - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    return part1by2(q[:, 0]) | (part1by2(q[:, 1]) << 1) | (part1by2(q[:, 2]) << 2)


def run(sc, pt, df, label):
    n = pt.shape[0]
    tb, bb = capi.Buffer(sc.triangles), capi.Buffer(sc.bvh)
    hb = capi.Buffer(nbytes=16 * n)
    lo, hi = pt[:, :3].min(0), pt[:, :3].max(0)
    m = morton(pt[:, :3], lo, hi)
    octant = ((df[:, 0] < 0).astype(np.uint64) | ((df[:, 1] < 0).astype(np.uint64) << 1) | ((df[:, 2] < 0).astype(np.uint64) << 2))
    orders = {
        "as generated": np.arange(n),
        "origin morton 30 bit": np.argsort(m, kind="stable"),
        "origin morton 15 bit": np.argsort(m >> 15, kind="stable"),
        "octant, origin 30 bit": np.argsort((octant << 30) | m, kind="stable"),
        "origin 15 bit, octant": np.argsort(((m >> 15) << 3) | octant, kind="stable"),
    }
    for name, o in orders.items():
        ro, rd = capi.Buffer(np.ascontiguousarray(pt[o])), capi.Buffer(np.ascontiguousarray(df[o]))
        capi.trace_rays(tb, bb, ro, rd, hb, n)
        best = 1e9
        for _ in range(3):
            capi.reset_counters()
            capi.trace_rays(tb, bb, ro, rd, hb, n)
            best = min(best, capi.counters()["traverse_ms"])
        print("%s, %d rays, %-24s %.3f ms  %.0f Mrays/s" % (label, n, name + ":", best, n / best / 1e3), flush=True)


sc = S.dragon_standin()
pt, df = random_rays(4_000_000, seed=5, lo=-1.0, hi=1.0)
run(sc, pt, df, "dragon stand-in")
sc = S.build_scene(S.soup(2_000_000, seed=4))
run(sc, pt, df, "soup 2M")

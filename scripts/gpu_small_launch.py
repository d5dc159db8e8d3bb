"""Traversal launch time vs number of rays (dragon stand-in, random rays), for A/B of NOVA_HIP_OPT_TUNE values.
usage: gpu_small_launch.py tune[,tune...]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nova_amd import scene as S, capi
from tests.conftest import random_rays

tunes = [int(x, 0) for x in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
capi.init(0)
sc = S.dragon_standin()
tb, bb = capi.Buffer(sc.triangles), capi.Buffer(sc.bvh)
for n in (64, 5000, 20000, 70000, 220000, 1400000, 4800000):
    pt, df = random_rays(n, seed=7, anyhit_every=3)
    ro, rd, hb = capi.Buffer(pt), capi.Buffer(df), capi.Buffer(nbytes=16 * n)
    base = None
    line = "%8d rays:" % n
    for t in tunes:
        capi.set_option(capi.OPT_TUNE, t)
        capi.trace_rays(tb, bb, ro, rd, hb, n)
        best = 1e9
        for rep in range(5):
            capi.reset_counters()
            capi.trace_rays(tb, bb, ro, rd, hb, n)
            best = min(best, capi.counters()["traverse_ms"])
        got = hb.read(np.uint32).reshape(n, 4)
        closest = (df[:, 3].view(np.uint32) & 1) == 0
        if base is None:
            base = got
        ok = bool((got[closest] == base[closest]).all())
        line += "  tune %#x: %.4f ms%s" % (t, best, "" if ok else " MISMATCH")
    print(line, flush=True)
capi.set_option(capi.OPT_TUNE, 0)

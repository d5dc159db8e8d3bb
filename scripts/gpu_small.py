"""Traversal time vs number of rays (fixed cost / tail of a launch)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nova_amd import scene as S, capi
from tests.conftest import random_rays
capi.init(0)
capi.set_option(capi.OPT_TUNE, int(os.environ.get('NOVA_TUNE', '0')))
sc = S.dragon_standin()
tb, bb = capi.Buffer(sc.triangles), capi.Buffer(sc.bvh)
kernels = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,6,7").split(",")]
for n in (64, 1024, 16384, 65536, 262144, 1048576, 4194304):
    pt, df = random_rays(n, seed=5, lo=-1.0, hi=1.0)
    ro, rd, hb = capi.Buffer(pt), capi.Buffer(df), capi.Buffer(nbytes=16 * n)
    line = "n=%8d" % n
    for k in kernels:
        capi.set_option(capi.OPT_TRAVERSE_KERNEL, k)
        capi.trace_rays(tb, bb, ro, rd, hb, n)
        best = 1e9
        for rep in range(3):
            capi.reset_counters()
            capi.trace_rays(tb, bb, ro, rd, hb, n)
            best = min(best, capi.counters()["traverse_ms"])
        line += "  k%d %.3f ms" % (k, best)
    print(line, flush=True)

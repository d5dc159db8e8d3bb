"""nova_hip_render_samples with a long sample list (more kernels than the timing-event pool holds):
1000 samples of a small frame in one call against 1000 four-call samples, bitwise."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
from nova_amd import scene as S, capi  # noqa: E402
from nova_amd.raytracer import Raytracer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
capi.init(0)
sc = S.cornell_box()
w = h = 96
p = S.make_scene_params(S.Camera(width=w, height=h, **S.CORNELL_CAMERA), path_tracing=True)
ok = True
for depth in (5, 0):
    capi.set_option(capi.OPT_MAX_DEPTH, depth)
    a = Raytracer(0)
    a.set_scene(sc, p, w, h)
    a.start()
    t0 = time.perf_counter()
    for k in range(n):
        a.raytrace(1 + k, read_back=False)
    t1 = time.perf_counter()
    b = Raytracer(0)
    b.set_scene(sc, p, w, h)
    b.start()
    t2 = time.perf_counter()
    b.raytrace_samples([1 + k for k in range(n)], read_back=False)
    t3 = time.perf_counter()
    same = bool((a.temp_color_img2.read().view(np.uint32) == b.temp_color_img2.read().view(np.uint32)).all()
                and (a.color_img.read() == b.color_img.read()).all())
    ok &= same
    print(f"max_depth {depth}: {n} samples, four calls {(t1 - t0) / n * 1e3:.3f} ms, one call {(t3 - t2) / n * 1e3:.3f} ms per sample; "
          f"bitwise equal: {same}", flush=True)
capi.set_option(capi.OPT_MAX_DEPTH, 0)
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)

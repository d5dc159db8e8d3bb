import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from nova_amd import scene as S, capi
from nova_amd.raytracer import Raytracer
from oracle import bind as O
capi.init(0)
W = H = 256
sc = S.cornell_box()
cam = S.Camera(width=W, height=H, **S.CORNELL_CAMERA)
rt = Raytracer(0)
p = S.make_scene_params(cam, path_tracing=True)
rt.set_scene(sc, p, W, H)
for depth in (1, 2, 3):
    capi.set_option(capi.OPT_MAX_DEPTH, depth)
    rt.kernel_raytrace(3)
    c = rt.temp_color_img1.read()[..., :3]
    co = O.kernel_raytrace(sc, p, 3, W, H, max_depth=depth)[0][..., :3]
    d = np.abs(c - co).max(-1)
    bad = np.argwhere(d > 1e-4 * (1 + np.abs(co).max(-1)))
    print("depth", depth, "bad px", len(bad), "max diff", d.max())
    for (y, x) in bad[:6]:
        print("   ", y, x, c[y, x], co[y, x])

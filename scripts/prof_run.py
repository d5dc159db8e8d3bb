"""Small fixed workload for rocprofv3 passes: dragon stand-in, 1920x1080, max_depth 5, N frames."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nova_amd import scene as S, capi
from nova_amd.raytracer import Raytracer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
capi.init(0)
sc = S.dragon_standin()
W, H = 1920, 1080
cam = S.Camera(width=W, height=H, **S.DRAGON_CAMERA)
p = S.make_scene_params(cam, shading_metallic=0.4, shading_roughness=0.1)
rt = Raytracer(0)
rt.set_scene(sc, p, W, H)
capi.set_option(capi.OPT_MAX_DEPTH, 5)
# one sample at a time: rocprofv3's counter passes serialise dispatches anyway, and a launch that believes it shares the
# GPU with other samples' kernels would run with 3 instead of 5 work-groups per CU (NOVA_HIP_OPT_CALL_LANES)
capi.set_option(capi.OPT_CALL_LANES, 1)
capi.set_option(capi.OPT_TRAVERSE_KERNEL, int(os.environ.get('NOVA_KERNEL', '0')))
rt.start()
for i in range(n):
    rt.raytrace(1 + i, read_back=False)
capi.synchronize()
print("ok")

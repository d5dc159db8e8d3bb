"""One build of the library on the benchmark frame: ms per sample (best of 3 x 16), work counters and a hash of the
frame bytes.  A/B of two builds: run it once per build with NOVA_HIP_LIB=<path to the .so> and compare the lines."""
import os, sys, time, hashlib
sys.path.insert(0, "/root/repo")
from nova_amd import scene as S, capi
from nova_amd.raytracer import Raytracer

if os.environ.get("NOVA_AB_TORCH"):  # a process that also holds PyTorch's HIP runtime, streams and a tensor (bench.py --gpus N)
    import torch
    torch.cuda.set_device(0)
    _keep = [torch.zeros(1 << 20, device="cuda"), torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
capi.init(0)
sc = S.dragon_standin()
W, H = 1920, 1080
p = S.make_scene_params(S.Camera(width=W, height=H, **S.DRAGON_CAMERA), shading_metallic=0.4, shading_roughness=0.1)
rt = Raytracer(0)
rt.set_scene(sc, p, W, H)
capi.set_option(capi.OPT_MAX_DEPTH, 5)
if os.environ.get("NOVA_AB_MIS_PRETEST") is not None and hasattr(capi, "OPT_MIS_PRETEST"):
    capi.set_option(capi.OPT_MIS_PRETEST, int(os.environ["NOVA_AB_MIS_PRETEST"]))
if os.environ.get("NOVA_AB_ASYNC") is not None and hasattr(capi, "OPT_ASYNC_LAUNCH"):
    capi.set_option(capi.OPT_ASYNC_LAUNCH, int(os.environ["NOVA_AB_ASYNC"]))
if len(sys.argv) > 1:
    capi.set_option(capi.OPT_TRAVERSE_KERNEL, int(sys.argv[1]))
rt.start()
for i in range(3):
    rt.raytrace(1 + i, read_back=False)
capi.synchronize()
best = 1e9
for rep in range(3):
    rt.start()
    capi.reset_counters()
    t = time.perf_counter()
    for i in range(16):
        rt.raytrace(1 + i, read_back=False)
    capi.synchronize()
    best = min(best, (time.perf_counter() - t) / 16)
k = capi.counters()
img = rt.color_img.read()
capi.set_option(capi.OPT_COUNT_WORK, 1)
capi.reset_counters()
rt.start()
rt.raytrace(1, read_back=False)
capi.synchronize()
w = capi.counters()
print("%s: %.4f ms per sample, traverse %.4f ms (camera rays %.4f), shade %.4f ms, node lines %d, triangle tests %d, frame sha %s, MIS rays not traced %d of %d bounce+MIS rays, shadow rays not traced %d of %d" % (
    capi.LIB_PATH.split("/")[-1], best * 1e3, k["traverse_ms"] / 16, k["camera_traverse_ms"] / 16, k["shade_ms"] / 16, w["node_visits"], w["triangle_tests"],
    hashlib.sha256(img.tobytes()).hexdigest()[:16], w.get("rays_mis_pretested", 0), w["rays_bounce"], w.get("rays_shadow_skipped", 0), w["rays_shadow"]))

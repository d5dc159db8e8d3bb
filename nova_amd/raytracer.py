"""Python mirror of ``nova::Raytracer`` (reference ``src/core/raytracer.cpp``) over the C ABI.

Same objects, same per-sample schedule and the same names as the reference's only caller of the
backend, so that the parity tests read like its flow:

    set_scene  -> create_buffer x4, create_image2D_* (raytracer.cpp:47-167)
    start      -> sample_index = 0, fill prev_color (raytracer.cpp:42-45)
    raytrace   -> kernel_raytrace, kernel_accumulate, copy_image2D, kernel_post_process,
                  read_image2D (raytracer.cpp:169-262)

The production C++ equivalent of this file is Nova's own, untouched ``Raytracer`` compiled with
``-DBACKEND=HIP`` against ``nova_amd/backend/hip``.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import capi
from .scene import Scene


class Raytracer:
    def __init__(self, device: int = 0, denoise_available: bool = False):
        capi.init(device)
        # OIDN is absent (.MISSING_LARGE_BLOBS), so the reference would run with
        # denoise_available = false; tests switch the feature images on explicitly.
        self.denoise_available = 1 if denoise_available else 0
        self.width = self.height = 0
        self.sample_index = 0
        self.albedo_img1 = self.normal_img1 = self.albedo_img2 = self.normal_img2 = None
        self.prev_albedo_img = self.prev_normal_img = None
        self.fixed_seed: Optional[int] = None

    # raytracer.cpp:47-167
    def set_scene(self, scene: Scene, params: np.ndarray, width: int, height: int, materials_u8=None):
        self.params = np.ascontiguousarray(params)
        if (width, height) != (self.width, self.height):
            f4 = dict(texel_bytes=16, address_mode=capi.ADDRESS_CLAMP, filter_mode=capi.FILTER_LINEAR)
            self.color_img = capi.Image(width, height, texel_bytes=4)
            self.prev_color_img = capi.Image(width, height, **f4)
            self.temp_color_img1 = capi.Image(width, height, **f4)
            self.temp_color_img2 = capi.Image(width, height, **f4)
            if self.denoise_available:
                self.albedo_img1 = capi.Image(width, height, **f4)
                self.normal_img1 = capi.Image(width, height, **f4)
                self.albedo_img2 = capi.Image(width, height, **f4)
                self.normal_img2 = capi.Image(width, height, **f4)
                self.prev_albedo_img = capi.Image(width, height, **f4)
                self.prev_normal_img = capi.Image(width, height, **f4)
        self.triangle_buf = capi.Buffer(scene.triangles)
        self.tri_meta_buf = capi.Buffer(scene.meta)
        self.bvh_buf = capi.Buffer(scene.bvh)
        self.light_buf = capi.Buffer(scene.lights) if scene.num_lights else None
        self.num_lights = scene.num_lights
        # float4 layers as MaterialLoader::build returns them, or its uint8 layers (8-bit storage: the backend
        # converts x / 255.0f at sampling time, include/nova_hip.h)
        m = scene.materials if materials_u8 is None else np.ascontiguousarray(materials_u8, np.uint8)
        self.material_imgs = capi.Image(m.shape[2], m.shape[1], layers=m.shape[0],
                                        texel_bytes=16 if materials_u8 is None else 4,
                                        address_mode=capi.ADDRESS_WRAP, data=m)
        s = scene.sky
        self.sky_img = capi.Image(s.shape[1], s.shape[0], texel_bytes=16,
                                  address_mode=capi.ADDRESS_WRAP, data=s)
        self.width, self.height = width, height

    def set_meta(self, meta: np.ndarray):
        self.tri_meta_buf.write(meta)

    # raytracer.cpp:42-45
    def start(self):
        self.sample_index = 0
        self.prev_color_img.fill([0.0, 0.0, 0.0, 0.0])

    def kernel_raytrace(self, time: int):
        capi.launch_raytrace(self.params, time, self.temp_color_img1, self.width, self.height,
                             self.triangle_buf, self.tri_meta_buf, self.bvh_buf, self.light_buf,
                             self.num_lights, self.material_imgs, self.sky_img,
                             self.denoise_available, self.albedo_img1, self.normal_img1)

    def kernel_accumulate(self):
        capi.launch_accumulate(self.sample_index, self.denoise_available, self.temp_color_img1,
                               self.albedo_img1, self.normal_img1, self.prev_color_img,
                               self.prev_albedo_img, self.prev_normal_img, self.temp_color_img2,
                               self.albedo_img2, self.normal_img2, self.width, self.height)

    def kernel_post_process(self):
        capi.launch_post_process(self.params, self.temp_color_img2, self.color_img, self.width,
                                 self.height)

    # raytracer.cpp:169-262 (without the OIDN hand-off, whose library is absent)
    def raytrace(self, time: int, read_back: bool = True, post_process: bool = True):
        """post_process=False: a rank of a tile partition with FXAA on leaves kernel_post_process to
        tiles.finish_frame (it must see the whole frame)."""
        self.kernel_raytrace(time)
        self.kernel_accumulate()
        self.prev_color_img.copy_from(self.temp_color_img2)
        if self.denoise_available:
            self.prev_albedo_img.copy_from(self.albedo_img2)
            self.prev_normal_img.copy_from(self.normal_img2)
        if post_process:
            self.kernel_post_process()
        self.sample_index += 1
        return self.color_img.read() if read_back else None

    def frame_images(self) -> capi.FrameImages:
        h = lambda im: None if im is None else im.handle  # noqa: E731
        return capi.FrameImages(
            temp_color1=h(self.temp_color_img1), temp_color2=h(self.temp_color_img2),
            prev_color=h(self.prev_color_img), albedo_feature1=h(self.albedo_img1),
            normal_feature1=h(self.normal_img1), albedo_feature2=h(self.albedo_img2),
            normal_feature2=h(self.normal_img2), prev_albedo_feature=h(self.prev_albedo_img),
            prev_normal_feature=h(self.prev_normal_img), pixels=h(self.color_img))

    def raytrace_samples(self, times, read_back: bool = True):
        """len(times) calls of raytrace() as ONE device schedule (nova_hip_render_samples,
        SURVEY.md §8f rank 2): same images afterwards, one host wait, one post-process."""
        times = list(times)
        capi.render_samples(self.params, times, self.sample_index, self.frame_images(), self.width,
                            self.height, self.triangle_buf, self.tri_meta_buf, self.bvh_buf,
                            self.light_buf, self.num_lights, self.material_imgs, self.sky_img,
                            self.denoise_available)
        self.sample_index += len(times)
        return self.color_img.read() if read_back else None

    def step_time(self, seed: int) -> int:
        """Fixed-seed rule of the HIP backend: time = seed + sample_index (the reference seeds
        with wall-clock milliseconds, raytracer.cpp:175-176)."""
        return (seed + self.sample_index) & 0xFFFFFFFF

"""Host-side mirror of the reference's `PathTracer` interface over the C ABI.

Reference: include/pathtracer.h:29-43 -- `Init(width, height)`, `Run()` (starts the
workers, returns), `Join()` (blocks until the stop flag is seen), with the stop flag
`outputing` polled once per sample (src/pathtracer.cpp:95-99) and the RGB8 image read by
the GUI every 0.3 s (src/raytracer.cpp:449).  The same three calls are kept here; the
workers are the HIP kernels behind libmi355pt.so.  There is no CPU fallback: every
method raises if the HIP library is missing or no gfx950 device is present.
"""
import ctypes as C
import os
import threading

import numpy as np

from . import capi


class DeviceScene:
    """`LoadScene` result uploaded to one GPU (the scene globals of src/raytracer.cpp:33-43)."""

    def __init__(self, scene, device=0):
        self.lib = capi.load_library()
        self.desc, self._keep = scene.to_desc()
        self.device = device
        h = C.c_void_p()
        capi.check(self.lib, self.lib.mi355pt_scene_create(C.byref(self.desc), device, C.byref(h)), "mi355pt_scene_create")
        self.h = h

    @classmethod
    def load(cls, path, device=0):
        """mi355pt_scene_load: a scene another process flattened and saved (one host build for N ranks)."""
        self = cls.__new__(cls)
        self.lib = capi.load_library()
        self.desc, self._keep, self.device = None, None, device
        h = C.c_void_p()
        capi.check(self.lib, self.lib.mi355pt_scene_load(os.fsencode(path), device, C.byref(h)), "mi355pt_scene_load")
        self.h = h
        return self

    @staticmethod
    def save(scene, path, with_trees=False):
        """mi355pt_scene_save: flatten `scene` on the host (no GPU needed) and write it to `path`."""
        lib = capi.load_library()
        desc, keep = scene.to_desc()
        capi.check(lib, lib.mi355pt_scene_save(C.byref(desc), os.fsencode(path), capi.SAVE_TREES if with_trees else 0), "mi355pt_scene_save")
        del keep

    def get_bvh(self, mesh=0):
        nn, nf = capi.u32(0), capi.u32(0)
        capi.check(self.lib, self.lib.mi355pt_scene_get_bvh(self.h, mesh, None, C.byref(nn), None, C.byref(nf)), "get_bvh")
        nodes = (capi.BvhNode * max(1, nn.value))()
        faces = np.zeros(max(1, nf.value), dtype=np.uint32)
        capi.check(self.lib, self.lib.mi355pt_scene_get_bvh(self.h, mesh, nodes, C.byref(nn), faces.ctypes.data_as(C.POINTER(capi.u32)),
                                                            C.byref(nf)), "get_bvh")
        return nodes, faces[:nf.value], nn.value

    def close(self):
        if getattr(self, "h", None):
            self.lib.mi355pt_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PathTracer:
    """Drop-in for the reference's PathTracer (same method names and meaning)."""

    def __init__(self, scene, device=0, seed=0x5EED, max_bounces=30, rr_start=3, sample_offset=0, sample_stride=1,
                 spp_per_step=1, tile_size=0, tile_offset=0, tile_stride=1):
        self.dscene = scene if isinstance(scene, DeviceScene) else DeviceScene(scene, device)
        self.lib = self.dscene.lib
        self._cfg = dict(seed=seed, max_bounces=max_bounces, rr_start=rr_start, sample_offset=sample_offset,
                         sample_stride=sample_stride, device=self.dscene.device,
                         tile_size=tile_size, tile_offset=tile_offset, tile_stride=tile_stride)
        self.spp_per_step = spp_per_step
        self.h = None
        self.outputing = threading.Event()      # the reference's global std::atomic<bool> outputing
        self._thread = None
        self.width = self.height = self.size = 0

    # -- reference interface ------------------------------------------------------
    def Init(self, width, height):
        self.close_render()
        self.width, self.height, self.size = int(width), int(height), int(width) * int(height)
        self.params = capi.make_params(self.width, self.height, **self._cfg)
        h = C.c_void_p()
        capi.check(self.lib, self.lib.mi355pt_render_create(self.dscene.h, C.byref(self.params), C.byref(h)), "mi355pt_render_create")
        self.h = h
        self.outputing.clear()

    def Run(self):
        """Starts progressive rendering (non-blocking), like PathTracer::Run."""
        def loop():
            while not self.outputing.is_set():
                self.step(self.spp_per_step)
                self.sync()
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def Join(self):
        if self._thread is not None:
            self._thread.join()
            self._thread = None

    def Pause(self):
        """RayTracer::Pause (src/raytracer.cpp:455-459)."""
        self.outputing.set()

    # -- bounded variants used by tests / bench -------------------------------------
    def step(self, spp=1):
        capi.check(self.lib, self.lib.mi355pt_render_step(self.h, spp), "mi355pt_render_step")

    def sync(self):
        capi.check(self.lib, self.lib.mi355pt_render_sync(self.h), "mi355pt_render_sync")

    def render(self, spp):
        self.step(spp)
        self.sync()
        return self.GetPixels()

    def GetPixels(self):
        """renderImage.GetPixels(): (h, w, 3) uint8."""
        out = np.zeros((self.height, self.width, 3), dtype=np.uint8)
        capi.check(self.lib, self.lib.mi355pt_render_read_rgb8(self.h, out.ctypes.data), "read_rgb8")
        return out

    def SaveImage(self, path="colorbuffer.png"):
        """RayTracer::SaveBuffer -> renderImage.SaveImage("colorbuffer.png") (src/raytracer.cpp:461-474): the RGB8
        frame as an 8-bit PNG (the reference writes it with lodepng; rows top to bottom)."""
        save_png_rgb8(path, self.GetPixels())

    def Restart(self):
        """RayTracer::Restart (src/raytracer.cpp:377-382): stop, then start over with an empty accumulator."""
        self.Pause()
        self.Join()
        self.Init(self.width, self.height)

    def read_accum(self):
        mean = np.zeros((self.size, 3), dtype=np.float32)
        cnt = np.zeros(self.size, dtype=np.uint32)
        capi.check(self.lib, self.lib.mi355pt_render_read_accum(self.h, mean.ctypes.data, cnt.ctypes.data), "read_accum")
        return mean, cnt

    def write_accum(self, mean, cnt):
        mean = np.ascontiguousarray(mean, dtype=np.float32)
        cnt = np.ascontiguousarray(cnt, dtype=np.uint32)
        capi.check(self.lib, self.lib.mi355pt_render_write_accum(self.h, mean.ctypes.data, cnt.ctypes.data), "write_accum")

    def accum_device_ptrs(self):
        a, b = C.c_void_p(), C.c_void_p()
        capi.check(self.lib, self.lib.mi355pt_render_accum_device_ptrs(self.h, C.byref(a), C.byref(b)), "accum_device_ptrs")
        return a.value, b.value

    def trace_primary(self, x0=0, y0=0, x1=None, y1=None):
        x1 = self.width if x1 is None else x1
        y1 = self.height if y1 is None else y1
        n = (x1 - x0) * (y1 - y0)
        node, mesh, face = (np.zeros(n, dtype=np.int32) for _ in range(3))
        t = np.zeros(n, dtype=np.float32)
        capi.check(self.lib, self.lib.mi355pt_trace_primary(self.h, x0, y0, x1, y1, node.ctypes.data, mesh.ctypes.data,
                                                            face.ctypes.data, t.ctypes.data), "trace_primary")
        return node, mesh, face, t

    def trace_rays(self, origins, dirs):
        origins = np.ascontiguousarray(origins, dtype=np.float32)
        dirs = np.ascontiguousarray(dirs, dtype=np.float32)
        n = origins.shape[0]
        node, mesh, face = (np.zeros(n, dtype=np.int32) for _ in range(3))
        t = np.zeros(n, dtype=np.float32)
        capi.check(self.lib, self.lib.mi355pt_trace_rays(self.h, origins.ctypes.data, dirs.ctypes.data, n, node.ctypes.data,
                                                         mesh.ctypes.data, face.ctypes.data, t.ctypes.data), "trace_rays")
        return node, mesh, face, t

    def trace_visibility(self, origins, dirs, tmax, light=0):
        """LightVisTest toward light `light` for arbitrary rays: uint8 array, 1 = blocked."""
        origins = np.ascontiguousarray(origins, dtype=np.float32)
        dirs = np.ascontiguousarray(dirs, dtype=np.float32)
        tmax = np.ascontiguousarray(tmax, dtype=np.float32)
        blocked = np.zeros(origins.shape[0], dtype=np.uint8)
        capi.check(self.lib, self.lib.mi355pt_trace_visibility(self.h, origins.ctypes.data, dirs.ctypes.data, tmax.ctypes.data, int(light),
                                                               origins.shape[0], blocked.ctypes.data), "trace_visibility")
        return blocked

    def sample_radiance(self, pixel, sample):
        pixel = np.ascontiguousarray(pixel, dtype=np.uint32)
        sample = np.ascontiguousarray(sample, dtype=np.uint32)
        lin = np.zeros((pixel.size, 3), dtype=np.float32)
        mp = np.zeros((pixel.size, 3), dtype=np.float32)
        capi.check(self.lib, self.lib.mi355pt_sample_radiance(self.h, pixel.ctypes.data, sample.ctypes.data, pixel.size,
                                                              lin.ctypes.data, mp.ctypes.data), "sample_radiance")
        return lin, mp

    def set_profiling(self, events=True, count_work=False):
        """events: False/0 none, 1 = HIP events around the closest-hit trace launches only, True/2 = around every stage."""
        level = 2 if events is True else int(events)
        capi.check(self.lib, self.lib.mi355pt_render_set_profiling(self.h, level, int(count_work)), "set_profiling")

    def set_overlap(self, from_bounce=0):
        """Bounce from which visibility + connect run beside the next bounce's closest-hit trace (include/mi355pt.h)."""
        capi.check(self.lib, self.lib.mi355pt_render_set_overlap(self.h, int(from_bounce)), "set_overlap")

    def stats(self):
        st = capi.Stats()
        capi.check(self.lib, self.lib.mi355pt_render_stats(self.h, C.byref(st)), "render_stats")
        return {k: (list(getattr(st, k)) if k == "debug" else getattr(st, k)) for k, _ in capi.Stats._fields_}

    def reset_stats(self):
        capi.check(self.lib, self.lib.mi355pt_render_reset_stats(self.h), "reset_stats")

    def stream(self):
        return self.lib.mi355pt_render_stream(self.h)

    def close_render(self):
        if getattr(self, "h", None):
            self.lib.mi355pt_render_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close_render()
        except Exception:
            pass


class PathTracerGroup:
    """mi355pt_group: one process driving several GPUs (or several shards of one GPU) -- one host build, one upload per
    device, the work split by sample index or by interleaved image tile, the read-out one RCCL reduce (include/mi355pt.h)."""

    def __init__(self, scene, devices=(0,), split="sample", seed=0x5EED, max_bounces=30, rr_start=3, force_rccl=False):
        self.lib = capi.load_library()
        self.scene = scene
        self.devices = [int(d) for d in devices]
        self.split = {"sample": capi.SPLIT_SAMPLE, "tile": capi.SPLIT_TILE}[split]
        self._cfg = dict(seed=seed, max_bounces=max_bounces, rr_start=rr_start)
        self.flags = capi.GROUP_FORCE_RCCL if force_rccl else 0
        self.h = None
        self.width = self.height = self.size = 0

    def Init(self, width, height):
        self.width, self.height, self.size = int(width), int(height), int(width) * int(height)
        if self.h:                                            # PathTracer::Init again: fresh accumulators, the uploaded scenes kept
            capi.check(self.lib, self.lib.mi355pt_group_reset(self.h, self.width, self.height), "mi355pt_group_reset")
            return
        desc, keep = self.scene.to_desc()
        prm = capi.make_params(self.width, self.height, **self._cfg)
        devs = (capi.i32 * len(self.devices))(*self.devices)
        h = C.c_void_p()
        capi.check(self.lib, self.lib.mi355pt_group_create(C.byref(desc), devs, len(self.devices), C.byref(prm), self.split, self.flags, C.byref(h)),
                   "mi355pt_group_create")
        del keep                                              # the library copied everything it needs
        self.h = h

    def step(self, spp=1):
        capi.check(self.lib, self.lib.mi355pt_group_step(self.h, spp), "mi355pt_group_step")

    def sync(self):
        capi.check(self.lib, self.lib.mi355pt_group_sync(self.h), "mi355pt_group_sync")

    def GetPixels(self):
        out = np.zeros((self.height, self.width, 3), dtype=np.uint8)
        capi.check(self.lib, self.lib.mi355pt_group_read_rgb8(self.h, out.ctypes.data), "group_read_rgb8")
        return out

    def read_accum(self):
        mean = np.zeros((self.size, 3), dtype=np.float32)
        cnt = np.zeros(self.size, dtype=np.uint32)
        capi.check(self.lib, self.lib.mi355pt_group_read_accum(self.h, mean.ctypes.data, cnt.ctypes.data), "group_read_accum")
        return mean, cnt

    def stats(self):
        st = capi.Stats()
        capi.check(self.lib, self.lib.mi355pt_group_stats(self.h, C.byref(st)), "group_stats")
        return {k: (list(getattr(st, k)) if k == "debug" else getattr(st, k)) for k, _ in capi.Stats._fields_}

    def readout_ms(self):
        return float(self.lib.mi355pt_group_readout_ms(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.mi355pt_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def save_png_rgb8(path, img):
    """Minimal PNG writer: (h, w, 3) uint8, filter 0 on every row, one zlib stream."""
    import struct
    import zlib
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    raw = np.concatenate([np.zeros((h, 1), dtype=np.uint8), img.reshape(h, w * 3)], axis=1).tobytes()

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))

"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
the pathtracer_amd package."""
import ctypes as C
import os
import subprocess

import numpy as np

from pathtracer_amd import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s", "all"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        vp = C.c_void_p
        L.oracle_scene_create.restype = vp
        L.oracle_scene_create.argtypes = [C.POINTER(capi.SceneDesc)]
        L.oracle_scene_destroy.argtypes = [vp]
        L.oracle_get_bvh.argtypes = [vp, capi.u32, C.c_void_p, C.POINTER(capi.u32), C.c_void_p, C.POINTER(capi.u32)]
        L.oracle_trace_primary.argtypes = [vp, C.POINTER(capi.RenderParams)] + [capi.u32] * 4 + [C.c_void_p] * 5
        L.oracle_count_ordered_primary.argtypes = [vp, C.POINTER(capi.RenderParams)] + [capi.u32] * 4 + [C.c_void_p] * 2
        L.oracle_sample_radiance.argtypes = [vp, C.POINTER(capi.RenderParams), C.c_void_p, C.c_void_p, capi.u32, C.c_void_p, C.c_void_p]
        L.oracle_render.argtypes = [vp, C.POINTER(capi.RenderParams)] + [capi.u32] * 6 + [C.c_void_p] * 4
        L.oracle_trace_rays.argtypes = [vp, C.c_void_p, C.c_void_p, capi.u32] + [C.c_void_p] * 4
        L.oracle_light_vis.argtypes = [vp, C.c_void_p, C.c_void_p, C.c_void_p, capi.i32, capi.u32, C.c_void_p]
        L.oracle_rng_u31.restype = capi.u32
        L.oracle_rng_u31.argtypes = [capi.u32] * 4
        L.oracle_halton.restype = C.c_float
        L.oracle_halton.argtypes = [C.c_int, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleScene:
    def __init__(self, scene):
        self.desc, self._keep = scene.to_desc()
        self.h = lib().oracle_scene_create(C.byref(self.desc))
        if not self.h:
            raise RuntimeError("oracle_scene_create failed")

    def close(self):
        if self.h:
            lib().oracle_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def get_bvh(self, mesh=0):
        nn, nf = capi.u32(0), capi.u32(0)
        lib().oracle_get_bvh(self.h, mesh, None, C.byref(nn), None, C.byref(nf))
        nodes = (capi.BvhNode * nn.value)()
        faces = np.zeros(nf.value, dtype=np.uint32)
        lib().oracle_get_bvh(self.h, mesh, nodes, C.byref(nn), _p(faces), C.byref(nf))
        return nodes, faces

    def trace_primary(self, params, x0, y0, x1, y1):
        n = (x1 - x0) * (y1 - y0)
        node, mesh, face = (np.zeros(n, dtype=np.int32) for _ in range(3))
        t = np.zeros(n, dtype=np.float32)
        cnt = np.zeros(2, dtype=np.uint64)
        lib().oracle_trace_primary(self.h, C.byref(params), x0, y0, x1, y1, _p(node), _p(mesh), _p(face), _p(t), _p(cnt))
        return node, mesh, face, t, cnt

    def trace_rays(self, origins, dirs):
        origins = np.ascontiguousarray(origins, dtype=np.float32)
        dirs = np.ascontiguousarray(dirs, dtype=np.float32)
        n = origins.shape[0]
        node, mesh, face = (np.zeros(n, dtype=np.int32) for _ in range(3))
        t = np.zeros(n, dtype=np.float32)
        lib().oracle_trace_rays(self.h, _p(origins), _p(dirs), n, _p(node), _p(mesh), _p(face), _p(t))
        return node, mesh, face, t

    def light_vis(self, origins, dirs, tmax, light=0):
        origins = np.ascontiguousarray(origins, dtype=np.float32)
        dirs = np.ascontiguousarray(dirs, dtype=np.float32)
        tmax = np.ascontiguousarray(tmax, dtype=np.float32)
        blocked = np.zeros(origins.shape[0], dtype=np.uint8)
        assert lib().oracle_light_vis(self.h, _p(origins), _p(dirs), _p(tmax), int(light), origins.shape[0], _p(blocked)) == 0
        return blocked

    def count_ordered_primary(self, params, x0, y0, x1, y1):
        n = (x1 - x0) * (y1 - y0)
        cnt = np.zeros(2, dtype=np.uint64)
        face = np.zeros(n, dtype=np.int32)
        lib().oracle_count_ordered_primary(self.h, C.byref(params), x0, y0, x1, y1, _p(cnt), _p(face))
        return cnt, face

    def sample_radiance(self, params, pixel, sample):
        pixel = np.ascontiguousarray(pixel, dtype=np.uint32)
        sample = np.ascontiguousarray(sample, dtype=np.uint32)
        lin = np.zeros((pixel.size, 3), dtype=np.float32)
        mp = np.zeros((pixel.size, 3), dtype=np.float32)
        lib().oracle_sample_radiance(self.h, C.byref(params), _p(pixel), _p(sample), pixel.size, _p(lin), _p(mp))
        return lin, mp

    def render(self, params, spp, rect=None, threads=1, mean=None, count=None):
        x0, y0, x1, y1 = rect or (0, 0, params.width, params.height)
        n = (x1 - x0) * (y1 - y0)
        mean = np.zeros((n, 3), dtype=np.float32) if mean is None else mean
        count = np.zeros(n, dtype=np.uint32) if count is None else count
        rgb8 = np.zeros((n, 3), dtype=np.uint8)
        stats = np.zeros(5, dtype=np.uint64)
        lib().oracle_render(self.h, C.byref(params), x0, y0, x1, y1, spp, threads, _p(mean), _p(count), _p(rgb8), _p(stats))
        return mean, count, rgb8, dict(closest=int(stats[0]), shadow=int(stats[1]), samples=int(stats[2]),
                                       node_visits=int(stats[3]), tri_tests=int(stats[4]))

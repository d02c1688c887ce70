"""ctypes mirror of include/mi355pt.h and loader of the HIP library.

The structures here are plain data carriers shared by the product binding
(`libmi355pt.so`) and -- in tests/bench only -- the CPU oracle, which takes the
same scene description.  There is no CPU fallback: `load_library()` raises if the
HIP library has not been built.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmi355pt.so")

OK = 0
OBJ_NONE, OBJ_PLANE, OBJ_SPHERE, OBJ_MODEL = 0, 1, 2, 3
MODEL_HAS_CDF = 1
MTL_DISNEY, MTL_STANDARD = 0, 1
TEX_RGB8, TEX_RGB32F, TEX_CHECKER = 0, 1, 2
BIGFLOAT = 1.0e30

f32, i32, u32, u64 = C.c_float, C.c_int32, C.c_uint32, C.c_uint64


class Node(C.Structure):
    _fields_ = [("tm", f32 * 9), ("itm", f32 * 9), ("pos", f32 * 3), ("local_to_world", f32 * 16),
                ("parent", i32), ("object_kind", i32), ("object_index", i32), ("material", i32), ("light", i32)]


class BvhNode(C.Structure):
    _fields_ = [("bound", f32 * 6), ("left", i32), ("right", i32), ("face_first", u32), ("face_count", u32)]


class Mesh(C.Structure):
    _fields_ = [("vertices", C.POINTER(f32)), ("normals", C.POINTER(f32)), ("uvw", C.POINTER(f32)),
                ("faces", C.POINTER(u32)), ("tangents", C.POINTER(f32)), ("bitangents", C.POINTER(f32)),
                ("num_vertices", u32), ("num_faces", u32), ("aabb", f32 * 6),
                ("bvh_nodes", C.POINTER(BvhNode)), ("bvh_faces", C.POINTER(u32)),
                ("num_bvh_nodes", u32), ("num_bvh_faces", u32)]


class Model(C.Structure):
    _fields_ = [("first_mesh", i32), ("num_meshes", i32), ("aabb", f32 * 6), ("flags", u32)]


class TexMap(C.Structure):
    _fields_ = [("itm", f32 * 9), ("pos", f32 * 3), ("texture", i32)]


class Texture(C.Structure):
    _fields_ = [("kind", i32), ("width", i32), ("height", i32), ("data", C.c_void_p),
                ("color1", f32 * 3), ("color2", f32 * 3)]


class Material(C.Structure):
    _fields_ = [("kind", i32), ("albedo", f32 * 3), ("roughness", f32 * 3), ("metalness", f32 * 3),
                ("albedo_map", i32), ("roughness_map", i32), ("metalness_map", i32), ("normal_map", i32),
                ("specular", f32), ("specular_tint", f32), ("sheen", f32), ("sheen_tint", f32),
                ("clearcoat", f32), ("clearcoat_gloss", f32), ("subsurface", f32)]


class Light(C.Structure):
    _fields_ = [("node", i32), ("intensity", f32 * 3)]


class Camera(C.Structure):
    _fields_ = [("pos", f32 * 3), ("dir", f32 * 3), ("up", f32 * 3), ("fov", f32), ("focaldist", f32),
                ("dof", f32), ("width", i32), ("height", i32)]


class SceneDesc(C.Structure):
    _fields_ = [("nodes", C.POINTER(Node)), ("num_nodes", u32),
                ("models", C.POINTER(Model)), ("num_models", u32),
                ("meshes", C.POINTER(Mesh)), ("num_meshes", u32),
                ("materials", C.POINTER(Material)), ("num_materials", u32),
                ("texmaps", C.POINTER(TexMap)), ("num_texmaps", u32),
                ("textures", C.POINTER(Texture)), ("num_textures", u32),
                ("lights", C.POINTER(Light)), ("num_lights", u32),
                ("env_color", f32 * 3), ("env_map", i32), ("camera", Camera)]


class RenderParams(C.Structure):
    _fields_ = [("width", u32), ("height", u32), ("max_bounces", u32), ("rr_start", u32), ("seed", u32),
                ("sample_offset", u32), ("sample_stride", u32), ("device", i32),
                ("tile_size", u32), ("tile_offset", u32), ("tile_stride", u32)]


class Stats(C.Structure):
    _fields_ = [("closest_rays", u64), ("shadow_rays", u64), ("samples", u64), ("bvh_node_visits", u64),
                ("bvh_tri_tests", u64), ("shadow_node_visits", u64), ("shadow_tri_tests", u64), ("max_nodes_per_ray", u64), ("extend_ms", C.c_double), ("shade_ms", C.c_double),
                ("connect_ms", C.c_double), ("other_ms", C.c_double), ("extend_launches", u64), ("debug", u64 * 8),
                ("shaded_vertices", u64), ("texture_taps", u64), ("shadow_rays_traversed", u64)]


def make_params(width, height, seed=0x5EED, max_bounces=30, rr_start=3, sample_offset=0, sample_stride=1, device=0,
                tile_size=0, tile_offset=0, tile_stride=1):
    return RenderParams(width, height, max_bounces, rr_start, seed, sample_offset, sample_stride, device,
                        tile_size, tile_offset, tile_stride)


# Every symbol include/mi355pt.h declares (checked by tests/test_abi.py).
SYMBOLS = [
    "mi355pt_version", "mi355pt_last_error", "mi355pt_device_count",
    "mi355pt_scene_create", "mi355pt_scene_destroy", "mi355pt_scene_get_bvh", "mi355pt_build_bvh",
    "mi355pt_render_create", "mi355pt_render_destroy", "mi355pt_render_step", "mi355pt_render_sync",
    "mi355pt_render_read_rgb8", "mi355pt_render_read_accum", "mi355pt_render_write_accum",
    "mi355pt_render_accum_device_ptrs", "mi355pt_trace_primary", "mi355pt_trace_rays", "mi355pt_sample_radiance",
    "mi355pt_render_stats", "mi355pt_render_reset_stats", "mi355pt_render_set_profiling", "mi355pt_render_set_overlap", "mi355pt_render_stream",
    "mi355pt_trace_rays_static", "mi355pt_trace_visibility",
    "mi355pt_abi_version", "mi355pt_scene_create_multi", "mi355pt_scene_save", "mi355pt_scene_load",
    "mi355pt_render_stats_sized", "mi355pt_render_get_params",
    "mi355pt_group_create", "mi355pt_group_destroy", "mi355pt_group_reset", "mi355pt_group_step", "mi355pt_group_sync", "mi355pt_group_read_rgb8",
    "mi355pt_group_read_accum", "mi355pt_group_stats", "mi355pt_group_size", "mi355pt_group_render", "mi355pt_group_readout_ms",
    "mi355pt_comm_unique_id", "mi355pt_comm_create", "mi355pt_comm_destroy", "mi355pt_render_reduce",
]
ABI_VERSION = 3
SPLIT_SAMPLE, SPLIT_TILE = 0, 1
GROUP_FORCE_RCCL = 1
SAVE_TREES = 1
COMM_ID_BYTES = 128
STATIC_MAX_RAYS = 131072

_lib = None


def load_library(path=None):
    """Load libmi355pt.so (built by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("MI355PT_LIBRARY") or LIB_PATH      # the override is for A/B runs of experimental builds
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: the HIP backend has not been built (run `python __graft_entry__.py build`). "
            "There is no CPU fallback.")
    lib = C.CDLL(p)
    vp = C.c_void_p
    lib.mi355pt_version.restype = C.c_char_p
    lib.mi355pt_last_error.restype = C.c_char_p
    lib.mi355pt_device_count.restype = C.c_int
    lib.mi355pt_scene_create.argtypes = [C.POINTER(SceneDesc), i32, C.POINTER(vp)]
    lib.mi355pt_scene_destroy.argtypes = [vp]
    lib.mi355pt_scene_destroy.restype = None
    lib.mi355pt_scene_get_bvh.argtypes = [vp, u32, C.POINTER(BvhNode), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    lib.mi355pt_build_bvh.argtypes = [C.c_void_p, u32, C.c_void_p, u32, C.POINTER(f32), C.POINTER(BvhNode), C.POINTER(u32),
                                      C.c_void_p, C.POINTER(u32)]
    lib.mi355pt_render_create.argtypes = [vp, C.POINTER(RenderParams), C.POINTER(vp)]
    lib.mi355pt_render_destroy.argtypes = [vp]
    lib.mi355pt_render_destroy.restype = None
    lib.mi355pt_render_step.argtypes = [vp, u32]
    lib.mi355pt_render_sync.argtypes = [vp]
    lib.mi355pt_render_read_rgb8.argtypes = [vp, C.c_void_p]
    lib.mi355pt_render_read_accum.argtypes = [vp, C.c_void_p, C.c_void_p]
    lib.mi355pt_render_write_accum.argtypes = [vp, C.c_void_p, C.c_void_p]
    lib.mi355pt_render_accum_device_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    lib.mi355pt_trace_primary.argtypes = [vp, u32, u32, u32, u32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mi355pt_trace_rays.argtypes = [vp, C.c_void_p, C.c_void_p, u32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mi355pt_sample_radiance.argtypes = [vp, C.c_void_p, C.c_void_p, u32, C.c_void_p, C.c_void_p]
    lib.mi355pt_render_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.mi355pt_render_reset_stats.argtypes = [vp]
    lib.mi355pt_render_set_profiling.argtypes = [vp, C.c_int, C.c_int]
    lib.mi355pt_render_set_overlap.argtypes = [vp, C.c_uint32]
    lib.mi355pt_render_stream.argtypes = [vp]
    lib.mi355pt_render_stream.restype = vp
    lib.mi355pt_trace_visibility.argtypes = [vp, C.c_void_p, C.c_void_p, C.c_void_p, i32, u32, C.c_void_p]
    lib.mi355pt_trace_rays_static.argtypes = [vp, vp, vp, u32, vp, vp, vp, vp, vp]
    lib.mi355pt_abi_version.restype = C.c_int
    lib.mi355pt_scene_create_multi.argtypes = [C.POINTER(SceneDesc), C.POINTER(i32), u32, C.POINTER(vp)]
    lib.mi355pt_scene_save.argtypes = [C.POINTER(SceneDesc), C.c_char_p, u32]
    lib.mi355pt_scene_load.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
    lib.mi355pt_render_stats_sized.argtypes = [vp, C.c_void_p, u32]
    lib.mi355pt_render_get_params.argtypes = [vp, C.POINTER(RenderParams)]
    lib.mi355pt_group_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(i32), u32, C.POINTER(RenderParams), i32, u32, C.POINTER(vp)]
    lib.mi355pt_group_destroy.argtypes = [vp]
    lib.mi355pt_group_destroy.restype = None
    lib.mi355pt_group_reset.argtypes = [vp, u32, u32]
    lib.mi355pt_group_step.argtypes = [vp, u32]
    lib.mi355pt_group_sync.argtypes = [vp]
    lib.mi355pt_group_read_rgb8.argtypes = [vp, C.c_void_p]
    lib.mi355pt_group_read_accum.argtypes = [vp, C.c_void_p, C.c_void_p]
    lib.mi355pt_group_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.mi355pt_group_size.argtypes = [vp]
    lib.mi355pt_group_size.restype = u32
    lib.mi355pt_group_render.argtypes = [vp, u32]
    lib.mi355pt_group_render.restype = vp
    lib.mi355pt_group_readout_ms.argtypes = [vp]
    lib.mi355pt_group_readout_ms.restype = C.c_double
    lib.mi355pt_comm_unique_id.argtypes = [C.c_void_p]
    lib.mi355pt_comm_create.argtypes = [C.c_void_p, i32, i32, i32, C.POINTER(vp)]
    lib.mi355pt_comm_destroy.argtypes = [vp]
    lib.mi355pt_comm_destroy.restype = None
    lib.mi355pt_render_reduce.argtypes = [vp, vp, i32, i32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    if path is None:
        _lib = lib
    return lib


class BackendError(RuntimeError):
    pass


def check(lib, status, what):
    if status != OK:
        err = lib.mi355pt_last_error()
        raise BackendError(f"{what} failed with status {status}: {err.decode() if err else ''}")


def build_bvh(mesh):
    """mi355pt_build_bvh for a scene.Mesh -> (BvhNode array, face list)."""
    import numpy as np
    lib = load_library()
    nn, nf = u32(0), u32(0)
    aabb = (f32 * 6)(*mesh.aabb.tolist())
    args = (mesh.vertices.ctypes.data, mesh.vertices.shape[0], mesh.faces.ctypes.data, mesh.faces.shape[0], aabb)
    check(lib, lib.mi355pt_build_bvh(*args, None, C.byref(nn), None, C.byref(nf)), "mi355pt_build_bvh")
    nodes = (BvhNode * max(1, nn.value))()
    faces = np.zeros(max(1, nf.value), dtype=np.uint32)
    check(lib, lib.mi355pt_build_bvh(*args, nodes, C.byref(nn), faces.ctypes.data, C.byref(nf)), "mi355pt_build_bvh")
    return nodes, faces[:nf.value], nn.value

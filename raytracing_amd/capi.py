"""ctypes mirror of include/rt_amd.h and of the host-surface helper API (raytracing_amd/host/scenes.cpp).

Everything here goes through the two in-tree shared libraries; there is no Python or CPU
implementation of the path behind it.  Importing fails loudly when a library is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.environ.get("RT_AMD_LIB_DIR") or os.path.join(_HERE, "lib")   # override: A/B builds of the kernels


class BvhNode(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("offset", C.c_int32),
                ("nprim", C.c_uint16), ("axis", C.c_uint8), ("pad", C.c_uint8)]


BVH_NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("offset", "<i4"),
                           ("nprim", "<u2"), ("axis", "u1"), ("pad", "u1")])
assert BVH_NODE_DTYPE.itemsize == 32 and C.sizeof(BvhNode) == 32


class Sphere(C.Structure):
    _fields_ = [("radius", C.c_float), ("o2w", C.c_float * 16), ("w2o", C.c_float * 16)]


class Texture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("value", C.c_float * 3),
                ("texels", C.POINTER(C.c_float)), ("su", C.c_float), ("sv", C.c_float), ("du", C.c_float), ("dv", C.c_float)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("tex", C.c_int32 * 4), ("remap_roughness", C.c_int32), ("pad", C.c_int32)]


class Light(C.Structure):
    _fields_ = [("prim", C.c_int32), ("Le", C.c_float * 3), ("two_sided", C.c_int32), ("area", C.c_float), ("flip_normal", C.c_int32),
                ("kind", C.c_int32), ("env", C.c_int32)]


class EnvLight(C.Structure):
    _fields_ = [("light_to_world", C.c_float * 16), ("world_to_light", C.c_float * 16), ("width", C.c_int32), ("height", C.c_int32),
                ("texels", C.POINTER(C.c_float)), ("nu", C.c_int32), ("nv", C.c_int32), ("func", C.POINTER(C.c_float)),
                ("cdf", C.POINTER(C.c_float)), ("func_int", C.POINTER(C.c_float)), ("marginal_cdf", C.POINTER(C.c_float)),
                ("marginal_int", C.c_float), ("world_center", C.c_float * 3), ("world_radius", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("n_prims", C.c_int32), ("prim_kind", C.POINTER(C.c_uint8)), ("prim_geom", C.POINTER(C.c_int32)),
        ("prim_material", C.POINTER(C.c_int32)), ("prim_light", C.POINTER(C.c_int32)),
        ("n_tris", C.c_int32), ("tri_p", C.POINTER(C.c_float)), ("tri_uv", C.POINTER(C.c_float)),
        ("tri_n", C.POINTER(C.c_float)), ("tri_has_n", C.POINTER(C.c_uint8)),
        ("n_spheres", C.c_int32), ("spheres", C.POINTER(Sphere)),
        ("n_materials", C.c_int32), ("materials", C.POINTER(Material)),
        ("n_textures", C.c_int32), ("textures", C.POINTER(Texture)),
        ("n_lights", C.c_int32), ("lights", C.POINTER(Light)),
        ("n_nodes", C.c_int32), ("bvh_nodes", C.POINTER(BvhNode)), ("prim_order", C.POINTER(C.c_int32)),
        ("n_env_lights", C.c_int32), ("env_lights", C.POINTER(EnvLight)),
    ]


class Camera(C.Structure):
    _fields_ = [("raster_to_camera", C.c_float * 16), ("camera_to_world", C.c_float * 16),
                ("width", C.c_int32), ("height", C.c_int32)]


class RenderOpts(C.Structure):
    _fields_ = [("strata_x", C.c_int32), ("strata_y", C.c_int32), ("sample_begin", C.c_int32), ("sample_end", C.c_int32),
                ("max_depth", C.c_int32), ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32),
                ("pool_paths", C.c_int32), ("flags", C.c_int32), ("rr_threshold", C.c_float)]


class RenderStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64), ("iterations", C.c_uint64),
                ("seconds_total", C.c_double), ("seconds_trace", C.c_double), ("seconds_shade", C.c_double),
                ("seconds_shadow", C.c_double), ("seconds_other", C.c_double), ("rays_replayed", C.c_uint64)]


class TraceStats(C.Structure):
    _fields_ = [("nodes_visited", C.c_uint64), ("prims_tested", C.c_uint64), ("seconds", C.c_double), ("rays_replayed", C.c_uint64)]


RT_RENDER_ACCUMULATE = 1
RT_RENDER_TIME_KERNELS = 2
RT_RENDER_LI = 4
RT_RENDER_ALL_LIGHTS = 8

# Every symbol include/rt_amd.h declares.
RT_AMD_SYMBOLS = [
    "rt_version", "rt_init", "rt_shutdown", "rt_last_error", "rt_bvh_build", "rt_bvh_build_ex", "rt_prim_bounds", "rt_scene_desc_check", "rt_scene_create",
    "rt_scene_destroy", "rt_scene_get_bvh", "rt_trace_closest", "rt_trace_any", "rt_trace_closest_device", "rt_trace_closest_device_noalloc",
    "rt_render", "rt_render_device", "rt_tonemap_u8",
    "rt_device_count", "rt_split_samples", "rt_multi_create", "rt_multi_destroy", "rt_multi_last_error", "rt_multi_render", "rt_write_ppm",
]


def _load(name):
    path = os.path.join(LIB_DIR, name)
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no fallback implementation)")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


def load_rt_amd():
    lib = _load("librt_amd.so")
    vp, i32p, f32p, u8p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_uint8)
    lib.rt_version.restype = C.c_char_p
    lib.rt_init.argtypes = [C.c_int, C.POINTER(vp)]
    lib.rt_shutdown.argtypes = [vp]; lib.rt_shutdown.restype = None
    lib.rt_last_error.argtypes = [vp]; lib.rt_last_error.restype = C.c_char_p
    lib.rt_bvh_build.argtypes = [C.c_int32, f32p, C.POINTER(BvhNode), i32p, i32p]
    lib.rt_bvh_build_ex.argtypes = [C.c_int32, f32p, C.c_int32, C.c_int32, C.POINTER(BvhNode), i32p, i32p]
    lib.rt_prim_bounds.argtypes = [C.POINTER(SceneDesc), f32p]
    lib.rt_scene_desc_check.argtypes = [C.POINTER(SceneDesc), C.c_char_p, C.c_size_t]
    lib.rt_scene_create.argtypes = [vp, C.POINTER(SceneDesc), C.POINTER(vp)]
    lib.rt_scene_destroy.argtypes = [vp]; lib.rt_scene_destroy.restype = None
    lib.rt_scene_get_bvh.argtypes = [vp, C.POINTER(BvhNode), i32p, i32p]
    lib.rt_trace_closest.argtypes = [vp, C.c_size_t, f32p, f32p, f32p, i32p, f32p, f32p]
    lib.rt_trace_any.argtypes = [vp, C.c_size_t, f32p, f32p, f32p, u8p]
    lib.rt_trace_closest_device.argtypes = [vp, C.c_size_t, vp, vp, C.c_int, C.c_int, C.POINTER(TraceStats)]
    lib.rt_trace_closest_device_noalloc.argtypes = [vp, C.c_size_t, vp, vp, C.c_int, C.POINTER(TraceStats)]
    lib.rt_render.argtypes = [vp, C.POINTER(Camera), C.POINTER(RenderOpts), f32p, C.POINTER(RenderStats)]
    lib.rt_render_device.argtypes = [vp, C.POINTER(Camera), C.POINTER(RenderOpts), C.POINTER(vp), C.POINTER(RenderStats)]
    lib.rt_split_samples.argtypes = [C.c_int32, C.c_int32, C.c_int32, i32p, i32p]
    lib.rt_multi_create.argtypes = [i32p, C.c_int32, C.POINTER(SceneDesc), C.POINTER(vp)]
    lib.rt_multi_destroy.argtypes = [vp]; lib.rt_multi_destroy.restype = None
    lib.rt_multi_last_error.argtypes = [vp]; lib.rt_multi_last_error.restype = C.c_char_p
    lib.rt_multi_render.argtypes = [vp, C.POINTER(Camera), C.POINTER(RenderOpts), f32p, C.POINTER(RenderStats), C.POINTER(RenderStats)]
    lib.rt_write_ppm.argtypes = [C.c_char_p, f32p, C.c_int32, C.c_int32, C.c_int32]
    lib.rt_tonemap_u8.argtypes = [f32p, C.c_int32, C.c_int32, C.c_int32, u8p]
    return lib


def load_rt_host():
    load_rt_amd()
    lib = _load("librt_host.so")
    vp = C.c_void_p
    lib.rth_last_error.restype = C.c_char_p
    lib.rth_scene_build.argtypes = [C.c_char_p, C.c_char_p, C.c_int64]; lib.rth_scene_build.restype = vp
    lib.rth_scene_desc.argtypes = [vp]; lib.rth_scene_desc.restype = C.POINTER(SceneDesc)
    lib.rth_scene_texture_source.argtypes = [vp, C.c_int]; lib.rth_scene_texture_source.restype = C.c_char_p
    lib.rth_scene_env_source.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]; lib.rth_scene_env_source.restype = C.c_char_p
    lib.rth_scene_camera.argtypes = [vp, C.c_int, C.c_int, C.POINTER(Camera)]
    lib.rth_scene_free.argtypes = [vp]; lib.rth_scene_free.restype = None
    lib.rth_register_image.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint8)]; lib.rth_register_image.restype = None
    lib.rth_write_hdr.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.rth_read_hdr.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float)]
    lib.rth_decode_image.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint8)]
    lib.rth_model_probe.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_char_p, C.c_int]
    lib.rth_render_like_main.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p,
                                         C.POINTER(C.c_float), C.POINTER(RenderStats)]
    lib.rth_kat_transform.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float),
                                      C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.rth_kat_camera.argtypes = [C.POINTER(C.c_float)] * 3 + [C.c_float, C.c_float, C.c_int, C.c_int, C.POINTER(Camera)]
    return lib


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def u8ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))

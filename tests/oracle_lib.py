"""ctypes binding of oracle/_build/liboracle.so — the CPU restatement of the reference algorithm.

Checker only: imported by tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
raytracing_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracing_amd import capi

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "_build", "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle"), "oracle"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp, f32p, i32p, u8p, u32p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
        L.orc_bvh_build.argtypes = [C.c_int32, f32p, C.POINTER(capi.BvhNode), i32p, i32p]
        L.orc_bvh_build_ex.argtypes = [C.c_int32, f32p, C.c_int32, C.c_int32, C.POINTER(capi.BvhNode), i32p, i32p]
        L.orc_prim_bounds.argtypes = [C.POINTER(capi.SceneDesc), f32p]
        L.orc_scene_create.argtypes = [C.POINTER(capi.SceneDesc)]; L.orc_scene_create.restype = vp
        L.orc_scene_destroy.argtypes = [vp]; L.orc_scene_destroy.restype = None
        L.orc_scene_get_bvh.argtypes = [vp, C.POINTER(capi.BvhNode), i32p, i32p]
        L.orc_trace_closest.argtypes = [vp, C.c_size_t, f32p, f32p, f32p, i32p, f32p, f32p, u32p, u32p, C.c_int]
        L.orc_trace_any.argtypes = [vp, C.c_size_t, f32p, f32p, f32p, u8p]
        L.orc_render.argtypes = [vp, C.POINTER(capi.Camera), C.POINTER(capi.RenderOpts), C.c_int, C.c_int64, C.c_int, f32p,
                                 C.POINTER(C.c_uint64)]
        L.orc_tonemap_u8.argtypes = [f32p, C.c_int32, C.c_int32, C.c_int32, u8p]
        L.orc_camera_ray.argtypes = [C.POINTER(capi.Camera), C.c_float, C.c_float, f32p, f32p]
        L.orc_philox.argtypes = [u32p, C.c_uint32, C.c_uint32, u32p]
        L.orc_film_sample.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, f32p, f32p]
        L.orc_texture_eval.argtypes = [vp, C.c_int, C.c_float, C.c_float, f32p]
        L.orc_env_sample.argtypes = [vp, C.c_int, f32p, C.c_float, C.c_float, f32p]
        L.orc_interaction.argtypes = [vp, f32p, f32p, C.c_float, f32p]
        L.orc_bsdf_probe.argtypes = [vp, f32p, f32p, f32p, C.c_float, C.c_float, f32p]
        _lib = L
    return _lib


def bvh_build(bounds, max_prims=1, method=2):
    """method: the reference's enum SplitMethod { SAH, HLBVH, Middle, EqualCounts } (bvh.h:17)."""
    bounds = np.ascontiguousarray(bounds, np.float32).reshape(-1, 6)
    n = bounds.shape[0]
    nodes = np.zeros(max(2 * n, 1), capi.BVH_NODE_DTYPE)
    order = np.zeros(max(n, 1), np.int32)
    nn = C.c_int32(0)
    lib().orc_bvh_build_ex(n, capi.fptr(bounds), max_prims, method, nodes.ctypes.data_as(C.POINTER(capi.BvhNode)), C.byref(nn), capi.iptr(order))
    return nodes[:nn.value].copy(), order[:n].copy()


class OracleScene:
    def __init__(self, desc_ptr):
        self._p = lib().orc_scene_create(desc_ptr)
        self.n_prims = desc_ptr.contents.n_prims

    def get_bvh(self):
        nn = C.c_int32(0)
        lib().orc_scene_get_bvh(self._p, None, C.byref(nn), None)
        nodes = np.zeros(nn.value, capi.BVH_NODE_DTYPE)
        order = np.zeros(self.n_prims, np.int32)
        lib().orc_scene_get_bvh(self._p, nodes.ctypes.data_as(C.POINTER(capi.BvhNode)), C.byref(nn), capi.iptr(order))
        return nodes, order

    def trace_closest(self, o, d, tmax, threads=8):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32); tmax = np.ascontiguousarray(tmax, np.float32)
        n = o.shape[0]
        prim = np.zeros(n, np.int32); t = np.zeros(n, np.float32); b = np.zeros((n, 2), np.float32)
        nodes = np.zeros(n, np.uint32); tests = np.zeros(n, np.uint32)
        lib().orc_trace_closest(self._p, n, capi.fptr(o), capi.fptr(d), capi.fptr(tmax), capi.iptr(prim), capi.fptr(t), capi.fptr(b),
                                nodes.ctypes.data_as(C.POINTER(C.c_uint32)), tests.ctypes.data_as(C.POINTER(C.c_uint32)), threads)
        return prim, t, b, nodes, tests

    def trace_any(self, o, d, tmax):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32); tmax = np.ascontiguousarray(tmax, np.float32)
        n = o.shape[0]
        hit = np.zeros(n, np.uint8)
        lib().orc_trace_any(self._p, n, capi.fptr(o), capi.fptr(d), capi.fptr(tmax), capi.u8ptr(hit))
        return hit

    def render(self, cam, opts, mode=1, seed_offset=0, threads=8):
        """mode 0: the reference's StratifiedSampler + PCG32 stream; mode 1: the Philox scheme of the HIP kernels."""
        out = np.zeros((cam.height, cam.width, 3), np.float32)
        counts = (C.c_uint64 * 3)()
        lib().orc_render(self._p, C.byref(cam), C.byref(opts), mode, seed_offset, threads, capi.fptr(out), counts)
        return out, {"samples": counts[0], "closest": counts[1], "shadow": counts[2]}

    def camera_ray(self, cam, fx, fy):
        o = np.zeros(3, np.float32); d = np.zeros(3, np.float32)
        lib().orc_camera_ray(C.byref(cam), fx, fy, capi.fptr(o), capi.fptr(d))
        return o, d

    def interaction(self, o, d, tmax=np.inf):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        out = np.zeros(14, np.float32)
        prim = lib().orc_interaction(self._p, capi.fptr(o), capi.fptr(d), tmax, capi.fptr(out))
        return prim, out

    def bsdf_probe(self, o, d, wi, u0, u1):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32); wi = np.ascontiguousarray(wi, np.float32)
        out = np.zeros(11, np.float32)
        prim = lib().orc_bsdf_probe(self._p, capi.fptr(o), capi.fptr(d), capi.fptr(wi), u0, u1, capi.fptr(out))
        return prim, out

    def env_sample(self, env, p, u0, u1):
        p = np.ascontiguousarray(p, np.float32); out = np.zeros(10, np.float32)
        if lib().orc_env_sample(self._p, env, capi.fptr(p), u0, u1, capi.fptr(out)) != 0:
            raise ValueError("no such infinite light")
        return out

    def texture_eval(self, tex, u, v):
        out = np.zeros(3, np.float32)
        lib().orc_texture_eval(self._p, tex, u, v, capi.fptr(out))
        return out

    def close(self):
        if self._p:
            lib().orc_scene_destroy(self._p)
            self._p = None


def tonemap_u8(sum_rgb, spp):
    sum_rgb = np.ascontiguousarray(sum_rgb, np.float32)
    h, w = sum_rgb.shape[:2]
    out = np.zeros((h, w, 3), np.uint8)
    lib().orc_tonemap_u8(capi.fptr(sum_rgb), w, h, spp, capi.u8ptr(out))
    return out


def philox(ctr, k0, k1):
    c = (C.c_uint32 * 4)(*ctr); o = (C.c_uint32 * 4)()
    lib().orc_philox(c, k0, k1, o)
    return list(o)

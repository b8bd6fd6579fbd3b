"""raytracing_amd — Python harness around the MI355X wavefront path-tracing backend.

The product is the pair of in-tree shared libraries:
  lib/librt_amd.so   HIP kernels + the C ABI of include/rt_amd.h
  lib/librt_host.so  C++ scene surface with the reference's class names + the reference's scenes
This package only binds them (ctypes) for tests and benchmarks.
"""
import ctypes as C
import os

import numpy as np

from . import capi
from .capi import Camera, RenderOpts, RenderStats, SceneDesc, TraceStats, BVH_NODE_DTYPE

__all__ = ["HostScene", "Context", "DeviceScene", "bvh_build", "tonemap_u8", "default_assets_dir"]

_rt = None
_host = None


def rt():
    global _rt
    if _rt is None:
        _rt = capi.load_rt_amd()
    return _rt


def host():
    global _host
    if _host is None:
        _host = capi.load_rt_host()
    return _host


def default_assets_dir():
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "assets")


def register_image_file(path, alias=None):
    """Decode an image with PIL and hand the 8-bit RGB pixels to the host library under `alias` (default: its file name)."""
    from PIL import Image
    im = np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8))
    host().rth_register_image((alias or os.path.basename(path)).encode(), im.shape[1], im.shape[0], capi.u8ptr(im))


def decode_image(data):
    """8-bit RGB pixels (H x W x 3) of a PNG or JPEG held in `data` (bytes), decoded by the host library's own readers."""
    w = C.c_int(0); h = C.c_int(0)
    if host().rth_decode_image(data, len(data), C.byref(w), C.byref(h), None) != 0:
        raise ValueError("not a PNG / JPEG this library reads")
    out = np.zeros((h.value, w.value, 3), np.uint8)
    if host().rth_decode_image(data, len(data), C.byref(w), C.byref(h), capi.u8ptr(out)) != 0:
        raise ValueError("decode failed")
    return out


def write_hdr(path, rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    if host().rth_write_hdr(str(path).encode(), rgb.shape[1], rgb.shape[0], capi.fptr(rgb)) != 0:
        raise RuntimeError("cannot write " + str(path))


def read_hdr(path, flip=False):
    w = C.c_int(0); h = C.c_int(0)
    if host().rth_read_hdr(str(path).encode(), int(flip), C.byref(w), C.byref(h), None) != 0:
        raise RuntimeError("cannot read " + str(path))
    out = np.zeros((h.value, w.value, 3), np.float32)
    host().rth_read_hdr(str(path).encode(), int(flip), C.byref(w), C.byref(h), capi.fptr(out))
    return out


def model_probe(path, apply_node_transforms=False):
    """Triangles (n, 3, 3), uvs (n, 3, 2) and the (base colour, metal-roughness) texture names ModelLoad takes from a .gltf / .glb."""
    n = C.c_int(0)
    if host().rth_model_probe(path.encode(), int(apply_node_transforms), C.byref(n), None, None, None, 0) != 0:
        raise RuntimeError("cannot load " + path)
    p = np.zeros((n.value, 3, 3), np.float32); uv = np.zeros((n.value, 3, 2), np.float32)
    names = C.create_string_buffer(1024)
    host().rth_model_probe(path.encode(), int(apply_node_transforms), C.byref(n), capi.fptr(p), capi.fptr(uv), names, 1024)
    return p, uv, tuple(names.value.decode().split("|"))


class HostScene:
    """A scene built by the C++ host surface (raytracing_amd/host/scenes.cpp) and flattened to an rt_scene_desc."""

    def __init__(self, name, param=0, assets_dir=None):
        assets_dir = assets_dir or default_assets_dir()
        # textures (earthmap.jpg, the helmet's JPEG twins of the PNG names in its .gltf) are decoded by the host library itself
        self.name = name
        self._h = host().rth_scene_build(name.encode(), assets_dir.encode(), int(param))
        if not self._h:
            raise RuntimeError("rth_scene_build(%s): %s" % (name, host().rth_last_error().decode()))
        self.desc_ptr = host().rth_scene_desc(self._h)
        self.desc = self.desc_ptr.contents

    def camera(self, width, height):
        cam = Camera()
        if host().rth_scene_camera(self._h, width, height, C.byref(cam)) != 0:
            raise RuntimeError("rth_scene_camera failed")
        return cam

    def texture_source(self, index):
        """File the image texture `index` was decoded from ("" when there is none)."""
        return host().rth_scene_texture_source(self._h, index).decode()

    def env_source(self, index):
        """(.hdr path, scale) the infinite light `index` was constructed with."""
        L = (C.c_float * 3)()
        path = host().rth_scene_env_source(self._h, index, L).decode()
        return path, list(L)

    def nodes(self):
        d = self.desc
        buf = C.string_at(d.bvh_nodes, d.n_nodes * 32)
        return np.frombuffer(buf, dtype=BVH_NODE_DTYPE).copy()

    def prim_order(self):
        d = self.desc
        return np.ctypeslib.as_array(d.prim_order, shape=(d.n_prims,)).copy()

    def tri_p(self):
        d = self.desc
        return np.ctypeslib.as_array(d.tri_p, shape=(d.n_tris, 3, 3)).copy()

    def render_like_main(self, width, height, sx, sy, max_depth=50, png_path=None):
        out = np.zeros((height, width, 3), np.float32)
        st = RenderStats()
        rc = host().rth_render_like_main(self._h, width, height, sx, sy, max_depth, png_path.encode() if png_path else None,
                                         capi.fptr(out), C.byref(st))
        if rc != 0:
            raise RuntimeError("rth_render_like_main: " + host().rth_last_error().decode())
        return out, st

    def close(self):
        if self._h:
            host().rth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bvh_build(bounds, max_prims_in_node=1, split_method=2):
    """split_method: RT_SPLIT_* of rt_amd.h = the reference's enum SplitMethod { SAH, HLBVH, Middle, EqualCounts }."""
    bounds = np.ascontiguousarray(bounds, np.float32).reshape(-1, 6)
    n = bounds.shape[0]
    nodes = np.zeros(max(2 * n, 1), BVH_NODE_DTYPE)
    order = np.zeros(max(n, 1), np.int32)
    nn = C.c_int32(0)
    rc = rt().rt_bvh_build_ex(n, capi.fptr(bounds), max_prims_in_node, split_method, nodes.ctypes.data_as(C.POINTER(capi.BvhNode)), C.byref(nn),
                              capi.iptr(order))
    if rc != 0:
        raise RuntimeError("rt_bvh_build failed")
    return nodes[:nn.value].copy(), order[:n].copy()


def prim_bounds(desc_ptr):
    n = desc_ptr.contents.n_prims
    out = np.zeros((n, 6), np.float32)
    if rt().rt_prim_bounds(desc_ptr, capi.fptr(out)) != 0:
        raise RuntimeError("rt_prim_bounds failed")
    return out


def tonemap_u8(sum_rgb, spp):
    sum_rgb = np.ascontiguousarray(sum_rgb, np.float32)
    h, w = sum_rgb.shape[:2]
    out = np.zeros((h, w, 3), np.uint8)
    if rt().rt_tonemap_u8(capi.fptr(sum_rgb), w, h, spp, capi.u8ptr(out)) != 0:
        raise RuntimeError("rt_tonemap_u8 failed")
    return out


def write_ppm(path, sum_rgb, spp):
    sum_rgb = np.ascontiguousarray(sum_rgb, np.float32)
    h, w = sum_rgb.shape[:2]
    if rt().rt_write_ppm(str(path).encode(), capi.fptr(sum_rgb), w, h, spp) != 0:
        raise RuntimeError("rt_write_ppm failed")


def make_opts(sx, sy, begin=0, end=None, max_depth=50, seed=(0x243F6A88, 0x85A308D3), pool_paths=0, flags=0, rr_threshold=1.0):
    o = RenderOpts()
    o.strata_x, o.strata_y = sx, sy
    o.sample_begin, o.sample_end = begin, (sx * sy if end is None else end)
    o.max_depth = max_depth
    o.seed_lo, o.seed_hi = seed
    o.pool_paths = pool_paths
    o.flags = flags
    o.rr_threshold = rr_threshold
    return o


class Context:
    """rt_ctx: one per process, bound to one HIP device."""

    def __init__(self, device=0):
        self._p = C.c_void_p()
        rc = rt().rt_init(device, C.byref(self._p))
        if rc != 0:
            raise RuntimeError("rt_init failed: %s" % rt().rt_last_error(None).decode())

    def error(self):
        return rt().rt_last_error(self._p).decode()

    def close(self):
        if self._p:
            rt().rt_shutdown(self._p)
            self._p = C.c_void_p()


class DeviceScene:
    """rt_scene: a flattened scene resident in HBM."""

    def __init__(self, ctx, desc_ptr):
        self.ctx = ctx
        self._p = C.c_void_p()
        if rt().rt_scene_create(ctx._p, desc_ptr, C.byref(self._p)) != 0:
            raise RuntimeError("rt_scene_create: " + ctx.error())

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s: %s" % (what, self.ctx.error()))

    def get_bvh(self, n_prims):
        nn = C.c_int32(0)
        self._chk(rt().rt_scene_get_bvh(self._p, None, C.byref(nn), None), "rt_scene_get_bvh")
        nodes = np.zeros(nn.value, BVH_NODE_DTYPE)
        order = np.zeros(n_prims, np.int32)
        self._chk(rt().rt_scene_get_bvh(self._p, nodes.ctypes.data_as(C.POINTER(capi.BvhNode)), C.byref(nn), capi.iptr(order)),
                  "rt_scene_get_bvh")
        return nodes, order

    def trace_closest(self, o, d, tmax):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        n = o.shape[0]
        prim = np.zeros(n, np.int32); t = np.zeros(n, np.float32); b = np.zeros((n, 2), np.float32)
        self._chk(rt().rt_trace_closest(self._p, n, capi.fptr(o), capi.fptr(d), capi.fptr(tmax), capi.iptr(prim), capi.fptr(t),
                                        capi.fptr(b)), "rt_trace_closest")
        return prim, t, b

    def trace_any(self, o, d, tmax):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        n = o.shape[0]
        hit = np.zeros(n, np.uint8)
        self._chk(rt().rt_trace_any(self._p, n, capi.fptr(o), capi.fptr(d), capi.fptr(tmax), capi.u8ptr(hit)), "rt_trace_any")
        return hit

    def trace_closest_device(self, n, rays_ptr, hits_ptr, repeats=1, count_work=False):
        st = TraceStats()
        self._chk(rt().rt_trace_closest_device(self._p, n, C.c_void_p(rays_ptr), C.c_void_p(hits_ptr), repeats, int(count_work),
                                               C.byref(st)), "rt_trace_closest_device")
        return st

    def trace_closest_device_noalloc(self, n, rays_ptr, hits_ptr, count_work=False, want_stats=True):
        st = TraceStats()
        self._chk(rt().rt_trace_closest_device_noalloc(self._p, n, C.c_void_p(rays_ptr), C.c_void_p(hits_ptr), int(count_work),
                                                       C.byref(st) if want_stats else None), "rt_trace_closest_device_noalloc")
        return st

    def render(self, cam, opts):
        out = np.zeros((cam.height, cam.width, 3), np.float32)
        st = RenderStats()
        self._chk(rt().rt_render(self._p, C.byref(cam), C.byref(opts), capi.fptr(out), C.byref(st)), "rt_render")
        return out, st

    def render_device(self, cam, opts):
        ptr = C.c_void_p()
        st = RenderStats()
        self._chk(rt().rt_render_device(self._p, C.byref(cam), C.byref(opts), C.byref(ptr), C.byref(st)), "rt_render_device")
        return ptr.value, st

    def close(self):
        if self._p:
            rt().rt_scene_destroy(self._p)
            self._p = C.c_void_p()


class MultiScene:
    """rt_multi: one context and scene copy per listed device (a device may appear twice), samples of every pixel shared out."""

    def __init__(self, devices, desc_ptr):
        self.devices = list(devices)
        arr = (C.c_int32 * len(self.devices))(*self.devices)
        self._p = C.c_void_p()
        if rt().rt_multi_create(arr, len(self.devices), desc_ptr, C.byref(self._p)) != 0:
            raise RuntimeError(rt().rt_multi_last_error(None).decode())

    def render(self, cam, opts):
        out = np.zeros((cam.height, cam.width, 3), np.float32)
        per = (RenderStats * len(self.devices))(); tot = RenderStats()
        if rt().rt_multi_render(self._p, C.byref(cam), C.byref(opts), capi.fptr(out), per, C.byref(tot)) != 0:
            raise RuntimeError(rt().rt_multi_last_error(self._p).decode())
        return out, list(per), tot

    def close(self):
        if self._p:
            rt().rt_multi_destroy(self._p)
            self._p = C.c_void_p()


def split_samples(total, rank, world):
    b = C.c_int32(); e = C.c_int32()
    if rt().rt_split_samples(total, rank, world, C.byref(b), C.byref(e)) != 0:
        raise ValueError("bad rank/world")
    return b.value, e.value

"""Files the reference driver (oracle/ref_driver.cpp, built from the reference's own sources into oracle/_ref/) reads: a flat
dump of an rt_scene_desc and a camera record.  Test infrastructure, used by tests/golden/make_golden.py and by the
cpu_baseline leg of bench.py."""
import os
import struct

import numpy as np

CAMERAS = {   # eye, look, up of the reference's scene functions (RayTracing.cpp:187-190, 406, 315); fov 40, near 0.01 (:424-426)
    "cornell": ((2.5, 2.5, 11.8), (2.5, 2.5, 0.0), (0, 1, 0)),
    "cornell_plastic": ((2.5, 2.5, 11.8), (2.5, 2.5, 0.0), (0, 1, 0)),
    "sphere1": ((13, 2, 3), (0, 0, 0), (0, 1, 0)),
    "spheres2": ((13, 2, 3), (0, 0, 0), (0, 1, 0)),
    "spheres3": ((13, 2, 3), (0, 0, 0), (0, 1, 0)),
    "cornell_metal": ((2.5, 2.5, 11.8), (2.5, 2.5, 0.0), (0, 1, 0)),
    "cornell_roughglass": ((2.5, 2.5, 11.8), (2.5, 2.5, 0.0), (0, 1, 0)),
    "cornell_env": ((2.5, 2.5, 11.8), (2.5, 2.5, 0.0), (0, 1, 0)),
    "spheres_env": ((13, 2, 3), (0, 0, 0), (0, 1, 0)),
    "helmet": ((3.5, 3.2, 1.5), (2.0, 2.3, 2.8), (0, 1, 0)),
    "mrspheres": ((0, 0, 50), (0, 0, 0), (0, 1, 0)),      # MetalRoughSphere(), RayTracing.cpp:214
}


def as_np(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def texture_png(scene_name, tex, idx, tmp):
    """Write the texture's 8-bit source as PNG for the reference's PNG-only reader (imageio.cpp:7-17)."""
    from PIL import Image
    w, h = tex.width, tex.height
    tx = np.ctypeslib.as_array(tex.texels, shape=(h, w, 3))
    u8 = np.rint(tx * 255.0).astype(np.uint8)
    assert np.array_equal((u8 / np.float32(255.0)).astype(np.float32), tx), "texture is not 8-bit data"
    path = os.path.join(tmp, "%s_tex%d.png" % (scene_name, idx))
    Image.fromarray(u8, "RGB").save(path)
    return path


def dump_scene(scene, path, tmp):
    d = scene.desc
    out = [struct.pack("<I6i", 0x43535452, d.n_prims, d.n_tris, d.n_spheres, d.n_materials, d.n_textures, d.n_lights)]
    out.append(as_np(d.prim_kind, d.n_prims, np.uint8).tobytes())
    out.append(as_np(d.prim_geom, d.n_prims, np.int32).tobytes())
    out.append(as_np(d.prim_material, d.n_prims, np.int32).tobytes())
    out.append(as_np(d.prim_light, d.n_prims, np.int32).tobytes())
    out.append(as_np(d.tri_p, 9 * d.n_tris, np.float32).tobytes())
    out.append(as_np(d.tri_uv, 6 * d.n_tris, np.float32).tobytes())
    has_n = bool(d.tri_n) and bool(d.tri_has_n)
    out.append(struct.pack("<i", int(has_n)))
    if has_n:
        out.append(as_np(d.tri_n, 9 * d.n_tris, np.float32).tobytes())
        out.append(as_np(d.tri_has_n, d.n_tris, np.uint8).tobytes())
    for i in range(d.n_spheres):
        s = d.spheres[i]
        out.append(struct.pack("<33f", s.radius, *list(s.o2w), *list(s.w2o)))
    for i in range(d.n_materials):
        m = d.materials[i]
        out.append(struct.pack("<7i", m.kind, *list(m.tex), m.remap_roughness, 0))
    for i in range(d.n_textures):
        t = d.textures[i]
        png = b""
        if t.kind == 1:
            # a PNG source goes to the reference as it is (the reference then does its own resampling of images that are not
            # powers of two, mipmap.h:70-121); anything else as a PNG of the texels, which must then be unresampled 8-bit data
            src = scene.texture_source(i) if hasattr(scene, "texture_source") else ""
            png = src.encode() if src.lower().endswith(".png") and os.path.exists(src) else texture_png(scene.name, t, i, tmp).encode()
        out.append(struct.pack("<3i3f4fi", t.kind, t.width, t.height, *list(t.value), t.su, t.sv, t.du, t.dv, len(png)) + png)
    for i in range(d.n_lights):
        L = d.lights[i]
        out.append(struct.pack("<i3fif3i", L.prim, *list(L.Le), L.two_sided, L.area, L.flip_normal, L.kind, L.env))
    # infinite lights: what the reference's constructor takes (transform, scale, .hdr file); it builds its tables itself
    out.append(struct.pack("<i", d.n_env_lights))
    for i in range(d.n_env_lights):
        e = d.env_lights[i]
        hdr, scale = scene.env_source(i)
        out.append(struct.pack("<35f", *list(e.light_to_world), *list(e.world_to_light), *scale) + struct.pack("<i", len(hdr.encode())) + hdr.encode())
    with open(path, "wb") as f:
        f.write(b"".join(out))


def cam_spec(name, w, h):
    eye, look, up = CAMERAS[name]
    return struct.pack("<11f2i", *eye, *look, *up, 40.0, 0.01, w, h)




def same_tree(g, nodes, order):
    """Is (nodes, order) the tree of the golden file g (ref_bvh_<scene>.npz)?  Large trees are stored as SHA-256 digests of the
    reference's node and leaf-order bytes instead of the bytes themselves."""
    import hashlib
    if "nodes_sha256" in g.files:
        return (len(nodes) == int(g["n_nodes"]) and hashlib.sha256(nodes.tobytes()).hexdigest() == str(g["nodes_sha256"]) and
                hashlib.sha256(np.ascontiguousarray(order, np.int32).tobytes()).hexdigest() == str(g["order_sha256"]))
    return len(nodes) == int(g["n_nodes"]) and nodes.tobytes() == g["nodes"].tobytes() and np.array_equal(order, g["order"])

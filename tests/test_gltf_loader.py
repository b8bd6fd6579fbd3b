"""glTF reader of the host surface: the binary container (.glb), embedded images, base64 buffers and node transforms, checked
against the split .gltf + .bin + .jpg files of the same model (the only form the reference's loader is used with)."""
import base64
import json
import os
import shutil
import struct

import numpy as np
import pytest

import raytracing_amd as ra

ASSETS = ra.default_assets_dir()
SPLIT = os.path.join(ASSETS, "DamagedHelmet", "glTF")


def _pad4(b, fill):
    return b + fill * ((4 - len(b) % 4) % 4)


def make_glb(dst):
    """Pack DamagedHelmet.gltf + .bin + the two JPEG textures the renderer uses into one .glb (images in buffer views)."""
    g = json.load(open(os.path.join(SPLIT, "DamagedHelmet.gltf")))
    blob = bytearray(open(os.path.join(SPLIT, g["buffers"][0]["uri"]), "rb").read())
    used = {g["textures"][g["materials"][0]["pbrMetallicRoughness"][k]["index"]]["source"] for k in ("baseColorTexture", "metallicRoughnessTexture")}
    for i, im in enumerate(g["images"]):
        if i not in used:
            im.pop("uri"); im["bufferView"] = 0; im["mimeType"] = "image/png"     # never read by the loader
            continue
        data = open(os.path.join(SPLIT, im["uri"].replace(".png", ".jpg")), "rb").read()
        while len(blob) % 4:
            blob.append(0)
        g["bufferViews"].append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)})
        blob += data
        im.pop("uri"); im["bufferView"] = len(g["bufferViews"]) - 1; im["mimeType"] = "image/jpeg"
    g["buffers"] = [{"byteLength": len(blob)}]
    js = _pad4(json.dumps(g).encode(), b" "); bn = _pad4(bytes(blob), b"\0")
    with open(dst, "wb") as f:
        f.write(struct.pack("<III", 0x46546C67, 2, 12 + 8 + len(js) + 8 + len(bn)))
        f.write(struct.pack("<II", len(js), 0x4E4F534A)); f.write(js)
        f.write(struct.pack("<II", len(bn), 0x004E4942)); f.write(bn)


@pytest.fixture(scope="module")
def glb_assets(tmp_path_factory):
    root = tmp_path_factory.mktemp("assets")
    os.makedirs(root / "DamagedHelmet" / "glTF-Binary")
    make_glb(str(root / "DamagedHelmet" / "glTF-Binary" / "DamagedHelmet.glb"))
    return str(root)


def test_glb_gives_the_same_triangles(glb_assets):
    p0, uv0, names0 = ra.model_probe(os.path.join(SPLIT, "DamagedHelmet.gltf"))
    p1, uv1, names1 = ra.model_probe(os.path.join(glb_assets, "DamagedHelmet", "glTF-Binary", "DamagedHelmet.glb"))
    assert p0.shape == (15452, 3, 3)
    assert np.array_equal(p0, p1) and np.array_equal(uv0, uv1)
    assert names0 == ("Default_albedo.png", "Default_metalRoughness.png")
    assert names1 == ("DamagedHelmet.glb#image0", "DamagedHelmet.glb#image1")


def test_glb_scene_is_the_gltf_scene(glb_assets):
    """Flattened scene (geometry, BVH, materials, texels decoded from the embedded JPEGs) byte for byte."""
    a = ra.HostScene("helmet"); b = ra.HostScene("helmet_glb", assets_dir=glb_assets)
    da, db = a.desc, b.desc
    assert (da.n_prims, da.n_tris, da.n_nodes, da.n_textures, da.n_materials) == (db.n_prims, db.n_tris, db.n_nodes, db.n_textures, db.n_materials)
    assert np.array_equal(a.tri_p(), b.tri_p())
    assert a.nodes().tobytes() == b.nodes().tobytes() and np.array_equal(a.prim_order(), b.prim_order())
    for i in range(da.n_textures):
        ta, tb = da.textures[i], db.textures[i]
        assert (ta.kind, ta.width, ta.height) == (tb.kind, tb.width, tb.height)
        if ta.kind == 1:
            n = ta.width * ta.height * 3
            assert np.array_equal(np.ctypeslib.as_array(ta.texels, shape=(n,)), np.ctypeslib.as_array(tb.texels, shape=(n,)))
    a.close(); b.close()


def test_node_transforms_are_an_option():
    """The helmet's node carries a 90 degree rotation about x.  The reference's loader never applies it (modelload.cpp:18-25) and
    neither does the default here; with applyNodeTransforms the rotation is applied to the raw positions before the z flip."""
    path = os.path.join(SPLIT, "DamagedHelmet.gltf")
    p0, _, _ = ra.model_probe(path)
    p1, _, _ = ra.model_probe(path, apply_node_transforms=True)
    raw = p0 * np.array([1, 1, -1], np.float32)                   # undo the z flip: file coordinates
    q = json.load(open(path))["nodes"][0]["rotation"]             # (x, y, z, w)
    x, y, z, w = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    want = (raw.astype(np.float64) @ R.T) * np.array([1, 1, -1])
    assert np.abs(p1 - want).max() < 1e-5
    assert np.abs(p1 - p0).max() > 0.5


def test_base64_buffers_and_matrix_nodes(tmp_path):
    """A two-triangle model written here: base64 data URI buffer, a parent node with `matrix` and a child with translation + scale."""
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], np.float32); idx = np.array([0, 1, 2, 2, 1, 3], np.uint16)
    blob = idx.tobytes() + pos.tobytes()
    M = np.eye(4); M[:3, 3] = [10, 0, 0]                       # parent: translate x by 10
    g = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}],
         "nodes": [{"matrix": M.T.reshape(-1).tolist(), "children": [1]}, {"mesh": 0, "translation": [0, 2, 0], "scale": [3, 1, 1]}],
         "meshes": [{"primitives": [{"attributes": {"POSITION": 1}, "indices": 0}]}],
         "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}],
         "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 12}, {"buffer": 0, "byteOffset": 12, "byteLength": 48}],
         "accessors": [{"bufferView": 0, "componentType": 5123, "count": 6, "type": "SCALAR"},
                       {"bufferView": 1, "componentType": 5126, "count": 4, "type": "VEC3"}]}
    path = str(tmp_path / "quad.gltf"); json.dump(g, open(path, "w"))
    p0, _, _ = ra.model_probe(path)
    assert np.array_equal(p0.reshape(6, 3), pos[idx] * np.array([1, 1, -1], np.float32))
    p1, _, _ = ra.model_probe(path, apply_node_transforms=True)
    want = (pos[idx] * np.array([3, 1, 1]) + np.array([10, 2, 0])) * np.array([1, 1, -1])
    assert np.allclose(p1.reshape(6, 3), want)


def test_metal_rough_spheres_file_with_every_mesh_and_both_textures():
    """MetalRoughSpheres.gltf as the reference ships it (five meshes under one parent node with a matrix, one material, 16-bit indices
    into three shared buffer views) over a synthetic geometry buffer of the stated layout (the reference lacks the .bin): ModelLoad
    walks the children in file order (5, 4, 3, 2, 1), takes every mesh whole, ignores the node matrix unless told otherwise, and hands
    back the material's two texture names; with node transforms applied the parent's 0.4 scale and axis swap show in the bounds."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import mrspheres_bin
    mrspheres_bin.ensure(ASSETS)
    path = os.path.join(ASSETS, "MetalRoughSpheres", "glTF", "MetalRoughSpheres.gltf")
    g = json.load(open(path))
    counts = [g["accessors"][m["primitives"][0]["indices"]]["count"] // 3 for m in g["meshes"]]
    p, uv, names = ra.model_probe(path)
    assert len(p) == sum(counts) == 501776
    assert names == ("Spheres_BaseColor.png", "Spheres_MetalRough.png")
    # node 0 lists its children as 5, 4, 3, 2, 1 = meshes 4, 3, 2, 1, 0: the triangles arrive in that order
    order = [4, 3, 2, 1, 0]
    start = 0
    for mi in order:
        acc = g["accessors"][g["meshes"][mi]["primitives"][0]["attributes"]["POSITION"]]
        blk = p[start:start + counts[mi]].reshape(-1, 3)
        lo, hi = np.array(acc["min"]), np.array(acc["max"])
        # (the loader negates z: aiProcess_MakeLeftHanded, modelload.cpp:9)
        assert blk[:, 0].min() >= lo[0] - 1e-4 and blk[:, 0].max() <= hi[0] + 1e-4 and (-blk[:, 2]).min() >= lo[2] - 1e-4 and (-blk[:, 2]).max() <= hi[2] + 1e-4
        start += counts[mi]
    assert uv.min() >= -1e-6 and uv.max() <= 1 + 1e-6
    q, _, _ = ra.model_probe(path, apply_node_transforms=True)
    assert len(q) == len(p) and np.abs(q).max() < 0.41 * np.abs(p).max() + 1e-3 and np.abs(q).max() > 0.39 * np.abs(p).max()

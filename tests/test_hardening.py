"""Malformed inputs are refused on the host, with a reason, before anything follows them: scene descriptions with indices out of
range or trees the kernels cannot walk, glTF files whose accessors or indices point outside their buffers, JPEG streams with
corrupt tables.  None of this needs a GPU."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

import raytracing_amd as ra
from raytracing_amd import capi


def _check(desc):
    msg = C.create_string_buffer(256)
    rc = ra.rt().rt_scene_desc_check(C.pointer(desc), msg, 256)
    return rc, msg.value.decode()


def _copy(desc):
    return type(desc).from_buffer_copy(desc)


def test_consistent_descriptions_pass(host_scene):
    for name in ("cornell", "sphere1", "helmet"):
        rc, msg = _check(host_scene(name).desc)
        assert rc == 0 and msg == "", (name, msg)


@pytest.mark.parametrize("with_tree", [True, False])
def test_bad_primitive_indices_are_refused(with_tree, host_scene):
    """Also with bvh_nodes == NULL, the case in which the library computes primitive bounds and builds the tree itself."""
    hs = host_scene("cornell")
    for field, bad in (("prim_material", 99), ("prim_geom", 10 ** 6), ("prim_geom", -1), ("prim_light", 77), ("prim_light", -2)):
        d = _copy(hs.desc)
        if not with_tree:
            d.bvh_nodes = None; d.prim_order = None; d.n_nodes = 0
        n = d.n_prims
        arr = (C.c_int32 * n)(*[getattr(hs.desc, field)[i] for i in range(n)])
        arr[3] = bad
        setattr(d, field, C.cast(arr, C.POINTER(C.c_int32)))
        rc, msg = _check(d)
        assert rc < 0 and "out of range" in msg, (field, bad, msg)
        if field == "prim_geom":
            out = np.zeros((n, 6), np.float32)
            assert ra.rt().rt_prim_bounds(C.pointer(d), capi.fptr(out)) < 0      # does not read through the bad index either
    d = _copy(hs.desc)
    kinds = (C.c_uint8 * d.n_prims)(*[hs.desc.prim_kind[i] for i in range(d.n_prims)])
    kinds[0] = 7
    d.prim_kind = C.cast(kinds, C.POINTER(C.c_uint8))
    assert _check(d)[0] < 0


def test_missing_arrays_and_bad_tables_are_refused(host_scene):
    hs = host_scene("cornell")
    for field in ("prim_kind", "prim_geom", "prim_material", "prim_light", "tri_p", "tri_uv", "materials", "textures", "lights"):
        d = _copy(hs.desc)
        setattr(d, field, None)
        rc, msg = _check(d)
        assert rc < 0 and "NULL" in msg, (field, msg)
    d = _copy(hs.desc)
    mats = (capi.Material * d.n_materials)(*[hs.desc.materials[i] for i in range(d.n_materials)])
    mats[0].tex[0] = d.n_textures
    d.materials = C.cast(mats, C.POINTER(capi.Material))
    assert "texture slot" in _check(d)[1]
    mats[0].tex[0] = 0; mats[0].kind = 17
    assert "material kind" in _check(d)[1]
    d = _copy(hs.desc)
    lights = (capi.Light * d.n_lights)(*[hs.desc.lights[i] for i in range(d.n_lights)])
    lights[0].prim = 5                                  # a wall triangle that does not point back at light 0
    d.lights = C.cast(lights, C.POINTER(capi.Light))
    assert "light" in _check(d)[1]
    d = _copy(hs.desc); d.n_prims = 0
    assert "empty" in _check(d)[1]


def _with_nodes(hs, nodes):
    d = _copy(hs.desc)
    d.n_nodes = len(nodes)
    d.bvh_nodes = nodes.ctypes.data_as(C.POINTER(capi.BvhNode))
    return d


def test_unwalkable_trees_are_refused(host_scene):
    hs = host_scene("cornell")
    base = hs.nodes()
    n = base.copy(); n[0]["offset"] = len(n) + 5
    assert "links" in _check(_with_nodes(hs, n))[1]
    n = base.copy(); n[0]["offset"] = 1                  # second child == first child
    assert "links" in _check(_with_nodes(hs, n))[1]
    n = base.copy(); n[0]["axis"] = 3
    assert "links" in _check(_with_nodes(hs, n))[1]
    leaf = int(np.nonzero(base["nprim"] > 0)[0][0])
    n = base.copy(); n[leaf]["offset"] = hs.desc.n_prims
    assert "leaf range" in _check(_with_nodes(hs, n))[1]
    other = int(np.nonzero(base["nprim"] > 0)[0][1])
    n = base.copy(); n[leaf]["offset"] = n[other]["offset"]
    assert "share a slot" in _check(_with_nodes(hs, n))[1]
    # two interior nodes naming the same second child: that child is reached twice, its old sibling never
    inner = [i for i in range(len(base)) if base[i]["nprim"] == 0 and i + 1 < len(base) and base[i + 1]["nprim"] == 0]
    n = base.copy(); n[inner[0]]["offset"] = n[inner[0] + 1]["offset"]
    msg = _check(_with_nodes(hs, n))[1]
    assert "twice" in msg or "reachable" in msg, msg
    d = _with_nodes(hs, np.concatenate([base, base]))    # more nodes than 2 * n_prims - 1
    assert "more nodes" in _check(d)[1]
    d = _copy(hs.desc)
    order = (C.c_int32 * d.n_prims)(*[hs.desc.prim_order[i] for i in range(d.n_prims)])
    order[0] = order[1]
    d.prim_order = C.cast(order, C.POINTER(C.c_int32))
    assert "permutation" in _check(d)[1]


def test_tree_deeper_than_the_traversal_stack_is_refused():
    """A degenerate chain: every interior node has a leaf as first child and the rest of the chain as second.  70 interior levels
    exceed the 64 entries a traversal stack holds (the reference's nodesToVisit[64], bvh.cpp:253); 60 are fine."""
    for levels, ok in ((60, True), (70, False)):
        n_prims = levels + 1
        nodes = np.zeros(2 * n_prims - 1, capi.BVH_NODE_DTYPE)
        for k in range(levels):
            i = 2 * k
            nodes[i]["nprim"] = 0; nodes[i]["offset"] = i + 2; nodes[i]["axis"] = 0
            nodes[i + 1]["nprim"] = 1; nodes[i + 1]["offset"] = k
        nodes[2 * levels]["nprim"] = 1; nodes[2 * levels]["offset"] = levels
        nodes["bmin"] = 0; nodes["bmax"] = 1
        tri = np.tile(np.array([0, 0, 0, 1, 0, 0, 0, 1, 0], np.float32), n_prims)
        uv = np.tile(np.array([0, 0, 1, 0, 1, 1], np.float32), n_prims)
        kind = np.zeros(n_prims, np.uint8); geom = np.arange(n_prims, dtype=np.int32); mat = np.zeros(n_prims, np.int32)
        light = np.full(n_prims, -1, np.int32); order = np.arange(n_prims, dtype=np.int32)
        tex = capi.Texture(); tex.kind = 0
        m = capi.Material(); m.kind = 0; m.tex[0] = 0; m.tex[1] = m.tex[2] = m.tex[3] = -1
        d = capi.SceneDesc()
        d.n_prims = n_prims; d.prim_kind = capi.u8ptr(kind); d.prim_geom = capi.iptr(geom); d.prim_material = capi.iptr(mat); d.prim_light = capi.iptr(light)
        d.n_tris = n_prims; d.tri_p = capi.fptr(tri); d.tri_uv = capi.fptr(uv)
        d.n_materials = 1; d.materials = C.pointer(m); d.n_textures = 1; d.textures = C.pointer(tex)
        d.n_nodes = len(nodes); d.bvh_nodes = nodes.ctypes.data_as(C.POINTER(capi.BvhNode)); d.prim_order = capi.iptr(order)
        rc, msg = _check(d)
        assert (rc == 0) == ok, (levels, msg)
        if not ok:
            assert "deeper" in msg


# ------------------------------------------------------------------ glTF --
def _write_gltf(tmp_path, name, n_verts=4, indices=(0, 1, 2, 0, 2, 3), view_override=None, acc_override=None):
    pos = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)[:n_verts]
    idx = np.array(indices, np.uint16)
    blob = idx.tobytes() + b"\0" * (-len(idx.tobytes()) % 4)
    ioff = 0; poff = len(blob)
    blob += pos.tobytes()
    views = [{"buffer": 0, "byteOffset": ioff, "byteLength": idx.nbytes}, {"buffer": 0, "byteOffset": poff, "byteLength": pos.nbytes}]
    accs = [{"bufferView": 0, "componentType": 5123, "count": len(idx), "type": "SCALAR"},
            {"bufferView": 1, "componentType": 5126, "count": n_verts, "type": "VEC3"}]
    if view_override:
        views[view_override[0]].update(view_override[1])
    if acc_override:
        accs[acc_override[0]].update(acc_override[1])
    (tmp_path / (name + ".bin")).write_bytes(blob)
    doc = {"asset": {"version": "2.0"}, "buffers": [{"uri": name + ".bin", "byteLength": len(blob)}], "bufferViews": views, "accessors": accs,
           "meshes": [{"primitives": [{"attributes": {"POSITION": 1}, "indices": 0}]}], "nodes": [{"mesh": 0}], "scenes": [{"nodes": [0]}], "scene": 0}
    p = tmp_path / (name + ".gltf")
    p.write_text(json.dumps(doc))
    return str(p)


def test_gltf_loader_refuses_what_points_outside_its_buffers(tmp_path):
    p, uv, _ = ra.model_probe(_write_gltf(tmp_path, "ok"))
    assert p.shape == (2, 3, 3)
    bad = [
        dict(indices=(0, 1, 2, 0, 2, 9)),                                     # an index beyond the vertices
        dict(acc_override=(0, {"bufferView": 7})),                            # bufferView that does not exist
        dict(view_override=(1, {"buffer": 3})),                               # buffer that does not exist
        dict(acc_override=(1, {"count": 4000})),                              # runs past the end of the buffer
        dict(acc_override=(1, {"count": -5})),                                # negative count
        dict(acc_override=(1, {"byteOffset": 40})),                           # the LAST element crosses the end of its view
        dict(view_override=(1, {"byteOffset": -8})),
    ]
    for k, kw in enumerate(bad):
        with pytest.raises(RuntimeError):
            ra.model_probe(_write_gltf(tmp_path, "bad%d" % k, **kw))
    # seven index values: the last one, which completes no triangle, is dropped
    p, _, _ = ra.model_probe(_write_gltf(tmp_path, "odd", indices=(0, 1, 2, 0, 2, 3, 1)))
    assert p.shape == (2, 3, 3)


# ------------------------------------------------------------------ JPEG --
def test_jpeg_with_corrupt_tables_fails_cleanly():
    data = bytearray(open(os.path.join(ra.default_assets_dir(), "earthmap.jpg"), "rb").read())
    assert ra.decode_image(bytes(data)).shape == (512, 1024, 3)
    # every DC Huffman table: all symbols become category 200 (no 8-bit image has a DC difference beyond category 11)
    pos = 2; patched = 0
    while pos + 4 <= len(data) and data[pos] == 0xFF:
        marker = data[pos + 1]; seg_len = struct.unpack(">H", data[pos + 2:pos + 4])[0]
        if marker == 0xC4:
            q = pos + 4
            while q < pos + 2 + seg_len:
                tc = data[q] >> 4; counts = data[q + 1:q + 17]; nvals = sum(counts)
                if tc == 0:
                    for k in range(nvals):
                        data[q + 17 + k] = 200
                    patched += 1
                q += 17 + nvals
        if marker == 0xDA:
            break
        pos += 2 + seg_len
    assert patched >= 1
    with pytest.raises(ValueError):
        ra.decode_image(bytes(data))
    # a frame header that claims 65535 x 65535 pixels is refused instead of sizing buffers after it
    data = bytearray(open(os.path.join(ra.default_assets_dir(), "earthmap.jpg"), "rb").read())
    pos = 2
    while data[pos + 1] not in (0xC0, 0xC1, 0xC2):
        pos += 2 + struct.unpack(">H", data[pos + 2:pos + 4])[0]
    data[pos + 5:pos + 9] = b"\xff\xff\xff\xff"
    with pytest.raises(ValueError):
        ra.decode_image(bytes(data))


def test_infinite_light_descriptions_are_checked(host_scene):
    hs = host_scene("cornell_env")
    assert _check(hs.desc)[0] == 0
    d = _copy(hs.desc)
    lights = (capi.Light * d.n_lights)(*[hs.desc.lights[i] for i in range(d.n_lights)])
    lights[2].env = 3
    d.lights = C.cast(lights, C.POINTER(capi.Light))
    assert "infinite light" in _check(d)[1]
    lights[2].env = 0; lights[2].kind = 9
    assert "light kind" in _check(d)[1]
    d = _copy(hs.desc)
    envs = (capi.EnvLight * 1)(hs.desc.env_lights[0])
    envs[0].cdf = None
    d.env_lights = C.cast(envs, C.POINTER(capi.EnvLight))
    assert "NULL table" in _check(d)[1]
    envs[0].cdf = hs.desc.env_lights[0].cdf; envs[0].nu = 0
    assert "empty" in _check(d)[1]
    d = _copy(hs.desc); d.env_lights = None
    assert "env_lights" in _check(d)[1]


def test_hdr_reader_flat_and_run_length_encoded(tmp_path):
    """Radiance .hdr as stb_image's loader reads it: value = mantissa byte * 2^(exponent - 136), rows flipped on request; a
    run-length encoded file (runs and literal spans per channel) decodes to the same floats as the flat one."""
    rng = np.random.default_rng(4)
    h, w = 9, 40
    img = (rng.random((h, w, 3)) ** 4 * 50).astype(np.float32); img[2, 5:30] = [1.5, 0.25, 0.0]; img[3] = 0
    flat = tmp_path / "flat.hdr"
    ra.write_hdr(flat, img)
    got = ra.read_hdr(flat)
    raw = open(flat, "rb").read()
    rgbe = np.frombuffer(raw[raw.index(b"+X %d\n" % w) + len(b"+X %d\n" % w):], np.uint8).reshape(h, w, 4)
    want = rgbe[:, :, :3].astype(np.float32) * np.where(rgbe[:, :, 3:] > 0, np.ldexp(np.float32(1), rgbe[:, :, 3:].astype(int) - 136), 0).astype(np.float32)
    assert np.array_equal(got, want) and np.abs(got - img).max() <= img.max() / 128
    assert np.array_equal(ra.read_hdr(flat, flip=True), want[::-1])
    # the same pixels, run-length encoded the way Radiance writes them
    out = bytearray(raw[:raw.index(b"+X %d\n" % w) + len(b"+X %d\n" % w)])
    for y in range(h):
        out += bytes([2, 2, w >> 8, w & 255])
        for c in range(4):
            row = rgbe[y, :, c].tolist(); i = 0
            while i < w:
                run = 1
                while i + run < w and run < 127 and row[i + run] == row[i]:
                    run += 1
                if run >= 3:
                    out += bytes([128 + run, row[i]]); i += run
                else:
                    j = i
                    while j < w and j - i < 128 and not (j + 2 < w and row[j] == row[j + 1] == row[j + 2]):
                        j += 1
                    j = max(j, i + 1)
                    out += bytes([j - i]) + bytes(row[i:j]); i = j
    rle = tmp_path / "rle.hdr"
    rle.write_bytes(bytes(out))
    assert np.array_equal(ra.read_hdr(rle), want)
    # the committed run-length encoded file (tests/golden/make_env_map.py wrote it with this encoder from these pixels) decodes to the same floats
    fixture = os.path.join(os.path.dirname(__file__), "golden", "assets", "env_rle_40x9.hdr")
    assert open(fixture, "rb").read() == bytes(out)
    assert np.array_equal(ra.read_hdr(fixture), want)
    # stb_image's rules: a scanline whose length word is not the width is an error; so is a scanline without the 2 2 header after the
    # first one; a FIRST scanline without it means the whole file is flat
    hdr_len = raw.index(b"+X %d\n" % w) + len(b"+X %d\n" % w)
    bad_len = bytearray(out); bad_len[hdr_len + 3] = (w - 1) & 255
    (tmp_path / "badlen.hdr").write_bytes(bytes(bad_len))
    with pytest.raises(RuntimeError):
        ra.read_hdr(tmp_path / "badlen.hdr")
    row1 = hdr_len
    for _ in range(1):                                            # find the start of the second scanline by decoding the first
        q = row1 + 4
        for c in range(4):
            i = 0
            while i < w:
                cnt = out[q]; q += 1
                if cnt > 128:
                    q += 1; i += cnt - 128
                else:
                    q += cnt; i += cnt
        row1 = q
    mixed = bytes(out[:row1]) + rgbe[1:].tobytes()                # first scanline encoded, the rest flat
    (tmp_path / "mixed.hdr").write_bytes(mixed)
    with pytest.raises(RuntimeError):
        ra.read_hdr(tmp_path / "mixed.hdr")
    (tmp_path / "bad.hdr").write_bytes(raw[:40])
    with pytest.raises(RuntimeError):
        ra.read_hdr(tmp_path / "bad.hdr")

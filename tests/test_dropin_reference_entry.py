"""Drop-in proof for the scene-building surface: the reference's own entry file RayTracing/RayTracing.cpp is compiled
UNCHANGED against raytracing_amd/host (forwarding headers in host/compat/), its scene functions Cornellbox(), sphere1() and
DamagedHelmet() are executed, and the flattened result must equal the scenes of raytracing_amd/host/scenes.cpp byte for
byte (geometry, BVH, materials, lights, camera).  Runs only where the reference tree is mounted."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

import raytracing_amd as ra
from raytracing_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENTRY = "/root/reference/RayTracing/RayTracing.cpp"
pytestmark = pytest.mark.skipif(not os.path.exists(ENTRY), reason="reference tree not mounted")


@pytest.fixture(scope="module")
def dropin(tmp_path_factory):
    d = tmp_path_factory.mktemp("dropin")
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "raytracing_amd", "host"),
           "-I" + os.path.join(ROOT, "raytracing_amd", "host", "compat")]
    lib = os.path.join(ROOT, "raytracing_amd", "lib")
    obj = str(d / "ref_entry.o")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-include", os.path.join(ROOT, "tests", "dropin", "dropin_prelude.h")]
                          + inc + ["-c", ENTRY, "-o", obj])
    exe = str(d / "run" / "dropin")
    os.makedirs(str(d / "run"))
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off"] + inc + [os.path.join(ROOT, "tests", "dropin", "dropin_main.cpp"), obj,
                           "-L" + lib, "-lrt_host", "-lrt_amd", "-Wl,-rpath," + lib, "-o", exe])
    # DamagedHelmet() opens "..\\DamagedHelmet\\glTF\\DamagedHelmet.gltf" relative to the working directory
    os.symlink(os.path.join(ra.default_assets_dir(), "DamagedHelmet"), str(d / "DamagedHelmet"))
    return exe


def _sections(scene, w=800, h=800):
    d = scene.desc
    def arr(ptr, n, dt):
        return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt).tobytes() if n else b""
    out = {"hdr": struct.pack("<4i", d.n_prims, d.n_tris, d.n_spheres, d.n_nodes),
           "nodes": C.string_at(d.bvh_nodes, 32 * d.n_nodes), "order": arr(d.prim_order, d.n_prims, np.int32),
           "tri_p": arr(d.tri_p, 9 * d.n_tris, np.float32), "tri_uv": arr(d.tri_uv, 6 * d.n_tris, np.float32),
           "spheres": C.string_at(d.spheres, 132 * d.n_spheres) if d.n_spheres else b"",
           "prim_material": arr(d.prim_material, d.n_prims, np.int32), "materials": C.string_at(d.materials, 28 * d.n_materials)}
    tex = b""
    for i in range(d.n_textures):
        t = d.textures[i]
        tex += struct.pack("<i3f2i", t.kind, *list(t.value), t.width, t.height)
    out["textures"] = tex
    out["lights"] = C.string_at(d.lights, C.sizeof(capi.Light) * d.n_lights) if d.n_lights else b""
    out["camera"] = bytes(scene.camera(w, h))
    return out


@pytest.mark.parametrize("name", ["cornell", "sphere1", "helmet"])
def test_reference_entry_file_builds_the_same_scene(name, dropin, host_scene, tmp_path):
    out = str(tmp_path / (name + ".bin"))
    subprocess.check_call([dropin, name, out], cwd=os.path.dirname(dropin), stdout=subprocess.DEVNULL)
    blob = open(out, "rb").read()
    sec = _sections(host_scene(name))
    pos = 0
    for key in ("hdr", "nodes", "order", "tri_p", "tri_uv", "spheres", "prim_material", "materials", "textures", "lights", "camera"):
        want = sec[key]
        if name == "helmet" and key == "textures":
            # the stand-alone binary has no JPEG decoder registered and falls back to the reference's 1x1 grey texture
            # (imagemap.cpp:16-22); sizes differ, contents are covered by the golden image tests
            n_tex = host_scene(name).desc.n_textures
            pos += 24 * n_tex
            continue
        got = blob[pos:pos + len(want)]
        assert got == want, "section %s differs" % key
        pos += len(want)
    assert pos == len(blob)

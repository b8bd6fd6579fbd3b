#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE renderer's own code.

Needs /root/reference (build the driver first: `make ref`).  The driver (oracle/ref_driver.cpp) compiles the
reference's sources where they lie; this script feeds it scenes flattened by this repository's host library,
ray batches and probe points, and stores what the reference computed:

  ref_bvh_<scene>.npz      the reference's flattened BVH (nodes as raw bytes, leaf order)
  ref_rays_<scene>.npz     ray batches (camera / diffuse bounce / shadow segments / random) with the reference's
                           closest-hit primitive, t, (u,v) and any-hit answer
  ref_probe_<scene>.npz    interaction + BSDF known answers at probe hits (p, n, wo, uv, dpdu, f, Sample_f)
  ref_env_<scene>.npz      InfiniteAreaLight: radiance map, Distribution2D tables, bounding sphere, Sample_Li known answers
  ref_camera.npz           PerspectiveCamera::GenerateRay known answers
  ref_transform.npz        Transform composition known answers (matrices, inverses, transformed points)
  ref_img_<scene>_*.npy    float radiance sums rendered by PathIntegrator::Li_re with the reference's
                           StratifiedSampler, two sampler seeds each

Run from the repository root:  python tests/golden/make_golden.py
"""
import ctypes as C
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracing_amd as ra  # noqa: E402
import oracle_lib as ol      # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
TMP = tempfile.mkdtemp(prefix="rt_golden_")

from ref_io import CAMERAS, as_np, cam_spec, dump_scene as _dump_scene  # noqa: E402


def dump_scene(scene, path):
    _dump_scene(scene, path, TMP)


def run(*args):
    subprocess.check_call([DRIVER] + [str(a) for a in args])


def make_rays(scene, osc, cam, n_each, seed):
    """Four batches: camera rays, cosine-ish bounce rays off primary hits, shadow segments between hit points, random rays."""
    rng = np.random.default_rng(seed)
    W, H = cam.width, cam.height
    fx = rng.uniform(0, W, n_each).astype(np.float32); fy = rng.uniform(0, H, n_each).astype(np.float32)
    co = np.zeros((n_each, 3), np.float32); cd = np.zeros((n_each, 3), np.float32)
    for i in range(n_each):
        co[i], cd[i] = osc.camera_ray(cam, float(fx[i]), float(fy[i]))
    inf = np.full(n_each, np.inf, np.float32)
    prim, t, _, _, _ = osc.trace_closest(co, cd, inf)
    hitp = (co + cd * np.where(np.isfinite(t), t, 0)[:, None]).astype(np.float32)
    ok = prim >= 0
    # bounce rays: from slightly off the surface in a random direction flipped to the incoming side
    rd = rng.normal(size=(n_each, 3)).astype(np.float32); rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    flip = (np.sum(rd * cd, axis=1) > 0)
    rd[flip] *= -1
    bo = (hitp + rd * np.float32(2e-4)).astype(np.float32); bd = rd.astype(np.float32)
    bo[~ok] = co[~ok]; bd[~ok] = cd[~ok]
    # shadow-style segments between pairs of hit points (unnormalised direction, tMax = 1 - 1e-4, interaction.h:27-32)
    perm = rng.permutation(n_each)
    so = (hitp - cd * np.float32(1e-3)).astype(np.float32)
    sd = ((hitp[perm] - cd[perm] * np.float32(1e-3)) - so).astype(np.float32)
    st = np.full(n_each, np.float32(1) - np.float32(1e-4), np.float32)
    bad = ~(ok & ok[perm]) | (np.linalg.norm(sd, axis=1) == 0)
    so[bad] = co[bad]; sd[bad] = cd[bad]
    # random rays through the scene's bounding box
    nodes = scene.nodes()
    lo, hi = nodes[0]["bmin"], nodes[0]["bmax"]
    lo = np.maximum(lo, -30); hi = np.minimum(hi, 30)
    ro = rng.uniform(lo - 1, hi + 1, (n_each, 3)).astype(np.float32)
    rdd = rng.normal(size=(n_each, 3)).astype(np.float32); rdd /= np.linalg.norm(rdd, axis=1, keepdims=True)
    o = np.concatenate([co, bo, so, ro]); d = np.concatenate([cd, bd, sd, rdd.astype(np.float32)])
    tm = np.concatenate([inf, inf, st, inf])
    return np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32), np.ascontiguousarray(tm, np.float32)


def normalise_nodes(raw_nodes):
    """flattenBVHTree never writes a leaf's axis byte or any node's pad byte (bvh.cpp:229-245) and the node array is not cleared
    when it is allocated, so those bytes hold whatever the allocator hands out (zeros in a fresh process, leftovers of the
    texture resampling buffers in a scene that resamples one).  They carry no information: stored as zero."""
    a = np.frombuffer(raw_nodes, dtype=np.uint8).copy().reshape(-1, 32)
    leaf = (a[:, 28].astype(np.uint16) | (a[:, 29].astype(np.uint16) << 8)) > 0
    a[leaf, 30] = 0
    a[:, 31] = 0
    return a.reshape(-1)


def golden_for_scene(name, n_rays, img_specs, probe_n=512):
    print("==", name)
    if name == "mrspheres":      # the reference lacks the model's geometry buffer: the test scene's geometry is synthetic (tests/mrspheres_bin.py)
        import mrspheres_bin
        mrspheres_bin.ensure(ra.default_assets_dir())
    scene = ra.HostScene(name)
    osc = ol.OracleScene(scene.desc_ptr)
    sbin = os.path.join(TMP, name + ".scene")
    dump_scene(scene, sbin)
    # --- BVH
    out = os.path.join(TMP, name + ".bvh")
    run("bvh", sbin, out)
    raw = open(out, "rb").read()
    nn = struct.unpack("<i", raw[:4])[0]
    nodes = normalise_nodes(raw[4:4 + 32 * nn])
    order = np.frombuffer(raw[4 + 32 * nn:], dtype=np.int32).copy()
    if nn > 200000:      # a tree of tens of megabytes: its SHA-256 digests stand in for the bytes
        import hashlib
        np.savez_compressed(os.path.join(GOLD, "ref_bvh_%s.npz" % name), nodes_sha256=hashlib.sha256(nodes.tobytes()).hexdigest(),
                            order_sha256=hashlib.sha256(order.tobytes()).hexdigest(), n_nodes=nn)
    else:
        np.savez_compressed(os.path.join(GOLD, "ref_bvh_%s.npz" % name), nodes=nodes, order=order, n_nodes=nn)
    # --- ray batches
    cam = scene.camera(256, 256)
    o, d, tm = make_rays(scene, osc, cam, n_rays, 7)
    rays = np.concatenate([o, d, tm[:, None]], axis=1).astype(np.float32)
    rpath, opath = os.path.join(TMP, name + ".rays"), os.path.join(TMP, name + ".hits")
    rays.tofile(rpath)
    run("trace", sbin, rpath, opath)
    res = np.fromfile(opath, dtype=np.dtype([("prim", "<i4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("any", "<u4")]))
    np.savez_compressed(os.path.join(GOLD, "ref_rays_%s.npz" % name), o=o, d=d, tmax=tm, prim=res["prim"], t=res["t"], u=res["u"], v=res["v"],
                        any=res["any"].astype(np.uint8))
    # --- interaction / BSDF probes at camera + bounce hits
    rng = np.random.default_rng(11)
    sel = rng.choice(2 * n_rays, probe_n, replace=False)
    wi = rng.normal(size=(probe_n, 3)).astype(np.float32); wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    u01 = rng.uniform(0, 1, (probe_n, 2)).astype(np.float32)
    probes = np.concatenate([o[sel], d[sel], wi.astype(np.float32), u01], axis=1).astype(np.float32)
    ppath, qpath = os.path.join(TMP, name + ".probes"), os.path.join(TMP, name + ".probeout")
    probes.tofile(ppath)
    run("probe", sbin, ppath, qpath)
    pout = np.fromfile(qpath, dtype=np.float32).reshape(probe_n, 26)
    np.savez_compressed(os.path.join(GOLD, "ref_probe_%s.npz" % name), probes=probes, out=pout)
    # --- infinite light: the tables its constructor and Preprocess build, and Sample_Li known answers
    if scene.desc.n_env_lights:
        epath = os.path.join(TMP, name + ".env")
        run("env", sbin, epath)
        a = np.fromfile(epath, dtype=np.float32)
        w_, h_, nu, nv = (int(v) for v in a[:4]); pos = [4]

        def take(n):
            v = a[pos[0]:pos[0] + n].copy(); pos[0] += n; return v
        np.savez_compressed(os.path.join(GOLD, "ref_env_%s.npz" % name), size=np.array([w_, h_, nu, nv]), texels=take(w_ * h_ * 3), func=take(nu * nv),
                            cdf=take(nv * (nu + 1)), func_int=take(nv), marginal_cdf=take(nv + 1), marginal_int=take(1), world_center=take(3),
                            world_radius=take(1), sample_li=take(256 * 15).reshape(256, 15))
    # --- images
    for (w, h, sx, sy, seeds) in img_specs:
        cpath = os.path.join(TMP, "%s_%d.cam" % (name, w))
        open(cpath, "wb").write(cam_spec(name, w, h))
        for tag, off in seeds:
            ipath = os.path.join(TMP, "img.f32")
            run("render", sbin, cpath, sx, sy, 50, off, 8, ipath)
            img = np.fromfile(ipath, dtype=np.float32).reshape(h, w, 3)
            np.save(os.path.join(GOLD, "ref_img_%s_%dx%d_%dspp_seed%s.npy" % (name, w, h, sx * sy, tag)), img)
            print("   image", w, h, sx * sy, tag, "mean", img.mean(axis=(0, 1)) / (sx * sy))
    osc.close(); scene.close()


def golden_estimators(only):
    """Images of the estimators the reference defines next to the one its Render calls: PathIntegrator::Li (estimator 1) and paths
    whose direct lighting is UniformSampleAllLights (2: in the recursive form of Li_re, 3: in the iterative form of Li; see
    AllLightsEstimator in oracle/ref_driver.cpp).  Stored as ref_est<k>_<scene>_<w>x<h>_<spp>spp_seed<A|B>.npy."""
    cases = [(1, "cornell", 100, 16, 16), (1, "cornell_env", 100, 16, 16), (1, "spheres_env", 96, 20, 20), (1, "cornell_roughglass", 100, 16, 16),
             (2, "cornell_metal", 100, 16, 16), (3, "cornell_env", 100, 16, 16)]
    for est, name, w, sx, sy in cases:
        if only and name not in only and "est" not in only:
            continue
        scene = ra.HostScene(name)
        sbin = os.path.join(TMP, name + ".scene")
        dump_scene(scene, sbin)
        cpath = os.path.join(TMP, "%s_%d.cam" % (name, w))
        open(cpath, "wb").write(cam_spec(name, w, w))
        for tag, off in (("A", 0), ("B", 1 << 20)):
            ipath = os.path.join(TMP, "img.f32")
            run("render", sbin, cpath, sx, sy, 50, off, 8, ipath, est, 1.0)
            img = np.fromfile(ipath, dtype=np.float32).reshape(w, w, 3)
            np.save(os.path.join(GOLD, "ref_est%d_%s_%dx%d_%dspp_seed%s.npy" % (est, name, w, w, sx * sy, tag)), img)
            print("   estimator", est, name, tag, "mean", img.mean(axis=(0, 1)) / (sx * sy))
        scene.close()


def golden_bvh_variants():
    """Trees of BVHAccel's other constructor arguments (bvh.h:17-20): SAH with maxPrimsInNode 1 and 4, EqualCounts."""
    for name in ("cornell", "sphere1", "helmet"):
        scene = ra.HostScene(name)
        sbin = os.path.join(TMP, name + ".scene")
        dump_scene(scene, sbin)
        out = {}
        for tag, maxp, method in (("sah1", 1, 0), ("sah4", 4, 0), ("equal", 1, 3), ("hlbvh", 2, 1)):
            path = os.path.join(TMP, "%s_%s.bvh" % (name, tag))
            env = dict(os.environ, REF_BVH_MAXPRIMS=str(maxp), REF_BVH_SPLIT=str(method))
            subprocess.run([DRIVER, "bvh", sbin, path], check=True, env=env)
            raw = open(path, "rb").read()
            nn = struct.unpack("<i", raw[:4])[0]
            out[tag + "_nodes"] = normalise_nodes(raw[4:4 + 32 * nn])
            out[tag + "_order"] = np.frombuffer(raw[4 + 32 * nn:], dtype=np.int32).copy()
            out[tag + "_args"] = np.array([maxp, method], np.int32)
            print("  ", name, tag, nn, "nodes")
        np.savez_compressed(os.path.join(GOLD, "ref_bvh_variants_%s.npz" % name), **out)
        scene.close()


def golden_camera():
    rng = np.random.default_rng(3)
    res = {}
    for name, (w, h) in (("cornell", (800, 800)), ("sphere1", (1440, 1440)), ("helmet", (1440, 1440)), ("cornell_wide", (640, 360))):
        key = "cornell" if name.startswith("cornell") else name
        film = np.stack([rng.uniform(0, w, 256), rng.uniform(0, h, 256)], axis=1).astype(np.float32)
        film[:4] = [[0, 0], [w, h], [w / 2, h / 2], [0.5, h - 0.5]]
        cpath, fpath, opath = [os.path.join(TMP, "k." + e) for e in ("cam", "film", "out")]
        eye, look, up = CAMERAS[key]
        open(cpath, "wb").write(struct.pack("<11f2i", *eye, *look, *up, 40.0, 0.01, w, h))
        film.tofile(fpath)
        run("camera", cpath, fpath, opath)
        res[name + "_film"] = film
        res[name + "_rays"] = np.fromfile(opath, dtype=np.float32).reshape(-1, 6)
        res[name + "_spec"] = np.array(list(eye) + list(look) + list(up) + [40.0, 0.01, w, h], np.float64)
    np.savez_compressed(os.path.join(GOLD, "ref_camera.npz"), **res)


def golden_transform():
    rng = np.random.default_rng(5)
    pts = rng.uniform(-3, 3, (64, 3)).astype(np.float32)
    cases = {   # the compositions the reference's scenes use (RayTracing.cpp:83,127,155,202,306) and a few more
        "light": [(0, (2.5, 0.0, 2.5))],
        "cube": [(0, (0.7, 0.0, 2.8)), (2, (18, 0, 0))],
        "block": [(0, (2.7, 0.0, 1.0)), (2, (-15, 0, 0))],
        "helmet": [(0, (2.5, 2.5, 2.5)), (2, (-90, 0, 0)), (1, (-90, 0, 0))],
        "mrspheres": [(3, (180, 0, 0)), (2, (180, 0, 0)), (1, (-90, 0, 0))],
        "scaled": [(4, (2.0, 0.5, 3.0)), (3, (33, 0, 0)), (0, (1, -2, 3))],
    }
    res = {"pts": pts}
    for name, ops in cases.items():
        blob = struct.pack("<i", len(ops)) + np.array([k for k, _ in ops], np.int32).tobytes() + \
            np.array([a for _, a in ops], np.float32).tobytes() + struct.pack("<i", len(pts)) + pts.tobytes()
        ipath, opath = os.path.join(TMP, "t.in"), os.path.join(TMP, "t.out")
        open(ipath, "wb").write(blob)
        run("transform", ipath, opath)
        out = np.fromfile(opath, dtype=np.float32)
        res[name + "_kinds"] = np.array([k for k, _ in ops], np.int32)
        res[name + "_args"] = np.array([a for _, a in ops], np.float32)
        res[name + "_pts"] = out[:3 * len(pts)].reshape(-1, 3)
        res[name + "_m"] = out[3 * len(pts):3 * len(pts) + 16]
        res[name + "_minv"] = out[3 * len(pts) + 16:]
    np.savez_compressed(os.path.join(GOLD, "ref_transform.npz"), **res)


if __name__ == "__main__":
    if not os.path.exists(DRIVER):
        sys.exit("build the reference driver first: make ref")
    only = sys.argv[1:]
    if not only or "bvhx" in only:
        golden_bvh_variants()
    if not only or "kat" in only:
        golden_camera(); golden_transform()
    A, B = ("A", 0), ("B", 1 << 20)
    if not only or "est" in only:
        golden_estimators(only)
    if not only or "cornell" in only:
        golden_for_scene("cornell", 4096, [(64, 64, 4, 4, [A, B]), (100, 100, 16, 16, [A, B])])
    if not only or "cornell_plastic" in only:      # the short cube carries a PlasticMaterial (two-lobe BSDF, material.cpp:32-49)
        golden_for_scene("cornell_plastic", 2048, [(100, 100, 16, 16, [A, B])])
    # MetalMaterial (conductor Fresnel, anisotropic roughness, remap), a light with vertex normals, one with reverseOrientation, a two-sided one
    if not only or "cornell_metal" in only:
        golden_for_scene("cornell_metal", 2048, [(100, 100, 16, 16, [A, B])])
    # MicrofacetTransmission (reflection.cpp:311-355), which no material of the reference adds to a BSDF: the driver's test material pairs it with
    # a dielectric MicrofacetReflection on the short cube
    if not only or "cornell_roughglass" in only:
        golden_for_scene("cornell_roughglass", 2048, [(100, 100, 16, 16, [A, B])])
    # a 300 x 200 texture (resampled to powers of two by the reference itself) behind UVMapping2D(2, 3, 0.25, 0.5)
    if not only or "spheres3" in only:
        golden_for_scene("spheres3", 2048, [(96, 96, 8, 8, [A, B])])
    # InfiniteAreaLight next to area lights (Cornell box with its open front) and as a scene's only light
    if not only or "cornell_env" in only:
        golden_for_scene("cornell_env", 2048, [(100, 100, 16, 16, [A, B])])
    if not only or "spheres_env" in only:
        golden_for_scene("spheres_env", 2048, [(96, 96, 20, 20, [A, B])])      # a small sun in the map: 400 spp bring the two seeds within 0.2 %
    if not only or "sphere1" in only:
        golden_for_scene("sphere1", 4096, [(64, 64, 4, 4, [A, B])])
    if not only or "spheres2" in only:
        golden_for_scene("spheres2", 4096, [(96, 96, 8, 8, [A, B]), (160, 90, 8, 8, [A, B])])      # 160 x 90: the 1440 x 810 frame's own aspect
    if not only or "helmet" in only:
        golden_for_scene("helmet", 4096, [(96, 96, 8, 8, [A, B])])
    # MetalRoughSphere() (RayTracing.cpp:194-216) on the real .gltf and textures with a synthetic geometry buffer: five meshes, half a million triangles
    if not only or "mrspheres" in only:
        golden_for_scene("mrspheres", 2048, [(96, 96, 8, 8, [A, B])])
    print("fixtures written to", GOLD)


def golden_shipped_images():
    """Statistics of result images the reference's authors checked in (RayTracing/*.png; file names encode
    scene_resolution_spp[_seconds]).  Stored: per-channel 8-bit means, and for the 1024-spp Cornell render a
    100x100 box-downsampled copy in linear radiance (inverse of write_color: (u8/256)^2) plus a mask of
    pixels that were not clipped by the tonemap."""
    import json
    from PIL import Image
    src = "/root/reference/RayTracing"
    stats = {}
    for f in ("cb3_800_1.png", "cb3_800_256_165s.png", "cb3_800_1024_660s.png", "sphere1_1440_1.png", "sphere1_1440_512.png"):
        im = np.asarray(Image.open(os.path.join(src, f)).convert("RGB"), dtype=np.float64)
        stats[f] = {"shape": list(im.shape[:2]), "mean_u8": [float(v) for v in im.mean(axis=(0, 1))]}
    json.dump(stats, open(os.path.join(GOLD, "shipped_image_stats.json"), "w"), indent=1)
    im = np.asarray(Image.open(os.path.join(src, "cb3_800_1024_660s.png")).convert("RGB"), dtype=np.float64)
    lin = (im / 256.0) ** 2
    clipped = (im >= 255).any(axis=2)
    lin100 = lin.reshape(100, 8, 100, 8, 3).mean(axis=(1, 3)).astype(np.float32)
    ok100 = ~clipped.reshape(100, 8, 100, 8).any(axis=(1, 3))
    np.savez_compressed(os.path.join(GOLD, "shipped_cb3_800_1024_lin100.npz"), lin=lin100, unclipped=ok100)


if __name__ == "__main__" and (len(sys.argv) == 1 or "shipped" in sys.argv[1:]):
    golden_shipped_images()

"""Synthetic MetalRoughSpheres0.bin.

The reference repository ships MetalRoughSpheres.gltf and its two textures but not the 11 MB geometry buffer the .gltf names, so the
scene its MetalRoughSphere() function loads (RayTracing.cpp:194-216) cannot be had.  This writes a stand-in with exactly the layout
the .gltf states — three buffer views, twenty accessors, the same element counts, component types and byte offsets — so that the
loader's path through that real file (five meshes under one parent node, one material with two PNG textures, 16-bit indices) is
exercised on 501 776 triangles: meshes 0-3 become 64 / 128 / 128 / 128 tessellated unit spheres (960 triangles each, which is what the
counts in the .gltf divide into) laid out inside the accessors' bounds with texture coordinates spread over the accessors' uv ranges,
mesh 4 a gently waved backdrop grid plus a few coplanar extra triangles (equal hit distances: the later primitive must win, as in the
reference).  Deterministic; not the original geometry."""
import json
import os

import numpy as np


def build(gltf_path):
    j = json.load(open(gltf_path))
    buf = np.zeros(j["buffers"][0]["byteLength"], np.uint8)
    views, accs = j["bufferViews"], j["accessors"]

    def put(acc_index, arr):
        a = accs[acc_index]; v = views[a["bufferView"]]
        off = v["byteOffset"] + a.get("byteOffset", 0)
        raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        assert len(arr) == a["count"] and off + len(raw) <= v["byteOffset"] + v["byteLength"]
        buf[off:off + len(raw)] = raw

    for mi, mesh in enumerate(j["meshes"]):
        pr = mesh["primitives"][0]
        at = pr["attributes"]
        n_idx, n_vert = accs[pr["indices"]]["count"], accs[at["POSITION"]]["count"]
        lo, hi = np.array(accs[at["POSITION"]]["min"]), np.array(accs[at["POSITION"]]["max"])
        uv_lo, uv_hi = np.array(accs[at["TEXCOORD_0"]]["min"]), np.array(accs[at["TEXCOORD_0"]]["max"])
        n_tri = n_idx // 3
        P = np.zeros((n_vert, 3), np.float32); N = np.zeros((n_vert, 3), np.float32); UV = np.zeros((n_vert, 2), np.float32)
        idx = np.zeros(n_idx, np.uint16)
        if n_tri % 960 == 0:
            # K unit spheres: 32 meridians x 15 rings + two poles = 482 vertices, 2 * 32 * 14 + 2 * 32 = 960 triangles each
            K, A, R = n_tri // 960, 32, 15
            assert K * 482 <= n_vert
            cols = max(1, int(round((hi[0] - lo[0]) / 2.5))); layers = 8; rows = -(-K // (cols * layers))
            theta = np.pi * (np.arange(R) + 1) / (R + 1); phi = 2 * np.pi * np.arange(A) / A
            ring = np.stack([np.outer(np.sin(theta), np.cos(phi)), np.outer(np.cos(theta), np.ones(A)), np.outer(np.sin(theta), np.sin(phi))], axis=2).reshape(-1, 3)
            unit = np.concatenate([ring, [[0, 1, 0]], [[0, -1, 0]]]).astype(np.float32)       # 480 ring vertices, north pole, south pole
            tri = []
            for r in range(R - 1):
                for a in range(A):
                    v00, v01 = r * A + a, r * A + (a + 1) % A
                    v10, v11 = v00 + A, v01 + A
                    tri += [(v00, v10, v11), (v00, v11, v01)]
            for a in range(A):
                tri.append((480, a, (a + 1) % A))
                tri.append((481, (R - 1) * A + (a + 1) % A, (R - 1) * A + a))
            tri = np.array(tri, np.int64)
            assert len(tri) == 960
            for s in range(K):
                c, rest = s % cols, s // cols
                layer, row = rest % layers, rest // layers
                centre = np.array([lo[0] + (c + 0.5) * (hi[0] - lo[0]) / cols, lo[1] + 1 + (row + 0.5) * (hi[1] - lo[1] - 2) / rows,
                                   lo[2] + 1 + (layer + 0.5) * (hi[2] - lo[2] - 2) / layers])
                radius = 0.45 * min((hi[0] - lo[0]) / cols, (hi[1] - lo[1] - 2) / rows, (hi[2] - lo[2] - 2) / layers, 2.0)
                v0 = s * 482
                P[v0:v0 + 482] = centre + radius * unit
                N[v0:v0 + 482] = unit
                cell = np.array([(s % 8 + 0.5) / 8, (s // 8 + 0.5) / -(-K // 8)])
                local = np.stack([np.arctan2(unit[:, 2], unit[:, 0]) / (2 * np.pi) * 0.08, np.arccos(np.clip(unit[:, 1], -1, 1)) / np.pi * 0.08 - 0.04], axis=1)
                UV[v0:v0 + 482] = uv_lo + (uv_hi - uv_lo) * np.clip(cell + local, 0, 1)
                idx[s * 2880:(s + 1) * 2880] = (tri + v0).reshape(-1)
            P[K * 482:] = P[480]; N[K * 482:] = N[480]; UV[K * 482:] = UV[480]            # vertices no triangle uses
        else:
            # backdrop: nx x ny cells, two triangles each, then the other diagonal's triangles of the first cells (coplanar overlaps)
            nx = 187; ny = min((n_tri // 2) // nx, n_vert // (nx + 1) - 1)
            assert (nx + 1) * (ny + 1) <= n_vert and 2 * nx * ny <= n_tri
            gx, gz = np.meshgrid(np.linspace(lo[0], hi[0], nx + 1), np.linspace(lo[2], hi[2], ny + 1))
            gy = lo[1] + 0.15 * (1 + np.sin(gx * 0.9) * np.cos(gz * 0.7))
            nv = (nx + 1) * (ny + 1)
            P[:nv] = np.stack([gx, gy, gz], axis=2).reshape(-1, 3)
            N[:nv] = [0, 1, 0]
            UV[:nv] = uv_lo + (uv_hi - uv_lo) * np.stack([(gx - lo[0]) / (hi[0] - lo[0]), (gz - lo[2]) / (hi[2] - lo[2])], axis=2).reshape(-1, 2)
            P[nv:] = P[0]; N[nv:] = N[0]; UV[nv:] = UV[0]
            tri = []
            for y in range(ny):
                for x in range(nx):
                    v00 = y * (nx + 1) + x
                    tri += [(v00, v00 + nx + 1, v00 + nx + 2), (v00, v00 + nx + 2, v00 + 1)]
            extra = n_tri - len(tri)
            for k in range(extra):
                v00 = (k // 2) * 7 % (nx * ny); v00 = (v00 // nx) * (nx + 1) + v00 % nx
                tri.append((v00, v00 + nx + 1, v00 + 1) if k % 2 == 0 else (v00 + 1, v00 + nx + 1, v00 + nx + 2))
            idx[:] = np.array(tri, np.int64).reshape(-1)
        assert idx.max() < n_vert
        put(pr["indices"], idx); put(at["POSITION"], P); put(at["NORMAL"], N); put(at["TEXCOORD_0"], UV)
    return buf


def ensure(assets_dir):
    """Write <assets>/MetalRoughSpheres/glTF/MetalRoughSpheres0.bin unless it is there already; returns its path."""
    d = os.path.join(assets_dir, "MetalRoughSpheres", "glTF")
    out = os.path.join(d, "MetalRoughSpheres0.bin")
    want = json.load(open(os.path.join(d, "MetalRoughSpheres.gltf")))["buffers"][0]["byteLength"]
    if not os.path.exists(out) or os.path.getsize(out) != want:
        tmp = out + ".tmp%d" % os.getpid()
        build(os.path.join(d, "MetalRoughSpheres.gltf")).tofile(tmp)
        os.replace(tmp, out)
    return out


if __name__ == "__main__":
    import sys
    print(ensure(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "assets")))

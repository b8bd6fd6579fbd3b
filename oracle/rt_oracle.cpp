// rt_oracle.cpp — CPU restatement of the reference renderer's hot path.
//
// TEST INFRASTRUCTURE ONLY.  This file is the checker the HIP path is compared
// against (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  The
// shipped library never links, loads or calls it.
//
// It restates, over the flat rt_scene_desc arrays of include/rt_amd.h, the
// algorithm of sedcftyv/RayTracing (paths relative to RayTracing/include/core/):
//   BVH build            bvh.cpp:63-245
//   BVH traversal        bvh.cpp:247-320, geometry.h:895-914
//   triangle / sphere    triangle.cpp:52-248, sphere.cpp:29-93
//   camera               camera.cpp:52-58
//   sampler              rng.h:85-100, sampling.cpp:5-23, sampler.cpp:7-82, stratified.cpp:4-35
//   path estimator       integrator.cpp:53-102,236-281,289-308
//   lights               light.cpp:43-54, shape.cpp:20-31, interaction.h:18-32, infinite.cpp:83-106, sampling.h:15-91
//   materials / BSDFs    material.cpp:5-82, reflection.cpp:6-300, reflection.h:11-180
//   microfacets          microfacet.cpp:84-107,149-236, microfacet.h:14-78
//   textures             texture.cpp:8-13, mipmap.h:152-200
// Float expressions keep the reference's operand order and its float/double
// promotions; build with -O2 -ffp-contract=off and no fast-math.
//
// Pinning: oracle/ref_driver.cpp compiles the reference's own sources and
// tests/test_oracle_vs_reference.py + tests/golden/ compare the two.

#include "../include/rt_amd.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

typedef float Float;

namespace {

const Float Pi = 3.14159265358979323846;
const Float InvPi = 0.31830988618379067154;
const Float Inv2Pi = 0.15915494309189533577;
const Float PiOver2 = 1.57079632679489661923;
const Float PiOver4 = 0.78539816339744830961;
const Float ShadowEpsilon = 0.0001f;
const Float kEps = 1e-4;
const Float MachineEpsilon = std::numeric_limits<Float>::epsilon() * 0.5;
const Float Infinity = std::numeric_limits<Float>::infinity();
const Float OneMinusEpsilon = 0.99999994;

inline Float gammaN(int n) { return (n * MachineEpsilon) / (1 - n * MachineEpsilon); }
template <typename T, typename U, typename V> inline T Clamp(T val, U low, V high) {
    if (val < low) return low; else if (val > high) return high; else return val;
}
inline int ModI(int a, int b) { int r = a - (a / b) * b; return (r < 0) ? r + b : r; }

// ---------------------------------------------------------------- vectors --
struct V3 { Float x, y, z; Float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); } };
inline V3 mk(Float x, Float y, Float z) { V3 v = {x, y, z}; return v; }
inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 v, Float s) { return mk(s * v.x, s * v.y, s * v.z); }   // geometry.h:144,294
inline V3 operator*(Float s, V3 v) { return v * s; }
inline V3 operator/(V3 v, Float f) { Float inv = (Float)1 / f; return mk(v.x * inv, v.y * inv, v.z * inv); }
inline Float Dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Float AbsDot(V3 a, V3 b) { return std::abs(Dot(a, b)); }
inline V3 Cross(V3 a, V3 b) {  // geometry.h:183-190 (double intermediates)
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return mk((ay * bz) - (az * by), (az * bx) - (ax * bz), (ax * by) - (ay * bx));
}
inline Float LengthSquared(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
inline Float Length(V3 v) { return std::sqrt(LengthSquared(v)); }
inline V3 Normalize(V3 v) { return v / Length(v); }
inline V3 AbsV(V3 v) { return mk(std::abs(v.x), std::abs(v.y), std::abs(v.z)); }
inline int MaxDimension(V3 v) { return (v.x > v.y) ? ((v.x > v.z) ? 0 : 2) : ((v.y > v.z) ? 1 : 2); }
inline V3 Permute(V3 v, int x, int y, int z) { return mk(v[x], v[y], v[z]); }
inline V3 Faceforward(V3 n, V3 v) { return (Dot(n, v) < 0.f) ? -n : n; }
inline void CoordinateSystem(V3 v1, V3 *v2, V3 *v3) {  // geometry.h:222-231
    if (std::abs(v1.x) > std::abs(v1.y))
        *v2 = mk(-v1.z, 0, v1.x) / std::sqrt(v1.x * v1.x + v1.z * v1.z);
    else
        *v2 = mk(0, v1.z, -v1.y) / std::sqrt(v1.y * v1.y + v1.z * v1.z);
    *v3 = Cross(v1, *v2);
}
struct RGB { Float c[3]; Float &operator[](int i) { return c[i]; } Float operator[](int i) const { return c[i]; } };
inline RGB rgb(Float a, Float b, Float c) { RGB r = {{a, b, c}}; return r; }
inline RGB rgb1(Float v) { return rgb(v, v, v); }
inline RGB operator+(RGB a, RGB b) { return rgb(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline RGB operator-(RGB a, RGB b) { return rgb(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline RGB operator*(RGB a, RGB b) { return rgb(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
inline RGB operator/(RGB a, RGB b) { return rgb(a[0] / b[0], a[1] / b[1], a[2] / b[2]); }
inline RGB operator*(RGB a, Float s) { return rgb(a[0] * s, a[1] * s, a[2] * s); }
inline RGB operator*(Float s, RGB a) { return a * s; }
inline RGB operator/(RGB a, Float s) { return rgb(a[0] / s, a[1] / s, a[2] / s); }
inline bool IsBlack(RGB a) { return a[0] == 0. && a[1] == 0. && a[2] == 0.; }
inline RGB ClampRGB(RGB a) { return rgb(Clamp(a[0], (Float)0, Infinity), Clamp(a[1], (Float)0, Infinity), Clamp(a[2], (Float)0, Infinity)); }
inline RGB SqrtRGB(RGB a) { return rgb(std::sqrt(a[0]), std::sqrt(a[1]), std::sqrt(a[2])); }
inline Float MaxComp(RGB a) { Float m = a[0]; m = std::max(m, a[1]); m = std::max(m, a[2]); return m; }

// transform.h:135-160 — row-major 4x4 applied to points / vectors / normals.
inline V3 XfPoint(const float *m, V3 p) {
    Float x = p.x, y = p.y, z = p.z;
    Float xp = m[0] * x + m[1] * y + m[2] * z + m[3];
    Float yp = m[4] * x + m[5] * y + m[6] * z + m[7];
    Float zp = m[8] * x + m[9] * y + m[10] * z + m[11];
    Float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
    if (wp == 1) return mk(xp, yp, zp);
    Float inv = (Float)1 / wp;
    return mk(inv * xp, inv * yp, inv * zp);
}
inline V3 XfVector(const float *m, V3 v) {
    Float x = v.x, y = v.y, z = v.z;
    return mk(m[0] * x + m[1] * y + m[2] * z, m[4] * x + m[5] * y + m[6] * z, m[8] * x + m[9] * y + m[10] * z);
}
inline V3 XfNormal(const float *mInv, V3 n) {  // uses the INVERSE matrix, transposed
    Float x = n.x, y = n.y, z = n.z;
    return mk(mInv[0] * x + mInv[4] * y + mInv[8] * z, mInv[1] * x + mInv[5] * y + mInv[9] * z,
              mInv[2] * x + mInv[6] * y + mInv[10] * z);
}

struct Ray { V3 o, d; mutable Float tMax; };

// geometry.h:916-923
inline V3 OffsetRayOrigin(V3 p, V3 n, V3 w) {
    Float d = Dot(AbsV(n), mk(1e-4, 1e-4, 1e-4));
    V3 offset = d * n;
    if (Dot(w, n) < 0) offset = -offset;
    return p + offset;
}

// ------------------------------------------------------------------ scene --
struct Scene {
    // deep copies of the desc
    std::vector<uint8_t> prim_kind; std::vector<int32_t> prim_geom, prim_material, prim_light;
    std::vector<float> tri_p, tri_uv, tri_n; std::vector<uint8_t> tri_has_n;
    std::vector<rt_sphere> spheres; std::vector<rt_material> materials; std::vector<rt_texture> textures;
    std::vector<std::vector<float>> texel_store; std::vector<rt_light> lights;
    std::vector<rt_bvh_node> nodes; std::vector<int32_t> order;
    int n_prims = 0;
    std::vector<rt_env_light> envs; std::vector<std::vector<float>> env_store;    // infinite lights and the tables they point at
    std::vector<float> light_cdf; float light_func_int = 0;     // Distribution1D of UniformLightDistribution (see LightDistrib)
};

struct Bounds { V3 pMin, pMax; };
inline Bounds EmptyBounds() {
    Float lo = std::numeric_limits<Float>::lowest(), hi = std::numeric_limits<Float>::max();
    Bounds b = {mk(hi, hi, hi), mk(lo, lo, lo)}; return b;
}
inline Bounds Bounds2P(V3 a, V3 b) {
    Bounds r = {mk(std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)),
                mk(std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z))}; return r;
}
inline Bounds UnionBP(Bounds b, V3 p) {
    Bounds r = {mk(std::min(b.pMin.x, p.x), std::min(b.pMin.y, p.y), std::min(b.pMin.z, p.z)),
                mk(std::max(b.pMax.x, p.x), std::max(b.pMax.y, p.y), std::max(b.pMax.z, p.z))}; return r;
}
inline Bounds UnionBB(Bounds a, Bounds b) {
    Bounds r = {mk(std::min(a.pMin.x, b.pMin.x), std::min(a.pMin.y, b.pMin.y), std::min(a.pMin.z, b.pMin.z)),
                mk(std::max(a.pMax.x, b.pMax.x), std::max(a.pMax.y, b.pMax.y), std::max(a.pMax.z, b.pMax.z))}; return r;
}
inline int MaximumExtent(Bounds b) {
    V3 d = b.pMax - b.pMin;
    if (d.x > d.y && d.x > d.z) return 0; else if (d.y > d.z) return 1; else return 2;
}

// Bounds3::SurfaceArea geometry.h:772-775, Bounds3::Offset geometry.h:794-800
inline Float SurfaceAreaB(Bounds b) { V3 d = b.pMax - b.pMin; return 2 * (d.x * d.y + d.x * d.z + d.y * d.z); }
inline V3 OffsetB(Bounds b, V3 p) {
    V3 o = p - b.pMin;
    if (b.pMax.x > b.pMin.x) o.x /= b.pMax.x - b.pMin.x;
    if (b.pMax.y > b.pMin.y) o.y /= b.pMax.y - b.pMin.y;
    if (b.pMax.z > b.pMin.z) o.z /= b.pMax.z - b.pMin.z;
    return o;
}

// ------------------------------------------------------------- BVH build --
// bvh.cpp:9-18
struct PrimInfo { size_t primitiveNumber; Bounds bounds; V3 centroid; };
struct BuildNode { Bounds bounds; BuildNode *children[2]; int splitAxis, firstPrimOffset, nPrimitives; };

struct Builder {
    std::vector<PrimInfo> info; std::vector<int32_t> ordered; std::vector<BuildNode *> pool; int totalNodes = 0;
    int maxPrimsInNode = 1, splitMethod = 2;   // bvh.h:17-20: enum SplitMethod { SAH, HLBVH, Middle, EqualCounts }
    BuildNode *alloc() { BuildNode *n = new BuildNode; pool.push_back(n); return n; }
    ~Builder() { for (BuildNode *n : pool) delete n; }
    // bvh.cpp:86-227
    BuildNode *recursiveBuild(int start, int end) {
        BuildNode *node = alloc();
        totalNodes++;
        Bounds bounds = EmptyBounds();
        for (int i = start; i < end; ++i) bounds = UnionBB(bounds, info[i].bounds);
        int nPrimitives = end - start;
        auto leaf = [&]() {
            int first = (int)ordered.size();
            for (int i = start; i < end; ++i) ordered.push_back((int32_t)info[i].primitiveNumber);
            node->firstPrimOffset = first; node->nPrimitives = nPrimitives; node->bounds = bounds;
            node->children[0] = node->children[1] = nullptr;
            return node;
        };
        if (nPrimitives == 1) return leaf();
        Bounds cb = EmptyBounds();
        for (int i = start; i < end; ++i) cb = UnionBP(cb, info[i].centroid);
        int dim = MaximumExtent(cb);
        int mid = (start + end) / 2;
        if (cb.pMax[dim] == cb.pMin[dim]) return leaf();
        switch (splitMethod) {
        case 2: {   // SplitMethod::Middle, bvh.cpp:122-132
            Float pmid = (cb.pMin[dim] + cb.pMax[dim]) / 2;
            PrimInfo *midPtr = std::partition(&info[start], &info[end - 1] + 1,
                                              [dim, pmid](const PrimInfo &pi) { return pi.centroid[dim] < pmid; });
            mid = (int)(midPtr - &info[0]);
            if (mid != start && mid != end) break;
        }   // falls through, as the reference does
        case 3: {   // SplitMethod::EqualCounts, bvh.cpp:133-142
            mid = (start + end) / 2;
            std::nth_element(&info[start], &info[mid], &info[end - 1] + 1,
                             [dim](const PrimInfo &a, const PrimInfo &b) { return a.centroid[dim] < b.centroid[dim]; });
            break;
        }
        default: {  // SplitMethod::SAH and HLBVH, bvh.cpp:143-217
            if (nPrimitives <= 2) {
                mid = (start + end) / 2;
                std::nth_element(&info[start], &info[mid], &info[end - 1] + 1,
                                 [dim](const PrimInfo &a, const PrimInfo &b) { return a.centroid[dim] < b.centroid[dim]; });
            } else {
                const int nBuckets = 12;
                struct Bucket { int count = 0; Bounds bounds = EmptyBounds(); } buckets[nBuckets];
                auto bucketOf = [&cb, dim](const PrimInfo &pi) {
                    int b = nBuckets * OffsetB(cb, pi.centroid)[dim];
                    if (b == nBuckets) b = nBuckets - 1;
                    return b;
                };
                for (int i = start; i < end; ++i) {
                    int b = bucketOf(info[i]);
                    buckets[b].count++;
                    buckets[b].bounds = UnionBB(buckets[b].bounds, info[i].bounds);
                }
                Float cost[nBuckets - 1];
                for (int i = 0; i < nBuckets - 1; ++i) {
                    Bounds b0 = EmptyBounds(), b1 = EmptyBounds();
                    int count0 = 0, count1 = 0;
                    for (int j = 0; j <= i; ++j) { b0 = UnionBB(b0, buckets[j].bounds); count0 += buckets[j].count; }
                    for (int j = i + 1; j < nBuckets; ++j) { b1 = UnionBB(b1, buckets[j].bounds); count1 += buckets[j].count; }
                    cost[i] = 1 + (count0 * SurfaceAreaB(b0) + count1 * SurfaceAreaB(b1)) / SurfaceAreaB(bounds);
                }
                Float minCost = cost[0];
                int minCostSplitBucket = 0;
                for (int i = 1; i < nBuckets - 1; ++i)
                    if (cost[i] < minCost) { minCost = cost[i]; minCostSplitBucket = i; }
                Float leafCost = nPrimitives;
                if (nPrimitives > maxPrimsInNode || minCost < leafCost) {
                    PrimInfo *pmid = std::partition(&info[start], &info[end - 1] + 1,
                                                    [&](const PrimInfo &pi) { return bucketOf(pi) <= minCostSplitBucket; });
                    mid = (int)(pmid - &info[0]);
                } else return leaf();
            }
            break;
        }
        }
        BuildNode *c0 = recursiveBuild(start, mid);
        BuildNode *c1 = recursiveBuild(mid, end);
        node->children[0] = c0; node->children[1] = c1;
        node->bounds = UnionBB(c0->bounds, c1->bounds);
        node->splitAxis = dim; node->nPrimitives = 0;
        return node;
    }
    // bvh.cpp:229-245
    int flatten(BuildNode *node, rt_bvh_node *nodes, int *offset) {
        rt_bvh_node *ln = &nodes[*offset];
        memset(ln, 0, sizeof(*ln));
        ln->bmin[0] = node->bounds.pMin.x; ln->bmin[1] = node->bounds.pMin.y; ln->bmin[2] = node->bounds.pMin.z;
        ln->bmax[0] = node->bounds.pMax.x; ln->bmax[1] = node->bounds.pMax.y; ln->bmax[2] = node->bounds.pMax.z;
        int myOffset = (*offset)++;
        if (node->nPrimitives > 0) {
            ln->offset = node->firstPrimOffset; ln->nprim = (uint16_t)node->nPrimitives;
        } else {
            ln->axis = (uint8_t)node->splitAxis; ln->nprim = 0;
            flatten(node->children[0], nodes, offset);
            ln->offset = flatten(node->children[1], nodes, offset);
        }
        return myOffset;
    }
};

int bvh_build(int n, const float *pb, rt_bvh_node *nodes_out, int32_t *n_nodes_out, int32_t *order_out, int maxPrims = 1, int method = 2) {
    if (n <= 0) { *n_nodes_out = 0; return 0; }
    Builder b; b.info.resize(n);
    b.maxPrimsInNode = std::min(255, maxPrims); b.splitMethod = method;   // bvh.cpp:65
    for (int i = 0; i < n; ++i) {
        Bounds bb = {mk(pb[6 * i], pb[6 * i + 1], pb[6 * i + 2]), mk(pb[6 * i + 3], pb[6 * i + 4], pb[6 * i + 5])};
        b.info[i].primitiveNumber = i; b.info[i].bounds = bb;
        b.info[i].centroid = .5f * bb.pMin + .5f * bb.pMax;  // bvh.cpp:14
    }
    b.ordered.reserve(n);
    BuildNode *root = b.recursiveBuild(0, n);
    int offset = 0;
    b.flatten(root, nodes_out, &offset);
    *n_nodes_out = b.totalNodes;
    memcpy(order_out, b.ordered.data(), sizeof(int32_t) * n);
    return 0;
}

// Triangle::WorldBound triangle.cpp:45-50; Sphere: Shape::WorldBound shape.cpp:16 + transform.cpp:181-192
void prim_bounds(const rt_scene_desc *d, float *out) {
    for (int i = 0; i < d->n_prims; ++i) {
        Bounds b;
        if (d->prim_kind[i] == RT_PRIM_TRIANGLE) {
            const float *p = d->tri_p + 9 * (size_t)d->prim_geom[i];
            b = UnionBP(Bounds2P(mk(p[0], p[1], p[2]), mk(p[3], p[4], p[5])), mk(p[6], p[7], p[8]));
        } else {
            const rt_sphere &s = d->spheres[d->prim_geom[i]];
            Float r = s.radius; V3 lo = mk(-r, -r, -r), hi = mk(r, r, r);
            const float *M = s.o2w;
            Bounds ret = {XfPoint(M, mk(lo.x, lo.y, lo.z)), XfPoint(M, mk(lo.x, lo.y, lo.z))};
            ret = UnionBP(ret, XfPoint(M, mk(hi.x, lo.y, lo.z)));
            ret = UnionBP(ret, XfPoint(M, mk(lo.x, hi.y, lo.z)));
            ret = UnionBP(ret, XfPoint(M, mk(lo.x, lo.y, hi.z)));
            ret = UnionBP(ret, XfPoint(M, mk(lo.x, hi.y, hi.z)));
            ret = UnionBP(ret, XfPoint(M, mk(hi.x, hi.y, lo.z)));
            ret = UnionBP(ret, XfPoint(M, mk(hi.x, lo.y, hi.z)));
            ret = UnionBP(ret, XfPoint(M, mk(hi.x, hi.y, hi.z)));
            b = ret;
        }
        out[6 * i + 0] = b.pMin.x; out[6 * i + 1] = b.pMin.y; out[6 * i + 2] = b.pMin.z;
        out[6 * i + 3] = b.pMax.x; out[6 * i + 4] = b.pMax.y; out[6 * i + 5] = b.pMax.z;
    }
}

// ------------------------------------------------------------ intersection --
struct Hit {
    int prim;            // ORIGINAL primitive index, -1 = none
    Float t, b0, b1, b2; // barycentrics (triangles)
};

// geometry.h:895-914
inline bool SlabTest(const rt_bvh_node &nd, const Ray &ray, V3 invDir, const int dirIsNeg[3]) {
    const float *bnd[2] = {nd.bmin, nd.bmax};
    Float tMin = (bnd[dirIsNeg[0]][0] - ray.o.x) * invDir.x;
    Float tMax = (bnd[1 - dirIsNeg[0]][0] - ray.o.x) * invDir.x;
    Float tyMin = (bnd[dirIsNeg[1]][1] - ray.o.y) * invDir.y;
    Float tyMax = (bnd[1 - dirIsNeg[1]][1] - ray.o.y) * invDir.y;
    tMax *= 1 + 2 * gammaN(3);
    tyMax *= 1 + 2 * gammaN(3);
    if (tMin > tyMax || tyMin > tMax) return false;
    if (tyMin > tMin) tMin = tyMin;
    if (tyMax < tMax) tMax = tyMax;
    Float tzMin = (bnd[dirIsNeg[2]][2] - ray.o.z) * invDir.z;
    Float tzMax = (bnd[1 - dirIsNeg[2]][2] - ray.o.z) * invDir.z;
    tzMax *= 1 + 2 * gammaN(3);
    if (tMin > tzMax || tzMin > tMax) return false;
    if (tzMin > tMin) tMin = tzMin;
    if (tzMax < tMax) tMax = tzMax;
    return (tMin < ray.tMax) && (tMax > 0);
}

// triangle.cpp:52-119 (shared by Intersect and IntersectP)
inline bool TriangleCore(const float *P, const Ray &ray, Float *t, Float *b0o, Float *b1o, Float *b2o) {
    V3 p0 = mk(P[0], P[1], P[2]), p1 = mk(P[3], P[4], P[5]), p2 = mk(P[6], P[7], P[8]);
    V3 p0t = p0 - ray.o, p1t = p1 - ray.o, p2t = p2 - ray.o;
    int kz = MaxDimension(AbsV(ray.d));
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    V3 d = Permute(ray.d, kx, ky, kz);
    p0t = Permute(p0t, kx, ky, kz); p1t = Permute(p1t, kx, ky, kz); p2t = Permute(p2t, kx, ky, kz);
    Float Sx = -d.x / d.z, Sy = -d.y / d.z, Sz = 1.f / d.z;
    p0t.x += Sx * p0t.z; p0t.y += Sy * p0t.z;
    p1t.x += Sx * p1t.z; p1t.y += Sy * p1t.z;
    p2t.x += Sx * p2t.z; p2t.y += Sy * p2t.z;
    Float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    Float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    Float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        double p2txp1ty = (double)p2t.x * (double)p1t.y, p2typ1tx = (double)p2t.y * (double)p1t.x;
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)p0t.x * (double)p2t.y, p0typ2tx = (double)p0t.y * (double)p2t.x;
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)p1t.x * (double)p0t.y, p1typ0tx = (double)p1t.y * (double)p0t.x;
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
    Float det = e0 + e1 + e2;
    if (det == 0) return false;
    p0t.z *= Sz; p1t.z *= Sz; p2t.z *= Sz;
    Float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0 && (tScaled >= 0 || tScaled < ray.tMax * det)) return false;
    else if (det > 0 && (tScaled <= 0 || tScaled > ray.tMax * det)) return false;
    Float invDet = 1 / det;
    *b0o = e0 * invDet; *b1o = e1 * invDet; *b2o = e2 * invDet;
    *t = tScaled * invDet;
    if (*t <= kEps) return false;
    return true;
}

// triangle.cpp:121-144: partial derivatives; returns false when the reference
// rejects the hit because the triangle is degenerate.
inline bool TriangleDerivs(const float *P, const float *UV, V3 *dpdu, V3 *dpdv) {
    V3 p0 = mk(P[0], P[1], P[2]), p1 = mk(P[3], P[4], P[5]), p2 = mk(P[6], P[7], P[8]);
    Float duv02x = UV[0] - UV[4], duv02y = UV[1] - UV[5], duv12x = UV[2] - UV[4], duv12y = UV[3] - UV[5];
    V3 dp02 = p0 - p2, dp12 = p1 - p2;
    Float determinant = duv02x * duv12y - duv02y * duv12x;
    bool degenerateUV = std::abs(determinant) < 1e-8;
    if (!degenerateUV) {
        Float invdet = 1 / determinant;
        *dpdu = (duv12y * dp02 - duv02y * dp12) * invdet;
        *dpdv = (-duv12x * dp02 + duv02x * dp12) * invdet;
    }
    if (degenerateUV || LengthSquared(Cross(*dpdu, *dpdv)) == 0) {
        V3 ng = Cross(p2 - p0, p1 - p0);
        if (LengthSquared(ng) == 0) return false;
        CoordinateSystem(Normalize(ng), dpdu, dpdv);
    }
    return true;
}

// sphere.cpp:29-50 / 75-92
inline bool SphereCore(const rt_sphere &s, const Ray &r, Float *tHit) {
    V3 o = XfPoint(s.w2o, r.o), d = XfVector(s.w2o, r.d);
    V3 oc = o - mk(0.0f, 0.0f, 0.0f);
    Float a = LengthSquared(d);
    Float half_b = Dot(oc, d);
    Float c = LengthSquared(oc) - s.radius * s.radius;
    Float discriminant = half_b * half_b - a * c;
    if (discriminant < 0) return false;
    Float sqrtd = std::sqrt(discriminant);
    Float t0 = (-half_b - sqrtd) / a;
    Float t1 = (-half_b + sqrtd) / a;
    if (t0 > r.tMax || t1 <= 0) return false;
    Float tShapeHit = t0;
    if (tShapeHit <= 0) { tShapeHit = t1; if (tShapeHit > r.tMax) return false; }
    *tHit = tShapeHit;
    return true;
}

struct TraceCounters { uint32_t nodes, tests; };

// bvh.cpp:247-282 + primitive.cpp:35-42
bool IntersectClosest(const Scene &sc, const Ray &ray, Hit *hit, TraceCounters *cnt) {
    hit->prim = -1; hit->t = ray.tMax; hit->b0 = hit->b1 = hit->b2 = 0;
    if (sc.nodes.empty()) return false;
    bool found = false;
    V3 invDir = mk(1 / ray.d.x, 1 / ray.d.y, 1 / ray.d.z);
    int dirIsNeg[3] = {invDir.x < 0, invDir.y < 0, invDir.z < 0};
    int toVisitOffset = 0, cur = 0; int nodesToVisit[64];
    while (true) {
        const rt_bvh_node &node = sc.nodes[cur];
        if (cnt) cnt->nodes++;
        if (SlabTest(node, ray, invDir, dirIsNeg)) {
            if (node.nprim > 0) {
                for (int i = 0; i < node.nprim; ++i) {
                    int prim = sc.order[node.offset + i];
                    if (cnt) cnt->tests++;
                    Float t, b0 = 0, b1 = 0, b2 = 0; bool ok;
                    if (sc.prim_kind[prim] == RT_PRIM_TRIANGLE) {
                        size_t g = (size_t)sc.prim_geom[prim];
                        ok = TriangleCore(&sc.tri_p[9 * g], ray, &t, &b0, &b1, &b2);
                        if (ok) { V3 du, dv; ok = TriangleDerivs(&sc.tri_p[9 * g], &sc.tri_uv[6 * g], &du, &dv); }
                    } else {
                        ok = SphereCore(sc.spheres[sc.prim_geom[prim]], ray, &t);
                    }
                    if (ok) { found = true; ray.tMax = t; hit->prim = prim; hit->t = t; hit->b0 = b0; hit->b1 = b1; hit->b2 = b2; }
                }
                if (toVisitOffset == 0) break;
                cur = nodesToVisit[--toVisitOffset];
            } else {
                if (dirIsNeg[node.axis]) { nodesToVisit[toVisitOffset++] = cur + 1; cur = node.offset; }
                else { nodesToVisit[toVisitOffset++] = node.offset; cur = cur + 1; }
            }
        } else {
            if (toVisitOffset == 0) break;
            cur = nodesToVisit[--toVisitOffset];
        }
    }
    return found;
}

// bvh.cpp:284-320
bool IntersectAny(const Scene &sc, const Ray &ray, TraceCounters *cnt) {
    if (sc.nodes.empty()) return false;
    V3 invDir = mk(1.f / ray.d.x, 1.f / ray.d.y, 1.f / ray.d.z);
    int dirIsNeg[3] = {invDir.x < 0, invDir.y < 0, invDir.z < 0};
    int nodesToVisit[64]; int toVisitOffset = 0, cur = 0;
    while (true) {
        const rt_bvh_node &node = sc.nodes[cur];
        if (cnt) cnt->nodes++;
        if (SlabTest(node, ray, invDir, dirIsNeg)) {
            if (node.nprim > 0) {
                for (int i = 0; i < node.nprim; ++i) {
                    int prim = sc.order[node.offset + i];
                    if (cnt) cnt->tests++;
                    Float t, b0, b1, b2;
                    if (sc.prim_kind[prim] == RT_PRIM_TRIANGLE) {
                        if (TriangleCore(&sc.tri_p[9 * (size_t)sc.prim_geom[prim]], ray, &t, &b0, &b1, &b2)) return true;
                    } else if (SphereCore(sc.spheres[sc.prim_geom[prim]], ray, &t)) return true;
                }
                if (toVisitOffset == 0) break;
                cur = nodesToVisit[--toVisitOffset];
            } else {
                if (dirIsNeg[node.axis]) { nodesToVisit[toVisitOffset++] = cur + 1; cur = node.offset; }
                else { nodesToVisit[toVisitOffset++] = node.offset; cur = cur + 1; }
            }
        } else {
            if (toVisitOffset == 0) break;
            cur = nodesToVisit[--toVisitOffset];
        }
    }
    return false;
}

// ------------------------------------------------------- surface interaction --
struct SurfaceInteraction {
    V3 p, n, wo;        // n = geometric normal (== shading normal here)
    Float u, v;
    V3 dpdu;
    int prim;
};

// triangle.cpp:121-146 and sphere.cpp:51-70 + transform.cpp:205-235
void FillInteraction(const Scene &sc, const Ray &ray, const Hit &h, SurfaceInteraction *si) {
    si->prim = h.prim;
    if (sc.prim_kind[h.prim] == RT_PRIM_TRIANGLE) {
        size_t g = (size_t)sc.prim_geom[h.prim];
        const float *P = &sc.tri_p[9 * g], *UV = &sc.tri_uv[6 * g];
        V3 p0 = mk(P[0], P[1], P[2]), p1 = mk(P[3], P[4], P[5]), p2 = mk(P[6], P[7], P[8]);
        V3 dpdu = mk(0, 0, 0), dpdv = mk(0, 0, 0);
        TriangleDerivs(P, UV, &dpdu, &dpdv);
        si->u = h.b0 * UV[0] + h.b1 * UV[2] + h.b2 * UV[4];
        si->v = h.b0 * UV[1] + h.b1 * UV[3] + h.b2 * UV[5];
        si->p = h.b0 * p0 + h.b1 * p1 + h.b2 * p2;
        si->wo = -ray.d;
        si->dpdu = dpdu;
        si->n = Normalize(Cross(p0 - p2, p1 - p2));
    } else {
        const rt_sphere &s = sc.spheres[sc.prim_geom[h.prim]];
        V3 o = XfPoint(s.w2o, ray.o), d = XfVector(s.w2o, ray.d);
        V3 pHit = o + d * h.t;
        Float phi = std::atan2(pHit.y, pHit.x);
        if (phi < 0) phi += 2 * Pi;
        Float phiMax = 2 * Pi;
        si->u = phi / phiMax;
        Float theta = std::acos(Clamp(pHit.z / s.radius, -1, 1));
        si->v = theta / Pi;
        V3 dpdu = mk(-phiMax * pHit.y, phiMax * pHit.x, 0);
        V3 n1 = Normalize(pHit);
        si->p = XfPoint(s.o2w, pHit);
        si->wo = Normalize(XfVector(s.o2w, -d));
        si->dpdu = XfVector(s.o2w, dpdu);
        si->n = XfNormal(s.w2o, n1);   // (*ObjectToWorld)(Normal3f): inverse-transpose of o2w
    }
}

// ---------------------------------------------------------------- textures --
RGB EvalTexture(const Scene &sc, int tex, Float u, Float v) {
    const rt_texture &t = sc.textures[tex];
    if (t.kind == RT_TEX_CONST) return rgb(t.value[0], t.value[1], t.value[2]);
    // texture.cpp:8-13, mipmap.h:190-200,152-171 (level 0, Repeat)
    Float st0 = t.su * u + t.du, st1 = t.sv * v + t.dv;
    Float s = st0 * t.width - 0.5f, tt = st1 * t.height - 0.5f;
    int s0 = (int)std::floor(s), t0 = (int)std::floor(tt);
    Float ds = s - s0, dt = tt - t0;
    auto texel = [&](int a, int b) {
        a = ModI(a, t.width); b = ModI(b, t.height);
        const float *px = t.texels + 3 * ((size_t)b * t.width + a);
        return rgb(px[0], px[1], px[2]);
    };
    return (1 - ds) * (1 - dt) * texel(s0, t0) + (1 - ds) * dt * texel(s0, t0 + 1) +
           ds * (1 - dt) * texel(s0 + 1, t0) + ds * dt * texel(s0 + 1, t0 + 1);
}

// ------------------------------------------------------------------- BSDF --
enum { BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16, BSDF_ALL = 31 };
enum { LOBE_LAMBERT, LOBE_SPEC_REFL, LOBE_SPEC_TRANS, LOBE_MICRO_CONDUCTOR, LOBE_MICRO_DIELECTRIC, LOBE_METALROUGH, LOBE_MICRO_TRANS };

inline Float CosTheta(V3 w) { return w.z; }
inline Float Cos2Theta(V3 w) { return w.z * w.z; }
inline Float AbsCosTheta(V3 w) { return std::abs(w.z); }
inline Float Sin2Theta(V3 w) { return std::max((Float)0, (Float)1 - Cos2Theta(w)); }
inline Float SinTheta(V3 w) { return std::sqrt(Sin2Theta(w)); }
inline Float TanTheta(V3 w) { return SinTheta(w) / CosTheta(w); }
inline Float Tan2Theta(V3 w) { return Sin2Theta(w) / Cos2Theta(w); }
inline Float CosPhi(V3 w) { Float s = SinTheta(w); return (s == 0) ? 1 : Clamp(w.x / s, -1, 1); }
inline Float SinPhi(V3 w) { Float s = SinTheta(w); return (s == 0) ? 0 : Clamp(w.y / s, -1, 1); }
inline Float Cos2Phi(V3 w) { return CosPhi(w) * CosPhi(w); }
inline Float Sin2Phi(V3 w) { return SinPhi(w) * SinPhi(w); }
inline bool SameHemisphere(V3 w, V3 wp) { return w.z * wp.z > 0; }
inline V3 Reflect(V3 wo, V3 n) { return -wo + 2 * Dot(wo, n) * n; }
inline bool Refract(V3 wi, V3 n, Float eta, V3 *wt) {  // reflection.h:52-61
    Float cosThetaI = Dot(n, wi);
    Float sin2ThetaI = std::max(Float(0), Float(1 - cosThetaI * cosThetaI));
    Float sin2ThetaT = eta * eta * sin2ThetaI;
    if (sin2ThetaT >= 1) return false;
    Float cosThetaT = std::sqrt(1 - sin2ThetaT);
    *wt = eta * -wi + (eta * cosThetaI - cosThetaT) * n;
    return true;
}

// reflection.cpp:6-21
Float FrDielectric(Float cosThetaI, Float etaI, Float etaT) {
    cosThetaI = Clamp(cosThetaI, -1, 1);
    bool entering = cosThetaI > 0.f;
    if (!entering) { std::swap(etaI, etaT); cosThetaI = std::abs(cosThetaI); }
    Float sinThetaI = std::sqrt(std::max((Float)0, 1 - cosThetaI * cosThetaI));
    Float sinThetaT = etaI / etaT * sinThetaI;
    if (sinThetaT >= 1) return 1;
    Float cosThetaT = std::sqrt(std::max((Float)0, 1 - sinThetaT * sinThetaT));
    Float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    Float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return (Rparl * Rparl + Rperp * Rperp) / 2;
}
// reflection.cpp:23-42
RGB FrConductor(Float cosThetaI, RGB etai, RGB etat, RGB k) {
    cosThetaI = Clamp(cosThetaI, -1, 1);
    RGB eta = etat / etai, etak = k / etai;
    Float cosThetaI2 = cosThetaI * cosThetaI;
    Float sinThetaI2 = 1. - cosThetaI2;
    RGB eta2 = eta * eta, etak2 = etak * etak;
    RGB t0 = eta2 - etak2 - rgb1(sinThetaI2);
    RGB a2plusb2 = SqrtRGB(t0 * t0 + 4 * eta2 * etak2);
    RGB t1 = a2plusb2 + rgb1(cosThetaI2);
    RGB a = SqrtRGB(0.5f * (a2plusb2 + t0));
    RGB t2 = (Float)2 * cosThetaI * a;
    RGB Rs = (t1 - t2) / (t1 + t2);
    RGB t3 = cosThetaI2 * a2plusb2 + rgb1(sinThetaI2 * sinThetaI2);
    RGB t4 = t2 * sinThetaI2;
    RGB Rp = Rs * (t3 - t4) / (t3 + t4);
    return 0.5 * (Rp + Rs);
}

// TrowbridgeReitzDistribution, sampleVisibleArea = true (microfacet.h:59-63)
struct TR {
    Float alphax, alphay;
    Float D(V3 wh) const {  // microfacet.cpp:84-90
        Float tan2Theta = Tan2Theta(wh);
        if (std::isinf(tan2Theta)) return 0.;
        const Float cos4Theta = Cos2Theta(wh) * Cos2Theta(wh);
        Float e = (Cos2Phi(wh) / (alphax * alphax) + Sin2Phi(wh) / (alphay * alphay)) * tan2Theta;
        return 1 / (Pi * alphax * alphay * cos4Theta * (1 + e) * (1 + e));
    }
    Float Lambda(V3 w) const {  // microfacet.cpp:101-107
        Float absTanTheta = std::abs(TanTheta(w));
        if (std::isinf(absTanTheta)) return 0.;
        Float alpha = std::sqrt(Cos2Phi(w) * alphax * alphax + Sin2Phi(w) * alphay * alphay);
        Float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
        return (-1 + std::sqrt(1.f + alpha2Tan2Theta)) / 2;
    }
    Float G1(V3 w) const { return 1 / (1 + Lambda(w)); }
    Float G(V3 wo, V3 wi) const { return 1 / (1 + Lambda(wo) + Lambda(wi)); }
    Float Pdf(V3 wo, V3 wh) const { return D(wh) * G1(wo) * AbsDot(wo, wh) / AbsCosTheta(wo); }  // microfacet.cpp:230-236
    static void Sample11(Float cosTheta, Float U1, Float U2, Float *slope_x, Float *slope_y) {  // microfacet.cpp:149-184
        if (cosTheta > .9999) {
            Float r = std::sqrt(U1 / (1 - U1));
            Float phi = 6.28318530718 * U2;
            *slope_x = r * std::cos(phi);
            *slope_y = r * std::sin(phi);
            return;
        }
        Float sinTheta = std::sqrt(std::max((Float)0, (Float)1 - cosTheta * cosTheta));
        Float tanTheta = sinTheta / cosTheta;
        Float a = 1 / tanTheta;
        Float G1 = 2 / (1 + std::sqrt(1.f + 1.f / (a * a)));
        Float A = 2 * U1 / G1 - 1;
        Float tmp = 1.f / (A * A - 1.f);
        if (tmp > 1e10) tmp = 1e10;
        Float B = tanTheta;
        Float D = std::sqrt(std::max(Float(B * B * tmp * tmp - (A * A - B * B) * tmp), Float(0)));
        Float slope_x_1 = B * tmp - D;
        Float slope_x_2 = B * tmp + D;
        *slope_x = (A < 0 || slope_x_2 > 1.f / tanTheta) ? slope_x_1 : slope_x_2;
        Float S;
        if (U2 > 0.5f) { S = 1.f; U2 = 2.f * (U2 - .5f); }
        else { S = -1.f; U2 = 2.f * (.5f - U2); }
        Float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) /
                  (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
        *slope_y = S * z * std::sqrt(1.f + *slope_x * *slope_x);
    }
    V3 Sample_wh(V3 wo, Float u0, Float u1) const {  // microfacet.cpp:186-228
        bool flip = wo.z < 0;
        V3 wi = flip ? -wo : wo;
        V3 wiStretched = Normalize(mk(alphax * wi.x, alphay * wi.y, wi.z));
        Float slope_x, slope_y;
        Sample11(CosTheta(wiStretched), u0, u1, &slope_x, &slope_y);
        Float tmp = CosPhi(wiStretched) * slope_x - SinPhi(wiStretched) * slope_y;
        slope_y = SinPhi(wiStretched) * slope_x + CosPhi(wiStretched) * slope_y;
        slope_x = tmp;
        slope_x = alphax * slope_x;
        slope_y = alphay * slope_y;
        V3 wh = Normalize(mk(-slope_x, -slope_y, 1.));
        if (flip) wh = -wh;
        return wh;
    }
};
inline Float RoughnessToAlpha(Float roughness) {  // microfacet.h:74-78
    roughness = std::max(roughness, (Float)1e-3);
    Float x = std::log(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}

struct Lobe {
    int kind, type;
    RGB R;            // Lambert R / specular R / T / microfacet R / metal-rough base colour
    RGB eta, k;       // conductor
    Float etaA, etaB; // dielectric
    Float metallic;
    TR distrib;
};

// sampling.cpp:73-86, sampling.h:102-106
inline void ConcentricSampleDisk(Float u0, Float u1, Float *dx, Float *dy) {
    Float ox = 2.f * u0 - 1, oy = 2.f * u1 - 1;
    if (ox == 0 && oy == 0) { *dx = 0; *dy = 0; return; }
    Float theta, r;
    if (std::abs(ox) > std::abs(oy)) { r = ox; theta = PiOver4 * (oy / ox); }
    else { r = oy; theta = PiOver2 - PiOver4 * (ox / oy); }
    *dx = r * std::cos(theta); *dy = r * std::sin(theta);
}
inline V3 CosineSampleHemisphere(Float u0, Float u1) {
    Float dx, dy; ConcentricSampleDisk(u0, u1, &dx, &dy);
    Float z = std::sqrt(std::max((Float)0, 1 - dx * dx - dy * dy));
    return mk(dx, dy, z);
}

RGB LobeF(const Lobe &l, V3 wo, V3 wi) {
    switch (l.kind) {
    case LOBE_LAMBERT: return l.R * InvPi;  // reflection.cpp:135-137
    case LOBE_MICRO_CONDUCTOR:
    case LOBE_MICRO_DIELECTRIC: {           // reflection.cpp:241-249
        Float cosThetaO = AbsCosTheta(wo), cosThetaI = AbsCosTheta(wi);
        V3 wh = wi + wo;
        if (cosThetaI == 0 || cosThetaO == 0) return rgb1(0.);
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) return rgb1(0.);
        wh = Normalize(wh);
        Float c = Dot(wi, Faceforward(wh, mk(0, 0, 1)));
        RGB F = (l.kind == LOBE_MICRO_CONDUCTOR) ? FrConductor(std::abs(c), rgb1(1.), l.eta, l.k)
                                                 : rgb1(FrDielectric(c, l.etaA, l.etaB));
        return l.R * l.distrib.D(wh) * l.distrib.G(wo, wi) * F / (4 * cosThetaI * cosThetaO);
    }
    case LOBE_METALROUGH: {                 // reflection.cpp:268-284
        Float cosThetaO = AbsCosTheta(wo), cosThetaI = AbsCosTheta(wi);
        V3 wh = wi + wo;
        if (cosThetaI == 0 || cosThetaO == 0) return rgb1(0.);
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) return rgb1(0.);
        wh = Normalize(wh);
        Float cs = std::abs(Dot(wi, Faceforward(wh, mk(0, 0, 1))));
        RGB F0 = rgb1(0.04);
        F0 = (1 - l.metallic) * F0 + l.metallic * l.R;
        RGB F = F0 + (rgb1(1.0) - F0) * (Float)std::pow(1.0 - cs, 5.0);
        RGB specular = l.distrib.D(wh) * l.distrib.G(wo, wi) * F / (Float)(4 * cosThetaI * cosThetaO + 0.001);
        RGB kS = F;
        RGB kD = rgb1(1.0) - kS;
        kD = kD * (Float)(1.0 - l.metallic);
        return specular + kD * l.R * InvPi;
    }
    case LOBE_MICRO_TRANS: {                // MicrofacetTransmission::f, reflection.cpp:311-334 (TransportMode::Radiance)
        if (SameHemisphere(wo, wi)) return rgb1(0.f);
        Float cosThetaO = CosTheta(wo), cosThetaI = CosTheta(wi);
        if (cosThetaI == 0 || cosThetaO == 0) return rgb1(0.f);
        Float eta = CosTheta(wo) > 0 ? (l.etaB / l.etaA) : (l.etaA / l.etaB);
        V3 wh = Normalize(wo + wi * eta);
        if (wh.z < 0) wh = -wh;
        if (Dot(wo, wh) * Dot(wi, wh) > 0) return rgb1(0.f);
        RGB F = rgb1(FrDielectric(Dot(wo, wh), l.etaA, l.etaB));
        Float sqrtDenom = Dot(wo, wh) + eta * Dot(wi, wh);
        Float factor = 1 / eta;
        return (rgb1(1.f) - F) * l.R *
               std::abs(l.distrib.D(wh) * l.distrib.G(wo, wi) * eta * eta * AbsDot(wi, wh) * AbsDot(wo, wh) * factor * factor /
                        (cosThetaI * cosThetaO * sqrtDenom * sqrtDenom));
    }
    default: return rgb1(0.f);              // specular lobes: reflection.h:203-205,226-228
    }
}
Float LobePdf(const Lobe &l, V3 wo, V3 wi) {
    switch (l.kind) {
    case LOBE_MICRO_TRANS: {                // MicrofacetTransmission::Pdf, reflection.cpp:346-355
        if (SameHemisphere(wo, wi)) return 0;
        Float eta = CosTheta(wo) > 0 ? (l.etaB / l.etaA) : (l.etaA / l.etaB);
        V3 wh = Normalize(wo + wi * eta);
        if (Dot(wo, wh) * Dot(wi, wh) > 0) return 0;
        Float sqrtDenom = Dot(wo, wh) + eta * Dot(wi, wh);
        Float dwh_dwi = std::abs((eta * eta * Dot(wi, wh)) / (sqrtDenom * sqrtDenom));
        return l.distrib.Pdf(wo, wh) * dwh_dwi;
    }
    case LOBE_LAMBERT: return SameHemisphere(wo, wi) ? AbsCosTheta(wi) * InvPi : 0;  // reflection.cpp:107-109
    case LOBE_MICRO_CONDUCTOR: case LOBE_MICRO_DIELECTRIC: case LOBE_METALROUGH: {    // reflection.cpp:262-266,296-300
        if (!SameHemisphere(wo, wi)) return 0;
        V3 wh = Normalize(wo + wi);
        return l.distrib.Pdf(wo, wh) / (4 * Dot(wo, wh));
    }
    default: return 0;
    }
}
RGB LobeSample(const Lobe &l, V3 wo, V3 *wi, Float u0, Float u1, Float *pdf) {
    switch (l.kind) {
    case LOBE_LAMBERT: {                      // reflection.cpp:99-105
        *wi = CosineSampleHemisphere(u0, u1);
        if (wo.z < 0) wi->z *= -1;
        *pdf = LobePdf(l, wo, *wi);
        return LobeF(l, wo, *wi);
    }
    case LOBE_SPEC_REFL: {                    // reflection.cpp:62-68 with FresnelNoOp
        *wi = mk(-wo.x, -wo.y, wo.z);
        *pdf = 1;
        return rgb1(1.) * l.R / AbsCosTheta(*wi);
    }
    case LOBE_SPEC_TRANS: {                   // reflection.cpp:75-88, TransportMode::Radiance
        bool entering = CosTheta(wo) > 0;
        Float etaI = entering ? l.etaA : l.etaB;
        Float etaT = entering ? l.etaB : l.etaA;
        if (!Refract(wo, Faceforward(mk(0, 0, 1), wo), etaI / etaT, wi)) return rgb1(0);
        *pdf = 1;
        RGB ft = l.R * (rgb1(1.) - rgb1(FrDielectric(CosTheta(*wi), l.etaA, l.etaB)));
        ft = ft * ((etaI * etaI) / (etaT * etaT));
        return ft / AbsCosTheta(*wi);
    }
    case LOBE_MICRO_TRANS: {                  // MicrofacetTransmission::Sample_f, reflection.cpp:336-344
        if (wo.z == 0) return rgb1(0.);
        V3 wh = l.distrib.Sample_wh(wo, u0, u1);
        if (Dot(wo, wh) < 0) return rgb1(0.);
        Float eta = CosTheta(wo) > 0 ? (l.etaA / l.etaB) : (l.etaB / l.etaA);
        if (!Refract(wo, wh, eta, wi)) return rgb1(0.);
        *pdf = LobePdf(l, wo, *wi);
        return LobeF(l, wo, *wi);
    }
    default: {                                // reflection.cpp:251-260,286-294
        if (wo.z == 0) return rgb1(0.);
        V3 wh = l.distrib.Sample_wh(wo, u0, u1);
        if (Dot(wo, wh) < 0) return rgb1(0.);
        *wi = Reflect(wo, wh);
        if (!SameHemisphere(wo, *wi)) return rgb1(0.f);
        *pdf = l.distrib.Pdf(wo, wh) / (4 * Dot(wo, wh));
        return LobeF(l, wo, *wi);
    }
    }
}

struct BSDF {
    V3 ns, ng, ss, ts; int n; Lobe lobes[2]; Float eta;
    void Init(const SurfaceInteraction &si) {  // reflection.h:139-141
        ns = si.n; ng = si.n; ss = Normalize(si.dpdu); ts = Cross(ns, ss); n = 0; eta = 1;
    }
    V3 WorldToLocal(V3 v) const { return mk(Dot(v, ss), Dot(v, ts), Dot(v, ns)); }
    V3 LocalToWorld(V3 v) const {
        return mk(ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z, ss.z * v.x + ts.z * v.y + ns.z * v.z);
    }
    static bool Matches(int type, int flags) { return (type & flags) == type; }
    int NumComponents(int flags) const { int c = 0; for (int i = 0; i < n; ++i) if (Matches(lobes[i].type, flags)) ++c; return c; }
    RGB f(V3 woW, V3 wiW, int flags) const {  // reflection.cpp:140-152
        V3 wi = WorldToLocal(wiW), wo = WorldToLocal(woW);
        if (wo.z == 0) return rgb1(0.);
        bool reflect = Dot(wiW, ng) * Dot(woW, ng) > 0;
        RGB r = rgb1(0.f);
        for (int i = 0; i < n; ++i)
            if (Matches(lobes[i].type, flags) &&
                ((reflect && (lobes[i].type & BSDF_REFLECTION)) || (!reflect && (lobes[i].type & BSDF_TRANSMISSION))))
                r = r + LobeF(lobes[i], wo, wi);
        return r;
    }
    RGB Sample_f(V3 woWorld, V3 *wiWorld, Float u0, Float u1, Float *pdf, int type, int *sampledType) const {  // reflection.cpp:173-217
        int matchingComps = NumComponents(type);
        if (matchingComps == 0) { *pdf = 0; *sampledType = 0; return rgb1(0); }
        int comp = std::min((int)std::floor(u0 * matchingComps), matchingComps - 1);
        const Lobe *bxdf = nullptr; int count = comp;
        for (int i = 0; i < n; ++i) if (Matches(lobes[i].type, type) && count-- == 0) { bxdf = &lobes[i]; break; }
        Float ur0 = std::min(u0 * matchingComps - comp, OneMinusEpsilon), ur1 = u1;
        V3 wi, wo = WorldToLocal(woWorld);
        if (wo.z == 0) return rgb1(0.);
        *pdf = 0;
        *sampledType = bxdf->type;
        RGB f = LobeSample(*bxdf, wo, &wi, ur0, ur1, pdf);
        if (*pdf == 0) { *sampledType = 0; return rgb1(0); }
        *wiWorld = LocalToWorld(wi);
        if (!(bxdf->type & BSDF_SPECULAR) && matchingComps > 1)
            for (int i = 0; i < n; ++i) if (&lobes[i] != bxdf && Matches(lobes[i].type, type)) *pdf += LobePdf(lobes[i], wo, wi);
        if (matchingComps > 1) *pdf /= matchingComps;
        if (!(bxdf->type & BSDF_SPECULAR)) {
            bool reflect = Dot(*wiWorld, ng) * Dot(woWorld, ng) > 0;
            f = rgb1(0.);
            for (int i = 0; i < n; ++i)
                if (Matches(lobes[i].type, type) &&
                    ((reflect && (lobes[i].type & BSDF_REFLECTION)) || (!reflect && (lobes[i].type & BSDF_TRANSMISSION))))
                    f = f + LobeF(lobes[i], wo, wi);
        }
        return f;
    }
};

inline TR MakeTR(Float ax, Float ay) { TR d; d.alphax = std::max(Float(0.001), ax); d.alphay = std::max(Float(0.001), ay); return d; }

// material.cpp:5-82
void ComputeScatteringFunctions(const Scene &sc, const SurfaceInteraction &si, BSDF *b) {
    b->Init(si);
    const rt_material &m = sc.materials[sc.prim_material[si.prim]];
    auto tex = [&](int i) { return EvalTexture(sc, m.tex[i], si.u, si.v); };
    Lobe l; memset(&l, 0, sizeof(l));
    switch (m.kind) {
    case RT_MAT_MATTE:
        l.kind = LOBE_LAMBERT; l.type = BSDF_REFLECTION | BSDF_DIFFUSE; l.R = ClampRGB(tex(0)); b->lobes[b->n++] = l; break;
    case RT_MAT_MIRROR: {
        RGB R = ClampRGB(tex(0));
        if (!IsBlack(R)) { l.kind = LOBE_SPEC_REFL; l.type = BSDF_REFLECTION | BSDF_SPECULAR; l.R = R; b->lobes[b->n++] = l; }
        break;
    }
    case RT_MAT_GLASS: {
        Float eta = tex(2)[0];
        RGB R = ClampRGB(tex(0)), T = ClampRGB(tex(1));
        b->eta = eta;
        if (!IsBlack(R) && !IsBlack(T)) {
            l.kind = LOBE_SPEC_TRANS; l.type = BSDF_TRANSMISSION | BSDF_SPECULAR; l.R = T; l.etaA = 1.f; l.etaB = eta; b->lobes[b->n++] = l;
        }
        break;
    }
    case RT_MAT_PLASTIC: {
        RGB kd = ClampRGB(tex(0));
        if (!IsBlack(kd)) { Lobe d = l; d.kind = LOBE_LAMBERT; d.type = BSDF_REFLECTION | BSDF_DIFFUSE; d.R = kd; b->lobes[b->n++] = d; }
        RGB ks = ClampRGB(tex(1));
        if (!IsBlack(ks)) {
            Float rough = tex(2)[0];
            if (m.remap_roughness) rough = RoughnessToAlpha(rough);
            l.kind = LOBE_MICRO_DIELECTRIC; l.type = BSDF_REFLECTION | BSDF_GLOSSY; l.R = ks; l.etaA = 1.5f; l.etaB = 1.f;
            l.distrib = MakeTR(rough, rough); b->lobes[b->n++] = l;
        }
        break;
    }
    case RT_MAT_METAL: {
        Float uRough = tex(2)[0], vRough = tex(3)[0];
        if (m.remap_roughness) { uRough = RoughnessToAlpha(uRough); vRough = RoughnessToAlpha(vRough); }
        l.kind = LOBE_MICRO_CONDUCTOR; l.type = BSDF_REFLECTION | BSDF_GLOSSY; l.R = rgb1(1.);
        l.eta = tex(0); l.k = tex(1); l.distrib = MakeTR(uRough, vRough); b->lobes[b->n++] = l;
        break;
    }
    case RT_MAT_ROUGHGLASS: {      // no material of the reference: its two rough dielectric lobes, assembled as pbrt's glass does
        Float eta = tex(2)[0], rough = tex(3)[0];
        RGB R = ClampRGB(tex(0)), T = ClampRGB(tex(1));
        b->eta = eta;
        if (IsBlack(R) && IsBlack(T)) break;
        if (m.remap_roughness) rough = RoughnessToAlpha(rough);
        if (!IsBlack(R)) {
            Lobe r = l; r.kind = LOBE_MICRO_DIELECTRIC; r.type = BSDF_REFLECTION | BSDF_GLOSSY; r.R = R; r.etaA = 1.f; r.etaB = eta;
            r.distrib = MakeTR(rough, rough); b->lobes[b->n++] = r;
        }
        if (!IsBlack(T)) {
            l.kind = LOBE_MICRO_TRANS; l.type = BSDF_TRANSMISSION | BSDF_GLOSSY; l.R = T; l.etaA = 1.f; l.etaB = eta;
            l.distrib = MakeTR(rough, rough); b->lobes[b->n++] = l;
        }
        break;
    }
    case RT_MAT_METALROUGH: {
        RGB c = ClampRGB(tex(0));
        if (m.tex[1] < 0) { l.kind = LOBE_LAMBERT; l.type = BSDF_REFLECTION | BSDF_DIFFUSE; l.R = c; b->lobes[b->n++] = l; }
        else {
            RGB tmp = ClampRGB(tex(1));
            Float metallic = tmp[2], roughness = std::max(0.05f, tmp[1]);
            l.kind = LOBE_METALROUGH; l.type = BSDF_REFLECTION | BSDF_GLOSSY; l.R = c; l.metallic = metallic;
            l.distrib = MakeTR(roughness, roughness); b->lobes[b->n++] = l;
        }
        break;
    }
    }
}

// ---------------------------------------------------------------- samplers --
struct PCG32 {  // rng.h:22-100
    uint64_t state, inc;
    PCG32() : state(0x853c49e6748fea9bULL), inc(0xda3e39cb94b95bdbULL) {}
    void SetSequence(uint64_t initseq) { state = 0u; inc = (initseq << 1u) | 1u; UniformUInt32(); state += 0x853c49e6748fea9bULL; UniformUInt32(); }
    uint32_t UniformUInt32() {
        uint64_t oldstate = state;
        state = oldstate * 0x5851f42d4c957f2dULL + inc;
        uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
        uint32_t rot = (uint32_t)(oldstate >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    uint32_t UniformUInt32(uint32_t b) {
        uint32_t threshold = (~b + 1u) % b;
        while (true) { uint32_t r = UniformUInt32(); if (r >= threshold) return r % b; }
    }
    Float UniformFloat() { return std::min(OneMinusEpsilon, Float(UniformUInt32() * 2.3283064365386963e-10f)); }
};

// Philox4x32-10 (Salmon et al., SC'11): counter (c0..c3), key (k0,k1).
inline void Philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
inline Float U32ToFloat(uint32_t v) { return std::min(OneMinusEpsilon, Float(v * 2.3283064365386963e-10f)); }
// Stateless permutation of {0..l-1} (Kensler, "Correlated Multi-Jittered Sampling", 2013).
inline uint32_t KenslerPermute(uint32_t i, uint32_t l, uint32_t p) {
    uint32_t w = l - 1;
    w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
    do {
        i ^= p; i *= 0xe170893d; i ^= p >> 16; i ^= (i & w) >> 4; i ^= p >> 8; i *= 0x0929eb3f; i ^= p >> 23;
        i ^= (i & w) >> 1; i *= 1 | p >> 27; i *= 0x6935fa69; i ^= (i & w) >> 11; i *= 0x74dcb303;
        i ^= (i & w) >> 2; i *= 0x9e501cc3; i ^= (i & w) >> 2; i *= 0xc860a3df; i &= w; i ^= i >> 5;
    } while (i >= l);
    return (i + p) % l;
}

// Sampler interface used by the estimator.  Two implementations:
//  * RefSampler: StratifiedSampler(nx, ny, true, 1) + PCG32, stream-exact with the reference;
//  * PhiloxSampler: the counter-based scheme the HIP kernels use, keyed (pixel, sample, bounce).
struct Sampler {
    virtual ~Sampler() {}
    virtual Float Get1D() = 0;
    virtual void Get2D(Float *a, Float *b) = 0;
    virtual void BeginBounce(int depth) {}
    virtual void LightDims(Float *pick, Float *ul0, Float *ul1) { *pick = Get1D(); Get2D(ul0, ul1); Float a, b; Get2D(&a, &b); }
    virtual void BsdfDims(Float *u0, Float *u1) { Get2D(u0, u1); }
    virtual Float RRDim() { return Get1D(); }
    // UniformSampleAllLights with no sample arrays requested (integrator.cpp:35-39): uLight, then uScattering (which EstimateDirect never reads)
    virtual void AllLightDims(int light, Float *ul0, Float *ul1) { Get2D(ul0, ul1); Float a, b; Get2D(&a, &b); }
};

struct RefSampler : Sampler {
    int nx, ny; int64_t spp; PCG32 rng; std::vector<Float> s1; std::vector<Float> s2; int64_t idx; int d1, d2;
    RefSampler(int nx_, int ny_, uint64_t seed) : nx(nx_), ny(ny_), spp((int64_t)nx_ * ny_), s1(spp), s2(2 * spp) { rng.SetSequence(seed); }
    void StartPixel() {  // stratified.cpp:4-29 with one sampled dimension
        int n = (int)spp;
        Float invN = (Float)1 / n;
        for (int i = 0; i < n; ++i) { Float delta = rng.UniformFloat(); s1[i] = std::min((i + delta) * invN, OneMinusEpsilon); }
        for (int i = 0; i < n; ++i) { int other = i + rng.UniformUInt32(n - i); std::swap(s1[i], s1[other]); }
        Float dx = (Float)1 / nx, dy = (Float)1 / ny; int k = 0;
        for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
            Float jx = rng.UniformFloat(); Float jy = rng.UniformFloat();
            s2[2 * k] = std::min((x + jx) * dx, OneMinusEpsilon); s2[2 * k + 1] = std::min((y + jy) * dy, OneMinusEpsilon); ++k;
        }
        for (int i = 0; i < n; ++i) { int other = i + rng.UniformUInt32(n - i); std::swap(s2[2 * i], s2[2 * other]); std::swap(s2[2 * i + 1], s2[2 * other + 1]); }
        idx = 0; d1 = d2 = 0;
    }
    bool StartNextSample() { d1 = d2 = 0; return ++idx < spp; }
    Float Get1D() override { if (d1 < 1) { d1++; return s1[idx]; } return rng.UniformFloat(); }
    void Get2D(Float *a, Float *b) override {
        if (d2 < 1) { d2++; *a = s2[2 * idx]; *b = s2[2 * idx + 1]; return; }
        *a = rng.UniformFloat(); *b = rng.UniformFloat();
    }
};

struct PhiloxSampler : Sampler {
    uint32_t k0, k1, pixel, sample; int depth; uint32_t blk0[4], blk1[4]; bool have0, have1;
    void Begin(uint32_t pix, uint32_t smp) { pixel = pix; sample = smp; depth = 0; have0 = have1 = false; }
    void BeginBounce(int d) override { depth = d; have0 = have1 = false; }
    void need0() { if (!have0) { blk0[0] = pixel; blk0[1] = sample; blk0[2] = (uint32_t)depth; blk0[3] = 0; Philox4x32(blk0, k0, k1); have0 = true; } }
    void need1() { if (!have1) { blk1[0] = pixel; blk1[1] = sample; blk1[2] = (uint32_t)depth; blk1[3] = 1; Philox4x32(blk1, k0, k1); have1 = true; } }
    Float Get1D() override { return 0; }
    void Get2D(Float *a, Float *b) override { *a = *b = 0; }
    // block 0 = {light pick, uLight.x, uLight.y, roulette}; block 1 = {bsdf.x, bsdf.y, -, -}
    void LightDims(Float *pick, Float *ul0, Float *ul1) override { need0(); *pick = U32ToFloat(blk0[0]); *ul0 = U32ToFloat(blk0[1]); *ul1 = U32ToFloat(blk0[2]); }
    void BsdfDims(Float *u0, Float *u1) override { need1(); *u0 = U32ToFloat(blk1[0]); *u1 = U32ToFloat(blk1[1]); }
    Float RRDim() override { need0(); return U32ToFloat(blk0[3]); }
    // block 2 + j = {uLight.x, uLight.y, -, -} of light j when every light is sampled
    void AllLightDims(int light, Float *ul0, Float *ul1) override {
        uint32_t c[4] = {pixel, sample, (uint32_t)depth, 2u + (uint32_t)light}; Philox4x32(c, k0, k1);
        *ul0 = U32ToFloat(c[0]); *ul1 = U32ToFloat(c[1]);
    }
    // film sample: counter (pixel, sample, 0xFFFFFFFF, 0); stratum = Kensler permutation of the sample index
    void Film(int nx, int ny, Float *fx, Float *fy) {
        uint32_t c[4] = {pixel, sample, 0xFFFFFFFFu, 0}; Philox4x32(c, k0, k1);
        uint32_t n = (uint32_t)(nx * ny);
        uint32_t ph[4] = {pixel, 0xFFFFFFFFu, 0xFFFFFFFFu, 0}; Philox4x32(ph, k0, k1);
        uint32_t s = KenslerPermute(sample % n, n, ph[0]);
        int sx = (int)(s % (uint32_t)nx), sy = (int)(s / (uint32_t)nx);
        Float dx = (Float)1 / nx, dy = (Float)1 / ny;
        *fx = std::min((sx + U32ToFloat(c[0])) * dx, OneMinusEpsilon);
        *fy = std::min((sy + U32ToFloat(c[1])) * dy, OneMinusEpsilon);
    }
};

// ------------------------------------------------------------- estimator --
struct PathCounters { uint64_t closest, shadow; };

// light.h:73-75
inline RGB LightL(const rt_light &l, V3 n, V3 w) {
    return (l.two_sided || Dot(n, w) > 0) ? rgb(l.Le[0], l.Le[1], l.Le[2]) : rgb1(0.f);
}

// pbrt.h:213-226
template <typename Pred> inline int FindInterval(int size, const Pred &pred) {
    int first = 0, len = size;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (pred(middle)) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    return Clamp(first - 1, 0, size - 2);
}
// Distribution1D::SampleContinuous (sampling.h:30-40) over func[0..n), cdf[0..n], funcInt
inline Float SampleContinuous1D(const float *func, const float *cdf, int n, Float funcInt, Float u, Float *pdf, int *off) {
    int offset = FindInterval(n + 1, [&](int index) { return cdf[index] <= u; });
    if (off) *off = offset;
    Float du = u - cdf[offset];
    if ((cdf[offset + 1] - cdf[offset]) > 0) du /= (cdf[offset + 1] - cdf[offset]);
    if (pdf) *pdf = (funcInt > 0) ? func[offset] / funcInt : 0;
    return (offset + du) / n;
}
// MIPMap::Lookup(st) for a zero filter width = triangle(0, st) with the Repeat wrap (mipmap.h:173-200)
RGB EnvLookup(const rt_env_light &e, Float u, Float v) {
    Float s = u * e.width - 0.5f, t = v * e.height - 0.5f;
    int s0 = (int)std::floor(s), t0 = (int)std::floor(t);
    Float ds = s - s0, dt = t - t0;
    auto texel = [&](int a, int b) { const float *px = e.texels + 3 * ((size_t)ModI(b, e.height) * e.width + ModI(a, e.width)); return rgb(px[0], px[1], px[2]); };
    return (1 - ds) * (1 - dt) * texel(s0, t0) + (1 - ds) * dt * texel(s0, t0 + 1) + ds * (1 - dt) * texel(s0 + 1, t0) + ds * dt * texel(s0 + 1, t0 + 1);
}
// InfiniteAreaLight::Sample_Li (infinite.cpp:83-106) with Distribution2D::SampleContinuous (sampling.h:72-79) and
// MIPMap::Lookup(st) = triangle(0, st) for a zero filter width (mipmap.h:173-200).  Returns Li; *p1 = the far end of the
// visibility segment.
RGB EnvSampleLi(const rt_env_light &e, V3 refP, Float u0, Float u1, V3 *wi, Float *pdf, V3 *p1) {
    Float pdfs[2]; int v;
    Float d1 = SampleContinuous1D(e.func_int, e.marginal_cdf, e.nv, e.marginal_int, u1, &pdfs[1], &v);
    Float d0 = SampleContinuous1D(e.func + (size_t)v * e.nu, e.cdf + (size_t)v * (e.nu + 1), e.nu, e.func_int[v], u0, &pdfs[0], nullptr);
    Float mapPdf = pdfs[0] * pdfs[1];
    if (mapPdf == 0) return rgb1(0.f);
    Float theta = d1 * Pi, phi = d0 * 2 * Pi;
    Float cosTheta = std::cos(theta), sinTheta = std::sin(theta);
    Float sinPhi = std::sin(phi), cosPhi = std::cos(phi);
    *wi = XfVector(e.light_to_world, mk(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta));
    *pdf = mapPdf / (2 * Pi * Pi * sinTheta);
    if (sinTheta == 0) *pdf = 0;
    *p1 = refP + *wi * (2 * e.world_radius);
    return EnvLookup(e, d0, d1);
}

// InfiniteAreaLight::Le (infinite.cpp:77-81): the map looked up at the direction's spherical angles, zero filter width
RGB EnvLe(const rt_env_light &e, V3 d) {
    V3 w = Normalize(XfVector(e.world_to_light, d));
    Float p = std::atan2(w.y, w.x);
    Float phi = (p < 0) ? (p + 2 * Pi) : p;                       // SphericalPhi, geometry.h:939-942
    Float theta = std::acos(Clamp(w.z, -1, 1));                   // SphericalTheta, geometry.h:935-937
    return EnvLookup(e, phi * Inv2Pi, theta * InvPi);
}

// EstimateDirect without multiple importance sampling (integrator.cpp:74-102) for one light: light.cpp:43-54 + shape.cpp:20-31 +
// triangle.cpp:229-248 + interaction.h:27-32 for a triangle light, infinite.cpp:83-106 for an infinite one
RGB EstimateDirect(const Scene &sc, const SurfaceInteraction &si, const BSDF &bsdf, const rt_light &light, Float u0, Float u1, PathCounters *pc) {
    if (light.kind == RT_LIGHT_INFINITE) {      // EstimateDirect (integrator.cpp:74-102) around InfiniteAreaLight::Sample_Li
        V3 wi = mk(0, 0, 0), p1 = mk(0, 0, 0); Float lightPdf = 0;
        RGB Li = EnvSampleLi(sc.envs[light.env], si.p, u0, u1, &wi, &lightPdf, &p1);
        RGB Ld = rgb1(0.f);
        if (lightPdf > 0 && !IsBlack(Li)) {
            RGB f = bsdf.f(si.wo, wi, BSDF_ALL & ~BSDF_SPECULAR) * AbsDot(wi, si.n);
            if (!IsBlack(f)) {
                // VisibilityTester(ref, Interaction(p1)) -> SpawnRayTo (interaction.h:27-32); the far end has no normal, so no offset
                V3 origin = OffsetRayOrigin(si.p, si.n, p1 - si.p);
                V3 target = OffsetRayOrigin(p1, mk(0, 0, 0), origin - p1);
                Ray sr = {origin, target - origin, 1 - ShadowEpsilon};
                if (pc) pc->shadow++;
                if (IntersectAny(sc, sr, nullptr)) Li = rgb1(0.f);
                if (!IsBlack(Li)) Ld = Ld + f * Li / lightPdf;
            }
        }
        return Ld;
    }
    // Triangle::Sample
    size_t g = (size_t)sc.prim_geom[light.prim];
    const float *P = &sc.tri_p[9 * g];
    V3 p0 = mk(P[0], P[1], P[2]), p1 = mk(P[3], P[4], P[5]), p2 = mk(P[6], P[7], P[8]);
    Float su0 = std::sqrt(u0);
    Float b0 = 1 - su0, b1 = u1 * su0;
    V3 lp = b0 * p0 + b1 * p1 + (1 - b0 - b1) * p2;
    V3 ln = Normalize(Cross(p1 - p0, p2 - p0));
    if (!sc.tri_has_n.empty() && sc.tri_has_n[g]) {
        const float *N = &sc.tri_n[9 * g];
        V3 ns = b0 * mk(N[0], N[1], N[2]) + b1 * mk(N[3], N[4], N[5]) + (1 - b0 - b1) * mk(N[6], N[7], N[8]);
        ln = Faceforward(ln, ns);
    } else if (light.flip_normal) ln = -ln;   // reverseOrientation ^ transformSwapsHandedness, triangle.cpp:242-243
    Float area = 0.5 * Length(Cross(p1 - p0, p2 - p0));
    Float pdf = 1 / area;
    // Shape::Sample(ref, u, pdf)
    V3 wi = lp - si.p;
    if (LengthSquared(wi) == 0) pdf = 0;
    else {
        wi = Normalize(wi);
        pdf *= LengthSquared(si.p - lp) / AbsDot(ln, -wi);
        if (std::isinf(pdf)) pdf = 0.f;
    }
    // DiffuseAreaLight::Sample_Li
    RGB Li;
    if (pdf == 0 || LengthSquared(lp - si.p) == 0) { pdf = 0; Li = rgb1(0.f); }
    else { wi = Normalize(lp - si.p); Li = LightL(light, ln, -wi); }
    RGB Ld = rgb1(0.f);
    if (pdf > 0 && !IsBlack(Li)) {
        RGB f = bsdf.f(si.wo, wi, BSDF_ALL & ~BSDF_SPECULAR) * AbsDot(wi, si.n);
        if (!IsBlack(f)) {
            V3 origin = OffsetRayOrigin(si.p, si.n, lp - si.p);
            V3 target = OffsetRayOrigin(lp, ln, origin - lp);
            Ray sr = {origin, target - origin, 1 - ShadowEpsilon};
            if (pc) pc->shadow++;
            if (IntersectAny(sc, sr, nullptr)) Li = rgb1(0.f);
            if (!IsBlack(Li)) Ld = Ld + f * Li / pdf;
        }
    }
    return Ld;
}

// Distribution1D of UniformLightDistribution (lightdistrib.cpp:12-15, sampling.h:16-27): func = 1 for every light
inline void BuildLightCdf(int n, std::vector<float> &cdf, float &funcInt) {
    cdf.assign(n + 1, 0.f);
    for (int i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + Float(1) / n;
    funcInt = cdf[n];
    if (funcInt == 0) { for (int i = 1; i < n + 1; ++i) cdf[i] = Float(i) / Float(n); }
    else for (int i = 1; i < n + 1; ++i) cdf[i] /= funcInt;
}
struct LightDistrib {
    const std::vector<float> &cdf; Float funcInt; int n;
    int SampleDiscrete(Float u, Float *pdf) const {     // sampling.h:40-48
        int offset = FindInterval((int)cdf.size(), [&](int index) { return cdf[index] <= u; });
        *pdf = (funcInt > 0) ? Float(1) / (funcInt * n) : 0;
        return offset;
    }
};

// integrator.cpp:53-72; distrib == nullptr is Li_re's call, a distribution is Li's
RGB UniformSampleOneLight(const Scene &sc, const SurfaceInteraction &si, const BSDF &bsdf, Sampler &smp, PathCounters *pc, const LightDistrib *distrib = nullptr) {
    int nLights = (int)sc.lights.size();
    if (nLights == 0) return rgb1(0.f);
    Float pick, u0, u1;
    smp.LightDims(&pick, &u0, &u1);      // Get1D, Get2D (uLight), Get2D (uScattering): drawn before the early return below, which no light set here reaches
    int lightNum; Float lightPdf;
    if (distrib) {
        lightNum = distrib->SampleDiscrete(pick, &lightPdf);
        if (lightPdf == 0) return rgb1(0.f);
    } else {
        lightNum = std::min((int)(pick * nLights), nLights - 1);
        lightPdf = Float(1) / nLights;
    }
    return EstimateDirect(sc, si, bsdf, sc.lights[lightNum], u0, u1, pc) / lightPdf;
}

// integrator.cpp:26-51 with one sample per light and no sample arrays requested from the sampler (the branch at :35-40)
RGB UniformSampleAllLights(const Scene &sc, const SurfaceInteraction &si, const BSDF &bsdf, Sampler &smp, PathCounters *pc) {
    RGB L = rgb1(0.f);
    for (size_t j = 0; j < sc.lights.size(); ++j) {
        Float u0, u1;
        smp.AllLightDims((int)j, &u0, &u1);
        L = L + EstimateDirect(sc, si, bsdf, sc.lights[j], u0, u1, pc);
    }
    return L;
}

// integrator.cpp:236-281.  `specular` is the per-path replacement of the global perfectspecular.
RGB Li_re(const Scene &sc, const Ray &r, Sampler &smp, int depth, int maxDepth, bool *specular, PathCounters *pc, bool allLights = false) {
    RGB L = rgb1(0.f), beta = rgb1(1.f);
    Ray ray = r;
    Hit hit;
    if (pc) pc->closest++;
    bool found = IntersectClosest(sc, ray, &hit, nullptr);
    SurfaceInteraction si;
    if (found) FillInteraction(sc, ray, hit, &si);
    if (depth == 0 || *specular || sc.lights.size() == 0) {
        if (found) {
            int li = sc.prim_light[hit.prim];
            if (li >= 0) L = L + LightL(sc.lights[li], si.n, -ray.d);
            if (!IsBlack(L)) return L;
        } else if (sc.lights.size() == 0) {
            V3 unit_direction = Normalize(r.d);
            double t = 0.5 * (unit_direction.y + 1.0);
            L = L + ((Float)(1.0 - t) * rgb(1.0, 1.0, 1.0) + (Float)t * rgb(0.5, 0.7, 1.0));
        }
    }
    if (!found || depth >= maxDepth) return L;
    BSDF bsdf;
    ComputeScatteringFunctions(sc, si, &bsdf);
    smp.BeginBounce(depth);
    if (bsdf.NumComponents(BSDF_ALL & ~BSDF_SPECULAR) > 0) {
        RGB Ld = allLights ? UniformSampleAllLights(sc, si, bsdf, smp, pc) : UniformSampleOneLight(sc, si, bsdf, smp, pc);
        L = L + Ld;
    }
    V3 wo = -ray.d, wi; Float pdf; int flags = 0;
    Float u0, u1; smp.BsdfDims(&u0, &u1);
    RGB f = bsdf.Sample_f(wo, &wi, u0, u1, &pdf, BSDF_ALL, &flags);
    if (IsBlack(f) || pdf == 0.f) return L;
    beta = beta * (f * AbsDot(wi, si.n) / pdf);
    *specular = (flags & BSDF_SPECULAR) != 0;
    Ray next = {OffsetRayOrigin(si.p, si.n, wi), wi, Infinity};
    if (depth > 3) {
        Float q = std::max((Float).05, 1 - MaxComp(beta));
        if (smp.RRDim() < q) return L;
        beta = beta / (1 - q);
    }
    return L + beta * Li_re(sc, next, smp, depth + 1, maxDepth, specular, pc, allLights);
}

// PathIntegrator::Li, integrator.cpp:175-229 (every material here makes a BSDF, so the 'no bsdf' skip at :196-200 never runs)
RGB Li(const Scene &sc, const Ray &r, Sampler &smp, int maxDepth, Float rrThreshold, bool allLights, PathCounters *pc) {
    RGB L = rgb1(0.f), beta = rgb1(1.f);
    Ray ray = r;
    bool specularBounce = false;
    Float etaScale = 1;
    const LightDistrib distrib = {sc.light_cdf, sc.light_func_int, (int)sc.lights.size()};     // lightDistribution->Lookup(isect.p): the same for every point
    for (int bounces = 0;; ++bounces) {
        Hit hit;
        if (pc) pc->closest++;
        bool found = IntersectClosest(sc, ray, &hit, nullptr);
        SurfaceInteraction si;
        if (found) FillInteraction(sc, ray, hit, &si);
        if (bounces == 0 || specularBounce) {
            if (found) {
                int li = sc.prim_light[hit.prim];
                if (li >= 0) L = L + beta * LightL(sc.lights[li], si.n, -ray.d);
            } else
                for (const rt_light &l : sc.lights) if (l.kind == RT_LIGHT_INFINITE) L = L + beta * EnvLe(sc.envs[l.env], ray.d);
        }
        if (!found || bounces >= maxDepth) break;
        BSDF bsdf;
        ComputeScatteringFunctions(sc, si, &bsdf);
        smp.BeginBounce(bounces);
        if (bsdf.NumComponents(BSDF_ALL & ~BSDF_SPECULAR) > 0) {
            RGB Ld = beta * (allLights ? UniformSampleAllLights(sc, si, bsdf, smp, pc) : UniformSampleOneLight(sc, si, bsdf, smp, pc, &distrib));
            L = L + Ld;
        }
        V3 wo = -ray.d, wi; Float pdf; int flags = 0;
        Float u0, u1; smp.BsdfDims(&u0, &u1);
        RGB f = bsdf.Sample_f(wo, &wi, u0, u1, &pdf, BSDF_ALL, &flags);
        if (IsBlack(f) || pdf == 0.f) break;
        beta = beta * (f * AbsDot(wi, si.n) / pdf);
        specularBounce = (flags & BSDF_SPECULAR) != 0;
        if ((flags & BSDF_SPECULAR) && (flags & BSDF_TRANSMISSION)) {
            Float eta = bsdf.eta;
            etaScale *= (Dot(wo, si.n) > 0) ? (eta * eta) : 1 / (eta * eta);
        }
        Ray next = {OffsetRayOrigin(si.p, si.n, wi), wi, Infinity};
        ray = next;
        RGB rrBeta = beta * etaScale;
        if (MaxComp(rrBeta) < rrThreshold && bounces > 3) {
            Float q = std::max((Float).05, 1 - MaxComp(rrBeta));
            if (smp.RRDim() < q) break;
            beta = beta / (1 - q);
        }
    }
    return L;
}

// camera.cpp:52-58
inline Ray GenerateRay(const rt_camera &cam, Float fx, Float fy) {
    V3 pCamera = XfPoint(cam.raster_to_camera, mk(fx, fy, 0));
    Ray r = {mk(0, 0, 0), Normalize(pCamera), Infinity};
    Ray w = {XfPoint(cam.camera_to_world, r.o), XfVector(cam.camera_to_world, r.d), Infinity};
    return w;
}

}  // namespace

// ------------------------------------------------------------------ C API --
extern "C" {

struct orc_scene { Scene s; };

int orc_bvh_build(int32_t n, const float *prim_bounds, rt_bvh_node *nodes_out, int32_t *n_nodes_out, int32_t *order_out) {
    return bvh_build(n, prim_bounds, nodes_out, n_nodes_out, order_out);
}
int orc_bvh_build_ex(int32_t n, const float *prim_bounds, int32_t max_prims, int32_t method, rt_bvh_node *nodes_out, int32_t *n_nodes_out, int32_t *order_out) {
    return bvh_build(n, prim_bounds, nodes_out, n_nodes_out, order_out, max_prims, method);
}
int orc_prim_bounds(const rt_scene_desc *d, float *out) { prim_bounds(d, out); return 0; }

orc_scene *orc_scene_create(const rt_scene_desc *d) {
    orc_scene *h = new orc_scene; Scene &s = h->s;
    s.n_prims = d->n_prims;
    s.prim_kind.assign(d->prim_kind, d->prim_kind + d->n_prims);
    s.prim_geom.assign(d->prim_geom, d->prim_geom + d->n_prims);
    s.prim_material.assign(d->prim_material, d->prim_material + d->n_prims);
    s.prim_light.assign(d->prim_light, d->prim_light + d->n_prims);
    s.tri_p.assign(d->tri_p, d->tri_p + 9 * (size_t)d->n_tris);
    s.tri_uv.assign(d->tri_uv, d->tri_uv + 6 * (size_t)d->n_tris);
    if (d->tri_n && d->tri_has_n) { s.tri_n.assign(d->tri_n, d->tri_n + 9 * (size_t)d->n_tris); s.tri_has_n.assign(d->tri_has_n, d->tri_has_n + d->n_tris); }
    if (d->n_spheres) s.spheres.assign(d->spheres, d->spheres + d->n_spheres);
    s.materials.assign(d->materials, d->materials + d->n_materials);
    s.textures.assign(d->textures, d->textures + d->n_textures);
    s.texel_store.resize(d->n_textures);
    for (int i = 0; i < d->n_textures; ++i)
        if (s.textures[i].kind == RT_TEX_IMAGE) {
            size_t n = 3 * (size_t)s.textures[i].width * s.textures[i].height;
            s.texel_store[i].assign(d->textures[i].texels, d->textures[i].texels + n);
            s.textures[i].texels = s.texel_store[i].data();
        }
    if (d->n_lights) { s.lights.assign(d->lights, d->lights + d->n_lights); BuildLightCdf(d->n_lights, s.light_cdf, s.light_func_int); }
    for (int i = 0; i < d->n_env_lights; ++i) {
        rt_env_light e = d->env_lights[i];
        auto keep = [&](const float *p, size_t n) { s.env_store.emplace_back(p, p + n); return (const float *)s.env_store.back().data(); };
        e.texels = keep(e.texels, 3 * (size_t)e.width * e.height); e.func = keep(e.func, (size_t)e.nu * e.nv);
        e.cdf = keep(e.cdf, (size_t)e.nv * (e.nu + 1)); e.func_int = keep(e.func_int, e.nv); e.marginal_cdf = keep(e.marginal_cdf, e.nv + 1);
        s.envs.push_back(e);
    }
    if (d->bvh_nodes) {
        s.nodes.assign(d->bvh_nodes, d->bvh_nodes + d->n_nodes);
        s.order.assign(d->prim_order, d->prim_order + d->n_prims);
    } else if (d->n_prims > 0) {
        std::vector<float> pb(6 * (size_t)d->n_prims);
        prim_bounds(d, pb.data());
        s.nodes.resize(2 * (size_t)d->n_prims); s.order.resize(d->n_prims);
        int32_t nn = 0;
        bvh_build(d->n_prims, pb.data(), s.nodes.data(), &nn, s.order.data());
        s.nodes.resize(nn);
    }
    return h;
}
void orc_scene_destroy(orc_scene *h) { delete h; }
int orc_scene_get_bvh(orc_scene *h, rt_bvh_node *nodes_out, int32_t *n_nodes_out, int32_t *order_out) {
    *n_nodes_out = (int32_t)h->s.nodes.size();
    if (nodes_out) memcpy(nodes_out, h->s.nodes.data(), sizeof(rt_bvh_node) * h->s.nodes.size());
    if (order_out) memcpy(order_out, h->s.order.data(), sizeof(int32_t) * h->s.order.size());
    return 0;
}

// out_nodes / out_tests (may be NULL): per-ray counts of slab tests and primitive tests.
int orc_trace_closest(orc_scene *h, size_t n, const float *o, const float *d, const float *tmax, int32_t *out_prim, float *out_t,
                      float *out_b, uint32_t *out_nodes, uint32_t *out_tests, int threads) {
    if (threads < 1) threads = 1;
    auto work = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            Ray r = {mk(o[3 * i], o[3 * i + 1], o[3 * i + 2]), mk(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmax[i]};
            Hit hit; TraceCounters c = {0, 0};
            IntersectClosest(h->s, r, &hit, &c);
            out_prim[i] = hit.prim; out_t[i] = hit.t;
            if (out_b) {
                if (hit.prim >= 0 && h->s.prim_kind[hit.prim] == RT_PRIM_SPHERE) {
                    SurfaceInteraction si; FillInteraction(h->s, r, hit, &si); out_b[2 * i] = si.u; out_b[2 * i + 1] = si.v;
                } else { out_b[2 * i] = hit.b1; out_b[2 * i + 1] = hit.b2; }
            }
            if (out_nodes) out_nodes[i] = c.nodes;
            if (out_tests) out_tests[i] = c.tests;
        }
    };
    std::vector<std::thread> pool; size_t chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) { size_t lo = t * chunk, hi = std::min(n, lo + chunk); if (lo < hi) pool.emplace_back(work, lo, hi); }
    for (auto &t : pool) t.join();
    return 0;
}
int orc_trace_any(orc_scene *h, size_t n, const float *o, const float *d, const float *tmax, uint8_t *out_hit) {
    for (size_t i = 0; i < n; ++i) {
        Ray r = {mk(o[3 * i], o[3 * i + 1], o[3 * i + 2]), mk(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmax[i]};
        out_hit[i] = IntersectAny(h->s, r, nullptr) ? 1 : 0;
    }
    return 0;
}

// mode 0: reference sampler (StratifiedSampler + PCG32, seed = pixel index + seed_offset);
// mode 1: Philox sampler keyed (pixel, sample, bounce) with key (opts->seed_lo, seed_hi).
// Work is split over `threads` by pixel; the result does not depend on the thread count.
int orc_render(orc_scene *h, const rt_camera *cam, const rt_render_opts *opts, int mode, int64_t seed_offset, int threads,
               float *out_sum_rgb, uint64_t *out_counts /* [samples, closest, shadow] or NULL */) {
    const Scene &sc = h->s;
    int W = cam->width, H = cam->height, nx = opts->strata_x, ny = opts->strata_y;
    int64_t npix = (int64_t)W * H;
    if (threads < 1) threads = 1;
    const bool use_li = (opts->flags & RT_RENDER_LI) != 0, all_lights = (opts->flags & RT_RENDER_ALL_LIGHTS) != 0;
    std::atomic<int64_t> next(0);
    std::vector<PathCounters> pcs(threads); std::vector<uint64_t> nsamp(threads, 0);
    auto work = [&](int tid) {
        PathCounters pc; pc.closest = pc.shadow = 0;      // thread-local: neighbours in `pcs` / `nsamp` share cache lines
        uint64_t ns = 0;
        const int64_t grain = 256;
        while (true) {
            int64_t lo = next.fetch_add(grain); if (lo >= npix) break;
            int64_t hi = std::min(npix, lo + grain);
            for (int64_t pos = lo; pos < hi; ++pos) {
                int j = (int)(pos / W), i = (int)(pos - (int64_t)j * W);
                RGB col = rgb1(0.0);
                if (mode == 0) {
                    RefSampler smp(nx, ny, (uint64_t)(int)(i + j * W + seed_offset));
                    smp.StartPixel();
                    do {
                        if (smp.idx >= opts->sample_begin && smp.idx < opts->sample_end) {
                            Float fx, fy; smp.Get2D(&fx, &fy);
                            Float time = smp.Get1D(); (void)time;
                            Float l0, l1; smp.Get2D(&l0, &l1);
                            Ray r = GenerateRay(*cam, (Float)i + fx, (Float)j + fy);
                            bool spec = false;
                            col = col + (use_li ? Li(sc, r, smp, opts->max_depth, opts->rr_threshold, all_lights, &pc)
                                                : Li_re(sc, r, smp, 0, opts->max_depth, &spec, &pc, all_lights));
                            ns++;
                        }
                    } while (smp.StartNextSample());
                } else {
                    PhiloxSampler smp; smp.k0 = opts->seed_lo; smp.k1 = opts->seed_hi;
                    for (int s = opts->sample_begin; s < opts->sample_end; ++s) {
                        smp.Begin((uint32_t)pos, (uint32_t)s);
                        Float fx, fy; smp.Film(nx, ny, &fx, &fy);
                        Ray r = GenerateRay(*cam, (Float)i + fx, (Float)j + fy);
                        bool spec = false;
                        col = col + (use_li ? Li(sc, r, smp, opts->max_depth, opts->rr_threshold, all_lights, &pc)
                                            : Li_re(sc, r, smp, 0, opts->max_depth, &spec, &pc, all_lights));
                        ns++;
                    }
                }
                out_sum_rgb[3 * pos] = col[0]; out_sum_rgb[3 * pos + 1] = col[1]; out_sum_rgb[3 * pos + 2] = col[2];
            }
        }
        pcs[tid] = pc; nsamp[tid] = ns;
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(work, t);
    for (auto &t : pool) t.join();
    if (out_counts) {
        out_counts[0] = out_counts[1] = out_counts[2] = 0;
        for (int t = 0; t < threads; ++t) { out_counts[0] += nsamp[t]; out_counts[1] += pcs[t].closest; out_counts[2] += pcs[t].shadow; }
    }
    return 0;
}

// integrator.cpp:289-308 (after colObj / samplesPerPixel at :337)
int orc_tonemap_u8(const float *sum, int32_t W, int32_t H, int32_t spp, uint8_t *out) {
    for (int64_t p = 0; p < (int64_t)W * H; ++p)
        for (int c = 0; c < 3; ++c) {
            Float r = sum[3 * p + c] / (Float)spp;   // Spectrum / int64 samplesPerPixel -> operator/(Float)
            if (r != r) r = 0.0;
            double scale = 1.0 / 1;
            r = std::sqrt(scale * r);                // double sqrt, stored back into a Float
            Float lo = 0.0, hi = 0.999;              // clamp(Float, Float, Float)
            Float cl = r < lo ? lo : (r > hi ? hi : r);
            out[3 * p + c] = (uint8_t)static_cast<int>(256 * cl);
        }
    return 0;
}

// ---- known-answer helpers (single evaluations, used to pin pieces separately) ----
int orc_camera_ray(const rt_camera *cam, float fx, float fy, float *o, float *d) {
    Ray r = GenerateRay(*cam, fx, fy);
    o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; d[0] = r.d.x; d[1] = r.d.y; d[2] = r.d.z; return 0;
}
int orc_philox(const uint32_t ctr[4], uint32_t k0, uint32_t k1, uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]}; Philox4x32(c, k0, k1); memcpy(out, c, sizeof(c)); return 0;
}
int orc_film_sample(uint32_t pixel, uint32_t sample, uint32_t k0, uint32_t k1, int nx, int ny, float *fx, float *fy) {
    PhiloxSampler s; s.k0 = k0; s.k1 = k1; s.Begin(pixel, sample); s.Film(nx, ny, fx, fy); return 0;
}
int orc_texture_eval(orc_scene *h, int tex, float u, float v, float *out) {
    RGB c = EvalTexture(h->s, tex, u, v); out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; return 0;
}
// InfiniteAreaLight::Sample_Li of env light `env` for reference point p and sample u: out = wi[3], pdf, Li[3], p1[3] (10 floats)
int orc_env_sample(orc_scene *h, int env, const float *p, float u0, float u1, float *out10) {
    if (env < 0 || env >= (int)h->s.envs.size()) return -1;
    V3 wi = mk(0, 0, 0), p1 = mk(0, 0, 0); Float pdf = 0;
    RGB Li = EnvSampleLi(h->s.envs[env], mk(p[0], p[1], p[2]), u0, u1, &wi, &pdf, &p1);
    float v[10] = {wi.x, wi.y, wi.z, pdf, Li[0], Li[1], Li[2], p1.x, p1.y, p1.z};
    memcpy(out10, v, sizeof(v)); return 0;
}
// Interaction at the closest hit of one ray: p, n, wo, uv, dpdu (14 floats) ; returns prim or -1.
int orc_interaction(orc_scene *h, const float *o, const float *d, float tmax, float *out14) {
    Ray r = {mk(o[0], o[1], o[2]), mk(d[0], d[1], d[2]), tmax}; Hit hit;
    if (!IntersectClosest(h->s, r, &hit, nullptr)) return -1;
    SurfaceInteraction si; FillInteraction(h->s, r, hit, &si);
    float v[14] = {si.p.x, si.p.y, si.p.z, si.n.x, si.n.y, si.n.z, si.wo.x, si.wo.y, si.wo.z, si.u, si.v, si.dpdu.x, si.dpdu.y, si.dpdu.z};
    memcpy(out14, v, sizeof(v)); return hit.prim;
}
// BSDF at the closest hit of ray (o,d): f(wo, wiW) for the non-specular lobes and Sample_f(u0,u1).
// out: f[3], sampled wi[3], sampled f[3], pdf, flags  (11 floats)
int orc_bsdf_probe(orc_scene *h, const float *o, const float *d, const float *wiW, float u0, float u1, float *out11) {
    Ray r = {mk(o[0], o[1], o[2]), mk(d[0], d[1], d[2]), Infinity}; Hit hit;
    if (!IntersectClosest(h->s, r, &hit, nullptr)) return -1;
    SurfaceInteraction si; FillInteraction(h->s, r, hit, &si);
    BSDF b; ComputeScatteringFunctions(h->s, si, &b);
    RGB f = b.f(si.wo, mk(wiW[0], wiW[1], wiW[2]), BSDF_ALL & ~BSDF_SPECULAR);
    V3 wi = mk(0, 0, 0); Float pdf = 0; int flags = 0;
    RGB fs = b.Sample_f(-r.d, &wi, u0, u1, &pdf, BSDF_ALL, &flags);
    float v[11] = {f[0], f[1], f[2], wi.x, wi.y, wi.z, fs[0], fs[1], fs[2], pdf, (float)flags};
    memcpy(out11, v, sizeof(v)); return hit.prim;
}

}  // extern "C"

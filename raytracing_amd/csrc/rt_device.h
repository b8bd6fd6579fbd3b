// rt_device.h — device-side data layout and inline device functions of the wavefront path tracer.
//
// Semantics follow the reference renderer (paths relative to RayTracing/include/core/); every
// function cites what it restates.  Arithmetic that decides hit / miss keeps the reference's
// operand order and is compiled with -ffp-contract=off so that closest-hit results are
// bit-identical with the CPU renderer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_WAVE 64
#define RT_BLOCK 256
#define RT_LDS_STACK 16        // traversal stack entries kept in LDS per lane
#define RT_MAX_STACK 64        // the reference's nodesToVisit[64] (bvh.cpp:253)

#define RT_PI 3.14159265358979323846f
#define RT_INV_PI 0.31830988618379067154f
#define RT_INV_2PI 0.15915494309189533577f
#define RT_PI_OVER_2 1.57079632679489661923f
#define RT_PI_OVER_4 0.78539816339744830961f
#define RT_ONE_MINUS_EPS 0.99999994f
#define RT_INF __builtin_huge_valf()

// slot flags
#define RT_SLOT_REJECT_CLOSEST 1u   // Triangle::Intersect returns false for this triangle (degenerate, triangle.cpp:139-144)
#define RT_SLOT_SPHERE 2u

enum { RT_BSDF_REFLECTION = 1, RT_BSDF_TRANSMISSION = 2, RT_BSDF_DIFFUSE = 4, RT_BSDF_GLOSSY = 8, RT_BSDF_SPECULAR = 16 };

struct DevTexture { int kind, width, height, pad; float value[3]; float su, sv, du, dv; const float *texels; };
struct DevMaterial { int kind; int tex[4]; int remap; };
struct DevLight {
    float p0[3], p1[3], p2[3];
    float n[3];          // Normalize(Cross(p1-p0, p2-p0)), triangle.cpp:236
    float vn[9];         // vertex normals (has_n)
    float Le[3];
    float area;          // 0.5 * |Cross(p1-p0, p2-p0)|, triangle.cpp:222-227
    int two_sided, has_n;
    int kind, env;       // RT_LIGHT_INFINITE: index into DevScene::envs (the triangle fields are unused)
};
// InfiniteAreaLight as Sample_Li reads it (infinite.cpp:83-106): radiance map level 0, Distribution2D tables, transform, scene radius.
struct DevEnv {
    float l2w[12];                 // rows 0..2 of LightToWorld
    float w2l[12];                 // rows 0..2 of WorldToLight
    int width, height, nu, nv;
    const float *texels, *func, *cdf, *func_int, *marginal_cdf;
    float marginal_int, world_radius;
};

// Scene as the kernels see it.  Everything geometric is indexed by BVH leaf slot.
struct DevScene {
    const float4 *nodes;      // 2 x float4 per node == rt_bvh_node bytes
    const float4 *slots;      // 3 x float4 per slot: p0.xyz p1.x | p1.yz p2.xy | p2.z flags prim geom
    const float4 *frames;     // 3 x float4 per slot (triangles): n | ss | ts  (flat normal, Normalize(dpdu), Cross(n, ss))
    const float *slot_uv;     // 6 per slot
    const float *spheres;     // 32 floats per sphere: radius,0,0,0 | w2o rows 0..2 (12) | o2w rows 0..2 (12) | pad 4
    const int *slot_material;
    const int *slot_light;
    const DevMaterial *materials;
    const DevTexture *textures;
    const DevLight *lights;
    const float4 *pairs;      // 4 x float4 per INTERIOR node: left box | right box | left info, right info, split axis, -  (rt_traverse2.h)
    int n_nodes, n_slots, n_lights, n_pairs;
    float root_box[6];        // bounds of node 0
    uint32_t root_info;       // pair index of the root, or a leaf code when the tree is a single leaf
    const float4 *quads;      // 8 x float4 per even-depth interior node (rt_traverse2.h); null when the tree does not qualify
    int n_quads, quad_stack_cap;
    uint32_t pad;
    int n_materials, n_textures;
    const DevEnv *envs;       // infinite area lights
    const float *light_cdf;   // n_lights + 1: cdf of the uniform Distribution1D over the lights (lightdistrib.cpp:12-15, sampling.h:16-27)
    float light_func_int;     // its funcInt
    int n_envs;
    int root_implied;         // the root's box is bit for bit the union of its children's boxes (lane_begin may skip its test for plain rays)
    const float4 *octs;       // 8 x float4 per record of the eight-wide tree with quantised boxes (rt_oct.h); null when the scene has none
    int n_octs, oct_stack_cap;
    const float4 *flat;       // 2 x float4 per leaf, in depth-first order: box min xyz, max x | max yz, info word (as in the pair records), -
    int n_flat;               //   present (non-null) only for trees of at most RT_FLAT_MAX leaves whose boxes nest exactly (k_extend_flat)
};

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ F3 operator*(float s, F3 v) { return f3(s * v.x, s * v.y, s * v.z); }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ F3 vdiv(F3 v, float f) { float inv = 1.f / f; return f3(v.x * inv, v.y * inv, v.z * inv); }   // geometry.h:149-153
__device__ __forceinline__ F3 cdiv(F3 v, float f) { return f3(v.x / f, v.y / f, v.z / f); }                               // spectrum.h:92-96
// Division and square root for the VALUES of the shading arithmetic every path computes (light pdfs, BSDF value times cosine over
// pdf, throughput, roulette weights): the hardware reciprocal and square root, 1 ulp, two instructions instead of the ten to thirteen of
// the correctly rounded sequences.  Nothing that decides where a path goes uses these: the slab, triangle and sphere tests, inverse
// ray directions, camera rays, interaction frames and sampled directions keep IEEE '/' and sqrtf (which hipcc rounds correctly) — a
// last-bit change of a direction moves a hit across a triangle edge once in about 10^4 hits, which a per-pixel comparison with the
// CPU restatement shows on a fine mesh — and so do the microfacet lobes, whose sampling formulas amplify a last-bit difference.
// -DRT_SHADE_IEEE=1 makes these IEEE too.
#ifndef RT_SHADE_IEEE
#define RT_SHADE_IEEE 0
#endif
#if RT_SHADE_IEEE
__device__ __forceinline__ float frcp(float b) { return 1.f / b; }
__device__ __forceinline__ float fdiv(float a, float b) { return a / b; }
__device__ __forceinline__ float fsqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ float frsq(float x) { return 1.f / sqrtf(x); }
__device__ __forceinline__ F3 cdiv_s(F3 v, float f) { return cdiv(v, f); }
#else
__device__ __forceinline__ float frcp(float b) { return __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float frsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ F3 cdiv_s(F3 v, float f) { float inv = __builtin_amdgcn_rcpf(f); return f3(v.x * inv, v.y * inv, v.z * inv); }
#endif
__device__ __forceinline__ float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float lensq(F3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ F3 normalize3(F3 v) { return vdiv(v, sqrtf(lensq(v))); }
__device__ __forceinline__ F3 normalize3_s(F3 v) { float inv = frsq(lensq(v)); return f3(v.x * inv, v.y * inv, v.z * inv); }
__device__ __forceinline__ F3 cross3(F3 a, F3 b) {   // geometry.h:183-190: double intermediates
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return f3((float)((ay * bz) - (az * by)), (float)((az * bx) - (ax * bz)), (float)((ax * by) - (ay * bx)));
}
__device__ __forceinline__ bool black3(F3 a) { return a.x == 0.f && a.y == 0.f && a.z == 0.f; }
__device__ __forceinline__ float maxcomp3(F3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }

// geometry.h:916-923
__device__ __forceinline__ F3 offset_ray_origin(F3 p, F3 n, F3 w) {
    float d = fabsf(n.x) * 1e-4f + fabsf(n.y) * 1e-4f + fabsf(n.z) * 1e-4f;
    F3 off = d * n;
    if (dot3(w, n) < 0) off = -off;
    return p + off;
}

// --------------------------------------------------------------------- RNG --
// Philox4x32-10 (Salmon et al. 2011), counter (c0..c3), key (k0,k1).
__device__ __forceinline__ void philox4x32(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // (one 32 x 32 -> 64 bit multiply-add instruction per product instead of a low and a high multiply)
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
        uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
// RNG::UniformFloat's integer->float mapping (rng.h:38-45)
__device__ __forceinline__ float u32_to_unit(uint32_t v) { return fminf(RT_ONE_MINUS_EPS, (float)v * 2.3283064365386963e-10f); }

// Unsigned division by a number that is the same for the whole launch (Granlund & Montgomery 1994): the host precomputes
// l = ceil(log2 d), m = floor(2^32 (2^l - d) / d) + 1; then n / d = (t + ((n - t) >> min(l, 1))) >> max(l - 1, 0) with t = mulhi(m, n),
// exact for every 32-bit n.  Five instructions instead of the forty of a general 32-bit division.
struct FastDiv { uint32_t m, s1, s2, d; };
__host__ __device__ inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f; f.d = d;
    uint32_t l = 0; while (l < 32 && (1ull << l) < d) ++l;
    f.m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.s1 = l < 1 ? l : 1; f.s2 = l > 1 ? l - 1 : 0;
    return f;
}
__device__ __forceinline__ uint32_t fast_div(uint32_t n, const FastDiv &f) { uint32_t t = __umulhi(f.m, n); return (t + ((n - t) >> f.s1)) >> f.s2; }
__device__ __forceinline__ uint32_t fast_mod(uint32_t n, const FastDiv &f) { return n - fast_div(n, f) * f.d; }

// Stateless permutation of {0..l-1} (Kensler 2013, "Correlated Multi-Jittered Sampling").
__device__ __forceinline__ uint32_t kensler_permute(uint32_t i, uint32_t l, uint32_t p) {
    uint32_t w = l - 1;
    w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
    do {
        i ^= p; i *= 0xe170893d; i ^= p >> 16; i ^= (i & w) >> 4; i ^= p >> 8; i *= 0x0929eb3f; i ^= p >> 23;
        i ^= (i & w) >> 1; i *= 1 | p >> 27; i *= 0x6935fa69; i ^= (i & w) >> 11; i *= 0x74dcb303;
        i ^= (i & w) >> 2; i *= 0x9e501cc3; i ^= (i & w) >> 2; i *= 0xc860a3df; i &= w; i ^= i >> 5;
    } while (i >= l);
    return (i + p) % l;
}
__device__ __forceinline__ uint32_t kensler_permute(uint32_t i, const FastDiv &fl, uint32_t p) {      // the same, l = fl.d
    const uint32_t l = fl.d;
    uint32_t w = l - 1;
    w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
    do {
        i ^= p; i *= 0xe170893d; i ^= p >> 16; i ^= (i & w) >> 4; i ^= p >> 8; i *= 0x0929eb3f; i ^= p >> 23;
        i ^= (i & w) >> 1; i *= 1 | p >> 27; i *= 0x6935fa69; i ^= (i & w) >> 11; i *= 0x74dcb303;
        i ^= (i & w) >> 2; i *= 0x9e501cc3; i ^= (i & w) >> 2; i *= 0xc860a3df; i &= w; i ^= i >> 5;
    } while (i >= l);
    return fast_mod(i + p, fl);
}

// ------------------------------------------------------------ intersection --
// Bounds3::IntersectP(ray, invDir, dirIsNeg), geometry.h:895-914.  n0/n1 are the node's two float4.
__device__ __forceinline__ bool slab_test(float4 n0, float4 n1, F3 o, F3 invDir, int negx, int negy, int negz, float rayTMax) {
    const float widen = 1 + 2 * ((3 * 5.9604644775390625e-08f) / (1 - 3 * 5.9604644775390625e-08f));   // 1 + 2*gamma(3)
    float bminx = n0.x, bminy = n0.y, bminz = n0.z, bmaxx = n0.w, bmaxy = n1.x, bmaxz = n1.y;
    float tMin = ((negx ? bmaxx : bminx) - o.x) * invDir.x;
    float tMax = ((negx ? bminx : bmaxx) - o.x) * invDir.x;
    float tyMin = ((negy ? bmaxy : bminy) - o.y) * invDir.y;
    float tyMax = ((negy ? bminy : bmaxy) - o.y) * invDir.y;
    tMax *= widen;
    tyMax *= widen;
    if (tMin > tyMax || tyMin > tMax) return false;
    if (tyMin > tMin) tMin = tyMin;
    if (tyMax < tMax) tMax = tyMax;
    float tzMin = ((negz ? bmaxz : bminz) - o.z) * invDir.z;
    float tzMax = ((negz ? bminz : bmaxz) - o.z) * invDir.z;
    tzMax *= widen;
    if (tMin > tzMax || tzMin > tMax) return false;
    if (tzMin > tMin) tMin = tzMin;
    if (tzMax < tMax) tMax = tzMax;
    return (tMin < rayTMax) && (tMax > 0);
}

__device__ __forceinline__ float pick3(F3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }

// Triangle::Intersect / IntersectP up to the `t <= eps` test, triangle.cpp:52-119.
// The permutation (kz = largest |d| component) and the shear constants depend on the ray alone (triangle.cpp:62-72);
// ray_shear computes them once per ray and triangle_test_sheared uses them, with the same operations in the same order.
struct RayShear { int kz; float Sx, Sy, Sz; };
__device__ __forceinline__ RayShear ray_shear(F3 dir) {
    float ax = fabsf(dir.x), ay = fabsf(dir.y), az = fabsf(dir.z);
    RayShear r;
    r.kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
    int kx = r.kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    float dx = pick3(dir, kx), dy = pick3(dir, ky), dz = pick3(dir, r.kz);
    r.Sx = -dx / dz; r.Sy = -dy / dz; r.Sz = 1.f / dz;
    return r;
}
// The same with the ray's reciprocal direction at hand: Sz = 1 / d[kz] is that component of it, bit for bit (one IEEE division fewer per ray).
__device__ __forceinline__ RayShear ray_shear(F3 dir, F3 invDir) {
    float ax = fabsf(dir.x), ay = fabsf(dir.y), az = fabsf(dir.z);
    RayShear r;
    r.kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
    int kx = r.kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    float dx = pick3(dir, kx), dy = pick3(dir, ky), dz = pick3(dir, r.kz);
    r.Sx = -dx / dz; r.Sy = -dy / dz; r.Sz = pick3(invDir, r.kz);
    return r;
}
__device__ __forceinline__ bool triangle_test_sheared(F3 p0, F3 p1, F3 p2, F3 o, const RayShear &sh, float rayTMax, float &t, float &b0, float &b1, float &b2) {
    F3 p0t = p0 - o, p1t = p1 - o, p2t = p2 - o;
    int kz = sh.kz;
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    float p0x = pick3(p0t, kx), p0y = pick3(p0t, ky), p0z = pick3(p0t, kz);
    float p1x = pick3(p1t, kx), p1y = pick3(p1t, ky), p1z = pick3(p1t, kz);
    float p2x = pick3(p2t, kx), p2y = pick3(p2t, ky), p2z = pick3(p2t, kz);
    float Sx = sh.Sx, Sy = sh.Sy, Sz = sh.Sz;
    p0x += Sx * p0z; p0y += Sy * p0z;
    p1x += Sx * p1z; p1y += Sy * p1z;
    p2x += Sx * p2z; p2y += Sy * p2z;
    float e0 = p1x * p2y - p1y * p2x;
    float e1 = p2x * p0y - p2y * p0x;
    float e2 = p0x * p1y - p0y * p1x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        double p2txp1ty = (double)p2x * (double)p1y, p2typ1tx = (double)p2y * (double)p1x;
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)p0x * (double)p2y, p0typ2tx = (double)p0y * (double)p2x;
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)p1x * (double)p0y, p1typ0tx = (double)p1y * (double)p0x;
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
    float det = e0 + e1 + e2;
    if (det == 0) return false;
    p0z *= Sz; p1z *= Sz; p2z *= Sz;
    float tScaled = e0 * p0z + e1 * p1z + e2 * p2z;
    if (det < 0 && (tScaled >= 0 || tScaled < rayTMax * det)) return false;
    else if (det > 0 && (tScaled <= 0 || tScaled > rayTMax * det)) return false;
    float invDet = 1 / det;
    b0 = e0 * invDet; b1 = e1 * invDet; b2 = e2 * invDet;
    t = tScaled * invDet;
    if (t <= 1e-4f) return false;
    return true;
}
__device__ __forceinline__ bool triangle_test(F3 p0, F3 p1, F3 p2, F3 o, F3 dir, float rayTMax, float &t, float &b0, float &b1, float &b2) {
    RayShear sh = ray_shear(dir);
    return triangle_test_sheared(p0, p1, p2, o, sh, rayTMax, t, b0, b1, b2);
}

// Affine point / vector transforms by a 3x4 row-major matrix (Transform::operator(), transform.h:135-153;
// the projective row of every shape transform is (0,0,0,1), so w == 1 and the divide never runs).
__device__ __forceinline__ F3 xf_point(const float *m, F3 p) {
    return f3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7], m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
__device__ __forceinline__ F3 xf_vector(const float *m, F3 v) {
    return f3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
// Normal transform by o2w == transpose of w2o's upper 3x3 applied to n (transform.h:154-160).
__device__ __forceinline__ F3 xf_normal_by_inverse(const float *mInv, F3 n) {
    return f3(mInv[0] * n.x + mInv[4] * n.y + mInv[8] * n.z, mInv[1] * n.x + mInv[5] * n.y + mInv[9] * n.z, mInv[2] * n.x + mInv[6] * n.y + mInv[10] * n.z);
}

// Sphere::Intersect / IntersectP up to the parametric hit, sphere.cpp:29-50.  sp = 32-float sphere record.
__device__ __forceinline__ bool sphere_test(const float *sp, F3 o, F3 d, float rayTMax, float &tHit) {
    float radius = sp[0];
    F3 oo = xf_point(sp + 4, o), dd = xf_vector(sp + 4, d);
    F3 oc = oo - f3(0.f, 0.f, 0.f);
    float a = lensq(dd);
    float half_b = dot3(oc, dd);
    float c = lensq(oc) - radius * radius;
    float disc = half_b * half_b - a * c;
    if (disc < 0) return false;
    float sqrtd = sqrtf(disc);
    float t0 = (-half_b - sqrtd) / a;
    float t1 = (-half_b + sqrtd) / a;
    if (t0 > rayTMax || t1 <= 0) return false;
    float ts = t0;
    if (ts <= 0) { ts = t1; if (ts > rayTMax) return false; }
    tHit = ts;
    return true;
}

struct HitRec { int slot; float t, b0, b1, b2; };

// Per-lane traversal stack.  The newest N entries live in an LDS ring ([entry][thread]: bank = thread, conflict-free);
// when the ring is full the oldest entry moves to a global spill area ([entry][global thread]) and comes back when the
// stack shrinks down to it, so the hot top of the stack never leaves LDS however deep the tree is.  Logical depth is
// capped at RT_MAX_STACK like the reference's nodesToVisit[64] (bvh.cpp:253).
template <int N>
struct TravStackT {
    int *lds;            // &shared[0][threadIdx.x], stride RT_BLOCK
    int *spill;          // &spill[0][global thread]
    size_t spill_stride;
    int sp, lo;          // entries [lo, sp) are in the ring
    __device__ __forceinline__ void reset() { sp = 0; lo = 0; }
    __device__ __forceinline__ void push(int v) {
        if (sp - lo == N) {
            if (lo < RT_MAX_STACK) spill[(size_t)lo * spill_stride] = lds[(lo & (N - 1)) * RT_BLOCK];
            lo++;
        }
        lds[(sp & (N - 1)) * RT_BLOCK] = v;
        sp++;
    }
    __device__ __forceinline__ int pop() {
        sp--;
        if (sp < lo) { lo = sp; return sp < RT_MAX_STACK ? spill[(size_t)sp * spill_stride] : 0; }
        return lds[(sp & (N - 1)) * RT_BLOCK];
    }
};
typedef TravStackT<RT_LDS_STACK> TravStack;

// Where traversal reads nodes and leaf slots from: HBM (through L1/L2), or a copy of a small scene in LDS.
struct GlobalFetch {
    const float4 *nodes, *slots;
    __device__ __forceinline__ float4 node(int i) const { return nodes[i]; }
    __device__ __forceinline__ float4 slot(int i) const { return slots[i]; }
};
struct LdsFetch {
    const float4 *nodes, *slots;   // point into __shared__ arrays (address space resolved after inlining)
    __device__ __forceinline__ float4 node(int i) const { return nodes[i]; }
    __device__ __forceinline__ float4 slot(int i) const { return slots[i]; }
};
#define RT_LDS_SCENE_NODES 128
#define RT_LDS_SCENE_SLOTS 64

// BVHAccel::Intersect (bvh.cpp:247-282) when ANY == false, BVHAccel::IntersectP (bvh.cpp:284-320) when ANY == true.
// The visit order, the slab tests and the primitive tests are the reference's; only the control flow is arranged for
// a 64-wide wave: an inner loop walks interior nodes until the lane reaches a leaf (or runs out of nodes), then the
// lanes that hold a leaf test its primitives together.  COUNT adds the number of slab / primitive tests to nNodes / nTests.
template <bool ANY, bool COUNT, class Fetch, class Stack>
__device__ __forceinline__ bool traverse(const DevScene &sc, const Fetch &fetch, F3 o, F3 d, float rayTMax, Stack &st, HitRec &hit, uint32_t &nNodes,
                                         uint32_t &nTests) {
    hit.slot = -1; hit.t = rayTMax; hit.b0 = hit.b1 = hit.b2 = 0.f;
    if (sc.n_nodes == 0) return false;
    F3 invDir = f3(1.f / d.x, 1.f / d.y, 1.f / d.z);
    int negx = invDir.x < 0, negy = invDir.y < 0, negz = invDir.z < 0;
    bool found = false;
    int cur = 0;
    st.reset();
    while (cur >= 0) {
        int leafFirst = 0, leafCount = 0;
        while (true) {   // interior walk
            float4 n0 = fetch.node(2 * cur), n1 = fetch.node(2 * cur + 1);
            if (COUNT) nNodes++;
            if (slab_test(n0, n1, o, invDir, negx, negy, negz, rayTMax)) {
                int offset = __float_as_int(n1.z);
                uint32_t meta = __float_as_uint(n1.w);
                int nprim = meta & 0xFFFF, axis = (meta >> 16) & 0xFF;
                if (nprim > 0) { leafFirst = offset; leafCount = nprim; break; }
                int neg = axis == 0 ? negx : (axis == 1 ? negy : negz);
                if (neg) { st.push(cur + 1); cur = offset; }
                else { st.push(offset); cur = cur + 1; }
            } else {
                if (st.sp == 0) { cur = -1; break; }
                cur = st.pop();
            }
        }
        for (int i = 0; i < leafCount; ++i) {
            int slot = leafFirst + i;
            float4 r0 = fetch.slot(3 * slot), r1 = fetch.slot(3 * slot + 1), r2 = fetch.slot(3 * slot + 2);
            uint32_t flags = __float_as_uint(r2.y);
            if (COUNT) nTests++;
            float t, b0 = 0.f, b1 = 0.f, b2 = 0.f;
            bool ok;
            if (flags & RT_SLOT_SPHERE) ok = sphere_test(sc.spheres + 32 * (size_t)__float_as_int(r2.w), o, d, rayTMax, t);
            else {
                ok = triangle_test(f3(r0.x, r0.y, r0.z), f3(r0.w, r1.x, r1.y), f3(r1.z, r1.w, r2.x), o, d, rayTMax, t, b0, b1, b2);
                if (!ANY && (flags & RT_SLOT_REJECT_CLOSEST)) ok = false;
            }
            if (ok) {
                if (ANY) return true;
                found = true; rayTMax = t;                       // primitive.cpp:39
                hit.slot = slot; hit.t = t; hit.b0 = b0; hit.b1 = b1; hit.b2 = b2;
            }
        }
        if (leafCount > 0) {
            if (st.sp == 0) cur = -1;
            else cur = st.pop();
        }
    }
    return found;
}

// --------------------------------------------------------------- shading ----
// Texture::Evaluate: ConstantTexture (texture.h:18-25) or ImageTexture -> MIPMap::triangle(0, st)
// (imagemap.h:44-51, texture.cpp:8-13, mipmap.h:190-200,152-171): bilinear, level 0, Repeat.
__device__ __forceinline__ int mod_wrap(int a, int b) { int r = a - (a / b) * b; return r < 0 ? r + b : r; }
__device__ __forceinline__ F3 eval_texture(const DevTexture &t, float u, float v) {
    if (t.kind == 0) return f3(t.value[0], t.value[1], t.value[2]);
    float st0 = t.su * u + t.du, st1 = t.sv * v + t.dv;
    float s = st0 * t.width - 0.5f, tt = st1 * t.height - 0.5f;
    float fs = floorf(s), ft = floorf(tt);
    int s0 = (int)fs, t0 = (int)ft;
    float ds = s - s0, dt = tt - t0;
    int sa, sb, ta, tb;
    if (((t.width & (t.width - 1)) | (t.height & (t.height - 1))) == 0) {       // powers of two (what MIPMap's constructor makes of every image,
        sa = s0 & (t.width - 1); sb = (s0 + 1) & (t.width - 1);                   // mipmap.h:70-121): the Repeat wrap is a mask, negative texels included
        ta = t0 & (t.height - 1); tb = (t0 + 1) & (t.height - 1);
    } else { sa = mod_wrap(s0, t.width); sb = mod_wrap(s0 + 1, t.width); ta = mod_wrap(t0, t.height); tb = mod_wrap(t0 + 1, t.height); }
    const float *A = t.texels + 3 * ((size_t)ta * t.width + sa), *B = t.texels + 3 * ((size_t)tb * t.width + sa);
    const float *Cc = t.texels + 3 * ((size_t)ta * t.width + sb), *D = t.texels + 3 * ((size_t)tb * t.width + sb);
    float w00 = (1 - ds) * (1 - dt), w01 = (1 - ds) * dt, w10 = ds * (1 - dt), w11 = ds * dt;
    return f3(A[0] * w00 + B[0] * w01 + Cc[0] * w10 + D[0] * w11, A[1] * w00 + B[1] * w01 + Cc[1] * w10 + D[1] * w11,
              A[2] * w00 + B[2] * w01 + Cc[2] * w10 + D[2] * w11);
}
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ F3 clamp0(F3 a) { return f3(a.x < 0.f ? 0.f : a.x, a.y < 0.f ? 0.f : a.y, a.z < 0.f ? 0.f : a.z); }   // Spectrum::Clamp(0, inf)

// Distribution1D::SampleContinuous (sampling.h:30-40) with FindInterval (pbrt.h:213-226) over func[0..n), cdf[0..n].
__device__ __forceinline__ float sample_continuous_1d(const float *func, const float *cdf, int n, float funcInt, float u, float &pdf, int &offset) {
    int first = 0, len = n + 1;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (cdf[middle] <= u) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    offset = min(max(first - 1, 0), n - 1);
    float c0 = cdf[offset], c1 = cdf[offset + 1];
    float du = u - c0;
    if ((c1 - c0) > 0) du /= (c1 - c0);
    pdf = (funcInt > 0) ? func[offset] / funcInt : 0.f;
    return (offset + du) / n;
}
// MIPMap::Lookup with a zero filter width: bilinear on level 0, Repeat wrap (mipmap.h:173-200)
__device__ __forceinline__ F3 env_lookup(const DevEnv &e, float u, float v) {
    float s = u * e.width - 0.5f, t = v * e.height - 0.5f;
    float fs = floorf(s), ft = floorf(t);
    int s0 = (int)fs, t0 = (int)ft;
    float ds = s - s0, dt = t - t0;
    int sa = mod_wrap(s0, e.width), sb = mod_wrap(s0 + 1, e.width), ta = mod_wrap(t0, e.height), tb = mod_wrap(t0 + 1, e.height);
    const float *A = e.texels + 3 * ((size_t)ta * e.width + sa), *B = e.texels + 3 * ((size_t)tb * e.width + sa);
    const float *Cc = e.texels + 3 * ((size_t)ta * e.width + sb), *D = e.texels + 3 * ((size_t)tb * e.width + sb);
    float w00 = (1 - ds) * (1 - dt), w01 = (1 - ds) * dt, w10 = ds * (1 - dt), w11 = ds * dt;
    return f3(A[0] * w00 + B[0] * w01 + Cc[0] * w10 + D[0] * w11, A[1] * w00 + B[1] * w01 + Cc[1] * w10 + D[1] * w11,
              A[2] * w00 + B[2] * w01 + Cc[2] * w10 + D[2] * w11);
}
// InfiniteAreaLight::Sample_Li (infinite.cpp:83-106): Distribution2D::SampleContinuous (sampling.h:72-79), direction from the
// (u, v) of the map, pdf with respect to solid angle, radiance = bilinear level-0 lookup (MIPMap::Lookup with zero width).
__device__ __forceinline__ F3 env_sample_li(const DevEnv &e, float u0, float u1, F3 &wi, float &pdf) {
    float pdf0, pdf1; int v, dummy;
    float d1 = sample_continuous_1d(e.func_int, e.marginal_cdf, e.nv, e.marginal_int, u1, pdf1, v);
    float d0 = sample_continuous_1d(e.func + (size_t)v * e.nu, e.cdf + (size_t)v * (e.nu + 1), e.nu, e.func_int[v], u0, pdf0, dummy);
    float mapPdf = pdf0 * pdf1;
    pdf = 0.f; wi = f3(0.f, 0.f, 1.f);
    if (mapPdf == 0) return f3(0.f, 0.f, 0.f);
    float theta = d1 * RT_PI, phi = d0 * 2 * RT_PI;
    float sinTheta, cosTheta, sinPhi, cosPhi;
    sincosf(theta, &sinTheta, &cosTheta); sincosf(phi, &sinPhi, &cosPhi);
    wi = xf_vector(e.l2w, f3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta));
    pdf = mapPdf / (2 * RT_PI * RT_PI * sinTheta);
    if (sinTheta == 0) pdf = 0.f;
    return env_lookup(e, d0, d1);
}
// InfiniteAreaLight::Le (infinite.cpp:77-81): the map at the spherical angles of the direction (geometry.h:935-942)
__device__ __forceinline__ F3 env_le(const DevEnv &e, F3 d) {
    F3 w = normalize3(xf_vector(e.w2l, d));
    float p = atan2f(w.y, w.x);
    float phi = (p < 0) ? (p + 2 * RT_PI) : p;
    float theta = acosf(clampf(w.z, -1.f, 1.f));
    return env_lookup(e, phi * RT_INV_2PI, theta * RT_INV_PI);
}

// reflection.h:11-69
__device__ __forceinline__ float cos2theta(F3 w) { return w.z * w.z; }
__device__ __forceinline__ float sin2theta(F3 w) { return fmaxf(0.f, 1.f - cos2theta(w)); }
__device__ __forceinline__ float sintheta(F3 w) { return sqrtf(sin2theta(w)); }
__device__ __forceinline__ float tantheta(F3 w) { return sintheta(w) / w.z; }
__device__ __forceinline__ float tan2theta(F3 w) { return sin2theta(w) / cos2theta(w); }
__device__ __forceinline__ float cosphi(F3 w) { float s = sintheta(w); return (s == 0) ? 1.f : clampf(w.x / s, -1.f, 1.f); }
__device__ __forceinline__ float sinphi(F3 w) { float s = sintheta(w); return (s == 0) ? 0.f : clampf(w.y / s, -1.f, 1.f); }
__device__ __forceinline__ bool same_hemisphere(F3 a, F3 b) { return a.z * b.z > 0; }

// sampling.cpp:73-86 + sampling.h:102-106
__device__ __forceinline__ F3 cosine_sample_hemisphere(float u0, float u1) {
    float ox = 2.f * u0 - 1, oy = 2.f * u1 - 1;
    float dx = 0.f, dy = 0.f;
    if (!(ox == 0 && oy == 0)) {
        float theta, r;
        if (fabsf(ox) > fabsf(oy)) { r = ox; theta = RT_PI_OVER_4 * (oy / ox); }
        else { r = oy; theta = RT_PI_OVER_2 - RT_PI_OVER_4 * (ox / oy); }
        float sn, cs;
        sincosf(theta, &sn, &cs);            // one argument reduction for both
        dx = r * cs; dy = r * sn;
    }
    float z = sqrtf(fmaxf(0.f, 1 - dx * dx - dy * dy));
    return f3(dx, dy, z);
}

// reflection.cpp:6-21
__device__ __forceinline__ float fr_dielectric(float cosThetaI, float etaI, float etaT) {
    cosThetaI = clampf(cosThetaI, -1.f, 1.f);
    bool entering = cosThetaI > 0.f;
    if (!entering) { float tmp = etaI; etaI = etaT; etaT = tmp; cosThetaI = fabsf(cosThetaI); }
    float sinThetaI = sqrtf(fmaxf(0.f, 1 - cosThetaI * cosThetaI));
    float sinThetaT = etaI / etaT * sinThetaI;
    if (sinThetaT >= 1) return 1;
    float cosThetaT = sqrtf(fmaxf(0.f, 1 - sinThetaT * sinThetaT));
    float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return (Rparl * Rparl + Rperp * Rperp) / 2;
}
// reflection.cpp:23-42, one channel (etai == 1)
__device__ __forceinline__ float fr_conductor1(float cosThetaI, float etat, float k) {
    cosThetaI = clampf(cosThetaI, -1.f, 1.f);
    float eta = etat / 1.f, etak = k / 1.f;
    float c2 = cosThetaI * cosThetaI;
    float s2 = (float)(1. - c2);
    float eta2 = eta * eta, etak2 = etak * etak;
    float t0 = eta2 - etak2 - s2;
    float a2plusb2 = sqrtf(t0 * t0 + 4 * eta2 * etak2);
    float t1 = a2plusb2 + c2;
    float a = sqrtf(0.5f * (a2plusb2 + t0));
    float t2 = 2.f * cosThetaI * a;
    float Rs = (t1 - t2) / (t1 + t2);
    float t3 = c2 * a2plusb2 + s2 * s2;
    float t4 = t2 * s2;
    float Rp = Rs * (t3 - t4) / (t3 + t4);
    return 0.5f * (Rp + Rs);
}

// TrowbridgeReitzDistribution with visible-area sampling (microfacet.cpp:84-107,149-236; microfacet.h:14-19,59-63)
struct TRDist {
    float ax, ay;
    __device__ __forceinline__ float D(F3 wh) const {
        float t2 = tan2theta(wh);
        if (isinf(t2)) return 0.f;
        float c4 = cos2theta(wh) * cos2theta(wh);
        float cp = cosphi(wh), sp = sinphi(wh);
        float e = ((cp * cp) / (ax * ax) + (sp * sp) / (ay * ay)) * t2;
        return 1 / (RT_PI * ax * ay * c4 * (1 + e) * (1 + e));
    }
    __device__ __forceinline__ float Lambda(F3 w) const {
        float at = fabsf(tantheta(w));
        if (isinf(at)) return 0.f;
        float cp = cosphi(w), sp = sinphi(w);
        float alpha = sqrtf((cp * cp) * ax * ax + (sp * sp) * ay * ay);
        float a2t2 = (alpha * at) * (alpha * at);
        return (-1 + sqrtf(1.f + a2t2)) / 2;
    }
    __device__ __forceinline__ float G1(F3 w) const { return 1 / (1 + Lambda(w)); }
    __device__ __forceinline__ float G(F3 wo, F3 wi) const { return 1 / (1 + Lambda(wo) + Lambda(wi)); }
    __device__ __forceinline__ float Pdf(F3 wo, F3 wh) const { return D(wh) * G1(wo) * fabsf(dot3(wo, wh)) / fabsf(wo.z); }
    __device__ __forceinline__ F3 Sample_wh(F3 wo, float U1, float U2) const {
        bool flip = wo.z < 0;
        F3 wi = flip ? -wo : wo;
        F3 ws = normalize3(f3(ax * wi.x, ay * wi.y, wi.z));
        float cosTheta = ws.z, slope_x, slope_y;
        if (cosTheta > .9999) {
            float r = sqrtf(U1 / (1 - U1));
            float phi = (float)(6.28318530718 * U2);
            float sn, cs;
            sincosf(phi, &sn, &cs);
            slope_x = r * cs; slope_y = r * sn;
        } else {
            float sinTheta = sqrtf(fmaxf(0.f, 1.f - cosTheta * cosTheta));
            float tanTheta = sinTheta / cosTheta;
            float a = 1 / tanTheta;
            float G1v = 2 / (1 + sqrtf(1.f + 1.f / (a * a)));
            float A = 2 * U1 / G1v - 1;
            float tmp = 1.f / (A * A - 1.f);
            if (tmp > 1e10) tmp = 1e10;
            float B = tanTheta;
            float Dd = sqrtf(fmaxf(B * B * tmp * tmp - (A * A - B * B) * tmp, 0.f));
            float sx1 = B * tmp - Dd, sx2 = B * tmp + Dd;
            slope_x = (A < 0 || sx2 > 1.f / tanTheta) ? sx1 : sx2;
            float S;
            if (U2 > 0.5f) { S = 1.f; U2 = 2.f * (U2 - .5f); }
            else { S = -1.f; U2 = 2.f * (.5f - U2); }
            float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) / (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
            slope_y = S * z * sqrtf(1.f + slope_x * slope_x);
        }
        float cp = cosphi(ws), sp = sinphi(ws);
        float tmp2 = cp * slope_x - sp * slope_y;
        slope_y = sp * slope_x + cp * slope_y;
        slope_x = tmp2;
        slope_x = ax * slope_x; slope_y = ay * slope_y;
        F3 wh = normalize3(f3(-slope_x, -slope_y, 1.f));
        if (flip) wh = -wh;
        return wh;
    }
};
__device__ __forceinline__ float roughness_to_alpha(float roughness) {   // microfacet.h:74-78
    roughness = fmaxf(roughness, 1e-3f);
    float x = (float)log((double)roughness);      // the device logf is a few ulp off the host library's; a rounded double log is not
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}

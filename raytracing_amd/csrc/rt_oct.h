// rt_oct.h — closest hit (and any hit) over eight-wide records with quantised boxes, for the order-free walks of large trees: the batch trace
// (k_trace_pairs) and the persistent-lane kernels of the frame (k_extend_pl, k_shadow_pl).
//
// The batch trace of incoherent rays through a tree that does not fit the caches is bound by the number of 128-byte lines it fetches, and
// once the visit order of BVHAccel::Intersect (bvh.cpp:247-282) is given up (rt_traverse2.h: nearest first, near-ties replayed in the
// reference's order) the interior boxes only have to be CONSERVATIVE: a subtree may be entered needlessly, never skipped wrongly.  So one
// line holds three levels of the tree instead of two: a record per interior node at a depth that is a multiple of three, with its (at most
// eight) descendants three levels down as 8-bit boxes relative to the node's own box, each containing the descendant's exact box (checked
// in exact arithmetic when the records are built, rt_scene_create).  What decides a hit stays exact: a leaf is tested only if its exact
// box — rebuilt from its triangles' vertices, which is how the reference computes it (triangle.cpp:45-50, bvh.cpp:30-34,69-71) — passes
// Bounds3::IntersectP as the reference evaluates it (for a plain ray on a tree with exactly nested boxes that is equivalent to all its
// ancestors passing), and the primitive tests are the reference's.
//
// Record (32 words): origin xyz | exponents x, y, z (biased by 128) and count, one byte each | x planes: (lo, hi) byte pairs of children
// 0..7 | y planes | z planes | info words of the children (as in the pair records: record index, or a leaf code) | 8 spare words.
// Plane a of child k lies at origin_a + q * 2^exponent_a.
//
// The slab test runs in the record's own units: t = q * A + B with A = 2^e * invDir and B = (origin - o) * invDir, one fused multiply-add
// per plane.  Its rounding errors are bounded by E = 16 u (|B| + 255 |A|) per axis (u = 2^-24: two roundings in B, one in the
// multiply-add; the factor also covers the reference's own roundings and its widening of the far planes by 1 + 2 gamma(3), geometry.h:902-909);
// near planes are lowered and far planes raised by E, so every comparison errs towards entering the box.
#pragma once
#include "rt_traverse2.h"

// byte selectors for v_perm_b32 over a dword of (lo, hi, lo, hi) plane bytes: as stored, or lo and hi swapped within each pair (a ray
// that runs against the axis enters through the hi plane)
#ifndef RT_OCT_STACK
#define RT_OCT_STACK 16            // LDS ring entries per lane of the eight-wide walk (up to seven pushes per step)
#endif
#ifndef RT_OCT_SECOND_ON_TOP
#define RT_OCT_SECOND_ON_TOP 1
#endif
#define RT_OCT_SEL_KEEP 0x03020100u
#define RT_OCT_SEL_SWAP 0x02030001u

struct OctAxis { float A, Bn, Bf; uint32_t sel; };
__device__ __forceinline__ OctAxis oct_axis(float origin, uint32_t ebiased, float o, float inv) {
    OctAxis x;
    const float step = __uint_as_float((ebiased - 1u) << 23);       // 2^(ebiased - 128)
    x.A = step * inv;
    const float B = (origin - o) * inv;
    const float E = (16.f * 5.9604644775390625e-08f) * (fabsf(B) + 255.f * fabsf(x.A)) + 1e-30f;      // (the constant: a step * invDir that underflows)
    x.Bn = B - E; x.Bf = B + E;
    x.sel = inv < 0.f ? RT_OCT_SEL_SWAP : RT_OCT_SEL_KEEP;
    return x;
}
// Is every component of the ray's reciprocal direction of a size the record-space arithmetic can work with (no overflow of step * invDir,
// no underflow that matters)?  Other rays are walked in the reference's order.
__device__ __forceinline__ bool oct_ray_ok(const LaneTrav &L) {
    const float ax = fabsf(L.invDir.x), ay = fabsf(L.invDir.y), az = fabsf(L.invDir.z);
    return fmaxf(fmaxf(ax, ay), az) < 1e25f && fminf(fminf(ax, ay), az) > 1e-25f;
}
__device__ __forceinline__ float ubyte_f(uint32_t v, int k) { return (float)((v >> (8 * k)) & 255u); }

// One interior step over record `node` for a plain ray whose reciprocal direction is neither tiny nor huge: the children within reach;
// the nearest is entered (or, with push_all, pushed like the others).
template <class Stack>
__device__ __forceinline__ void oct_step_free(const float4 *octs, const LaneTrav &L, uint32_t node, bool push_all, Stack &st, uint32_t &info, bool &have, float &entry) {
    const uint4 *R = (const uint4 *)octs + 8 * (size_t)node;
    const uint4 h = R[0], qx = R[1], qy = R[2], qz = R[3], ia = R[4], ib = R[5];
    const float reach = free_reach(L);
    const OctAxis X = oct_axis(__uint_as_float(h.x), h.w & 255u, L.o.x, L.invDir.x);
    const OctAxis Y = oct_axis(__uint_as_float(h.y), (h.w >> 8) & 255u, L.o.y, L.invDir.y);
    const OctAxis Z = oct_axis(__uint_as_float(h.z), (h.w >> 16) & 255u, L.o.z, L.invDir.z);
    const int n = (int)(h.w >> 24);
    const uint32_t wx[4] = {qx.x, qx.y, qx.z, qx.w}, wy[4] = {qy.x, qy.y, qy.z, qy.w}, wz[4] = {qz.x, qz.y, qz.z, qz.w};
    const uint32_t inf[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
    const float tiny = __uint_as_float(1u);
    // entry distances of all eight children first (all ones: not within reach), the same instructions for every lane
    uint32_t key[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t px = __builtin_amdgcn_perm(wx[j], wx[j], X.sel), py = __builtin_amdgcn_perm(wy[j], wy[j], Y.sel), pz = __builtin_amdgcn_perm(wz[j], wz[j], Z.sel);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int k = 2 * j + c;
            const float nx = fmaf(ubyte_f(px, 2 * c), X.A, X.Bn), fx = fmaf(ubyte_f(px, 2 * c + 1), X.A, X.Bf);
            const float ny = fmaf(ubyte_f(py, 2 * c), Y.A, Y.Bn), fy = fmaf(ubyte_f(py, 2 * c + 1), Y.A, Y.Bf);
            const float nz = fmaf(ubyte_f(pz, 2 * c), Z.A, Z.Bn), fz = fmaf(ubyte_f(pz, 2 * c + 1), Z.A, Z.Bf);
            const float tn = fmaxf(fmaxf(fmaxf(nx, ny), nz), tiny), tf = fminf(fminf(fx, fy), fz);
            const bool ok = k < n && tn <= tf && tn < reach;       // (tn >= tiny > 0, so tn <= tf also says that the box is not behind the ray)
            key[k] = ok ? __float_as_uint(tn) : ~0u;
        }
    }
    // the nearest child (carried with its info word through a min-reduction), then the others onto the stack in one go
    uint32_t bestKey = key[0], bestInfo = inf[0], bestIdx = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) { const bool b = key[k] < bestKey; bestKey = b ? key[k] : bestKey; bestInfo = b ? inf[k] : bestInfo; bestIdx = b ? (uint32_t)k : bestIdx; }
#if RT_OCT_SECOND_ON_TOP
    // ... the second nearest goes on last, so that it is the next to be popped
    uint32_t secKey = ~0u, secInfo = 0, secIdx = 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const bool b = (uint32_t)k != bestIdx && key[k] < secKey; secKey = b ? key[k] : secKey; secInfo = b ? inf[k] : secInfo; secIdx = b ? (uint32_t)k : secIdx; }
#else
    const uint32_t secKey = ~0u, secInfo = 0, secIdx = 8;
#endif
    uint32_t npush = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) npush += (key[k] != ~0u && (uint32_t)k != bestIdx) ? 1u : 0u;
    if (npush) {
        st.reserve((int)npush);
        uint32_t pos = (uint32_t)st.sp;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool p = key[k] != ~0u && (uint32_t)k != bestIdx && (uint32_t)k != secIdx;
            if (p) { st.put((int)pos, (int)inf[k], __uint_as_float(key[k])); pos++; }
        }
        if (secKey != ~0u) { st.put((int)pos, (int)secInfo, __uint_as_float(secKey)); pos++; }
        st.sp = (int)pos;
    }
    if (push_all) {
        if (bestKey != ~0u) st.push((int)bestInfo, __uint_as_float(bestKey));
        have = false; info = 0; entry = 0.f;
        return;
    }
    have = bestKey != ~0u;
    info = bestInfo; entry = __uint_as_float(bestKey);
    if (!have) have = stack_next_free(reach, st, info, entry);
}

// The pending leaf of the eight-wide walk: its exact box first (the union of its triangles' bounds, min / max of the vertices), then, if the
// ray enters that box within reach, its primitives as in lane_leaf_free.
// ANY: BVHAccel::IntersectP (bvh.cpp:284-320) — the ray's tMax is fixed, the first primitive hit ends the walk, and nothing depends on the
// order of the visits (so no tie can arise and the flag is left alone): the answer is whether any primitive of any leaf whose exact box the
// ray enters before tMax is hit.
template <bool ANY = false, class Stack>
__device__ __forceinline__ LeafOutFree oct_leaf_free(const DevScene &sc, const float4 *slots, const LaneTrav &L, Stack &st, const float4 *first = nullptr) {
    LeafOutFree r; r.found = L.found; r.amb = L.amb; r.tmax = L.tmax; r.hitSlot = L.hitSlot; r.b0 = L.b0; r.b1 = L.b1; r.b2 = L.b2;
    float reach = free_reach(L);
    float lox = RT_INF, loy = RT_INF, loz = RT_INF, hix = -RT_INF, hiy = -RT_INF, hiz = -RT_INF;
    float4 k0 = make_float4(0, 0, 0, 0), k1 = k0, k2 = k0;         // the first triangle, kept for the test below
    for (int i = 0; i < L.count; ++i) {
        float4 r0, r1, r2;
        if (first && i == 0) { r0 = first[0]; r1 = first[1]; r2 = first[2]; }
        else { const float4 *s = slots + 3 * (size_t)(L.cur + i); r0 = s[0]; r1 = s[1]; r2 = s[2]; }
        if (i == 0) { k0 = r0; k1 = r1; k2 = r2; }
        lox = fminf(lox, fminf(fminf(r0.x, r0.w), r1.z)); hix = fmaxf(hix, fmaxf(fmaxf(r0.x, r0.w), r1.z));
        loy = fminf(loy, fminf(fminf(r0.y, r1.x), r1.w)); hiy = fmaxf(hiy, fmaxf(fmaxf(r0.y, r1.x), r1.w));
        loz = fminf(loz, fminf(fminf(r0.z, r1.y), r2.x)); hiz = fmaxf(hiz, fmaxf(fmaxf(r0.z, r1.y), r2.x));
    }
    float entry;
    const bool enter = slab_fast(lox, loy, loz, hix, hiy, hiz, L.o, L.invDir, entry) && entry < reach;      // geometry.h:895-914 on the leaf's own box
    if (enter) {
        entry = fmaxf(entry, __uint_as_float(1u));
        for (int i = 0; i < L.count; ++i) {
            const int slot = L.cur + i;
            float4 r0 = k0, r1 = k1, r2 = k2;
            if (i > 0) { const float4 *s = slots + 3 * (size_t)slot; r0 = s[0]; r1 = s[1]; r2 = s[2]; }
            const uint32_t flags = __float_as_uint(r2.y);
            reach = (!ANY && r.found) ? fminf(r.tmax * (1.f + RT_FREE_CULL_BAND), L.t0) : L.t0;
            float t, b0 = 0.f, b1 = 0.f, b2 = 0.f;
            const bool ok = triangle_test_sheared(f3(r0.x, r0.y, r0.z), f3(r0.w, r1.x, r1.y), f3(r1.z, r1.w, r2.x), L.o, L.shear, reach, t, b0, b1, b2) &&
                            (ANY || !(flags & RT_SLOT_REJECT_CLOSEST));
            if (ANY) { if (ok) { r.found = true; break; } }
            else if (ok) {
                if (!r.found || t < r.tmax) {
                    r.amb = (r.found && r.tmax <= t * (1.f + RT_FREE_TIE_BAND)) || entry > t * (1.f + RT_FREE_TIE_BAND);
                    r.found = true; r.tmax = t; r.hitSlot = slot; r.b0 = b0; r.b1 = b1; r.b2 = b2;
                } else if (t <= r.tmax * (1.f + RT_FREE_TIE_BAND)) r.amb = true;
            }
        }
    }
    if (ANY && r.found) { r.have = false; r.info = 0; r.entry = 0.f; return r; }
    reach = (!ANY && r.found) ? fminf(r.tmax * (1.f + RT_FREE_CULL_BAND), L.t0) : L.t0;
    r.have = stack_next_free(reach, st, r.info, r.entry);
    return r;
}

// rt_traverse2.h — persistent-lane BVH traversal over "pair nodes".
//
// Same visit order, slab arithmetic and primitive tests as BVHAccel::Intersect / IntersectP (bvh.cpp:247-320), arranged
// for HBM latency and 64-wide waves:
//   * Pair nodes.  The device keeps one 64-byte record per INTERIOR node holding the boxes of both children, so one
//     memory round trip serves the two slab tests the reference performs for that node's children.  The reference
//     tests the near child at once and the far child when it is popped, against the ray's tMax at that moment; its test
//     is  tMin < ray.tMax && tMax > 0  (geometry.h:913) where tMin and tMax do not depend on ray.tMax.  The far child's
//     tMin is therefore computed now and kept on the stack, and only  tMin < ray.tMax  is evaluated at pop time — the same
//     comparison on the same numbers.  A far child that already fails is never pushed (ray.tMax only shrinks).
//   * Persistent lanes.  A wave owns a contiguous range of rays; a lane whose ray is finished takes the next one, so a wave
//     never idles 63 lanes behind its longest ray.
//   * Leaf postponing.  Lanes that reach a leaf wait until enough lanes hold one (or nobody can walk), then the
//     primitive tests run together.
#pragma once
#include "rt_device.h"

#define RT_PAIR_LEAF 0x80000000u        // child is a leaf with one primitive: low 30 bits = slot
#define RT_PAIR_LEAF_MULTI 0xC0000000u  // child is a leaf with several primitives: low 30 bits = linear node index
#define RT_PAIR_STACK 16                // LDS ring entries per lane (index + tMin)

struct PairStack {
    int *lds_i; float *lds_t;       // &shared[0][threadIdx.x], stride RT_BLOCK
    int *spill_i; float *spill_t;   // &spill[0][global thread], stride spill_stride
    size_t spill_stride;
    int sp, lo;                     // entries [lo, sp) are in the ring
    __device__ __forceinline__ void reset() { sp = 0; lo = 0; }
    __device__ __forceinline__ void push(int info, float t) {
        if (sp - lo == RT_PAIR_STACK) {
            if (lo < RT_MAX_STACK) {
                int k = (lo & (RT_PAIR_STACK - 1)) * RT_BLOCK;
                spill_i[(size_t)lo * spill_stride] = lds_i[k]; spill_t[(size_t)lo * spill_stride] = lds_t[k];
            }
            lo++;
        }
        int k = (sp & (RT_PAIR_STACK - 1)) * RT_BLOCK;
        lds_i[k] = info; lds_t[k] = t;
        sp++;
    }
    __device__ __forceinline__ void pop(int &info, float &t) {
        sp--;
        if (sp < lo) {
            lo = sp;
            if (sp < RT_MAX_STACK) { info = spill_i[(size_t)sp * spill_stride]; t = spill_t[(size_t)sp * spill_stride]; }
            else { info = 0; t = RT_INF; }
            return;
        }
        int k = (sp & (RT_PAIR_STACK - 1)) * RT_BLOCK;
        info = lds_i[k]; t = lds_t[k];
    }
};

// Bounds3::IntersectP(ray, invDir, dirIsNeg) (geometry.h:895-914) without its final  tMin < ray.tMax :
// returns every other condition and hands back tMin.
__device__ __forceinline__ bool slab_partial(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, F3 o, F3 invDir, int negx, int negy,
                                             int negz, float &tMinOut) {
    const float widen = 1 + 2 * ((3 * 5.9604644775390625e-08f) / (1 - 3 * 5.9604644775390625e-08f));   // 1 + 2*gamma(3)
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
    tMinOut = tMin;
    return tMax > 0;
}

// Per-lane traversal state.
struct LaneTrav {
    F3 o, d, invDir;
    float tmax;
    int negx, negy, negz;
    int mode;            // 0 walk (cur = pair index), 1 leaf pending (cur = first slot, count), 2 finished
    int cur, count;
    int hitSlot; float b0, b1, b2;
    bool found;
};

enum { RT_LANE_WALK = 0, RT_LANE_LEAF = 1, RT_LANE_DONE = 2 };

__device__ __forceinline__ void lane_enter(const DevScene &sc, LaneTrav &L, uint32_t info) {
    if (info & RT_PAIR_LEAF) {
        L.mode = RT_LANE_LEAF;
        if ((info & RT_PAIR_LEAF_MULTI) == RT_PAIR_LEAF_MULTI) {
            float4 n1 = sc.nodes[2 * (int)(info & 0x3FFFFFFFu) + 1];
            L.cur = __float_as_int(n1.z); L.count = (int)(__float_as_uint(n1.w) & 0xFFFF);
        } else { L.cur = (int)(info & 0x3FFFFFFFu); L.count = 1; }
    } else { L.mode = RT_LANE_WALK; L.cur = (int)info; }
}
__device__ __forceinline__ void lane_pop_next(const DevScene &sc, LaneTrav &L, PairStack &st) {
    L.mode = RT_LANE_DONE;
    while (st.sp > 0) {
        int info; float t;
        st.pop(info, t);
        if (t < L.tmax) { lane_enter(sc, L, (uint32_t)info); break; }
    }
}
// Start a ray: slab-test the root (the reference's first loop iteration) and enter it.
template <bool COUNT>
__device__ __forceinline__ void lane_begin(const DevScene &sc, LaneTrav &L, PairStack &st, F3 o, F3 d, float tmax, uint32_t &nNodes) {
    L.o = o; L.d = d; L.tmax = tmax;
    L.invDir = f3(1.f / d.x, 1.f / d.y, 1.f / d.z);
    L.negx = L.invDir.x < 0; L.negy = L.invDir.y < 0; L.negz = L.invDir.z < 0;
    L.hitSlot = -1; L.b0 = L.b1 = L.b2 = 0.f; L.found = false;
    st.reset();
    L.mode = RT_LANE_DONE;
    if (sc.n_nodes == 0) return;
    if (COUNT) nNodes++;
    float tMin;
    if (slab_partial(sc.root_box[0], sc.root_box[1], sc.root_box[2], sc.root_box[3], sc.root_box[4], sc.root_box[5], o, L.invDir, L.negx, L.negy, L.negz, tMin) &&
        tMin < tmax)
        lane_enter(sc, L, sc.root_info);
}
// One interior step: both children of pair node L.cur.
template <bool COUNT>
__device__ __forceinline__ void lane_walk(const DevScene &sc, LaneTrav &L, PairStack &st, uint32_t &nNodes) {
    const float4 *P = sc.pairs + 4 * (size_t)L.cur;
    float4 q0 = P[0], q1 = P[1], q2 = P[2], q3 = P[3];
    uint32_t infoL = __float_as_uint(q3.x), infoR = __float_as_uint(q3.y);
    int axis = __float_as_int(q3.z);
    int neg = axis == 0 ? L.negx : (axis == 1 ? L.negy : L.negz);
    float tL, tR;
    bool okL = slab_partial(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, L.o, L.invDir, L.negx, L.negy, L.negz, tL);
    bool okR = slab_partial(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, L.o, L.invDir, L.negx, L.negy, L.negz, tR);
    if (COUNT) nNodes += 2;
    okL = okL && tL < L.tmax; okR = okR && tR < L.tmax;
    // bvh.cpp:266-273: negative direction along the split axis visits the second child first
    bool okN = neg ? okR : okL, okF = neg ? okL : okR;
    uint32_t infoN = neg ? infoR : infoL, infoF = neg ? infoL : infoR;
    float tF = neg ? tL : tR;
    if (okN) { if (okF) st.push((int)infoF, tF); lane_enter(sc, L, infoN); }
    else if (okF) lane_enter(sc, L, infoF);
    else lane_pop_next(sc, L, st);
}
// Test the primitives of the pending leaf (bvh.cpp:258-261 / 294-299), then move on.
template <bool ANY, bool COUNT>
__device__ __forceinline__ void lane_leaf(const DevScene &sc, LaneTrav &L, PairStack &st, uint32_t &nTests) {
    for (int i = 0; i < L.count; ++i) {
        int slot = L.cur + i;
        float4 r0 = sc.slots[3 * (size_t)slot], r1 = sc.slots[3 * (size_t)slot + 1], r2 = sc.slots[3 * (size_t)slot + 2];
        uint32_t flags = __float_as_uint(r2.y);
        if (COUNT) nTests++;
        float t, b0 = 0.f, b1 = 0.f, b2 = 0.f;
        bool ok;
        if (flags & RT_SLOT_SPHERE) ok = sphere_test(sc.spheres + 32 * (size_t)__float_as_int(r2.w), L.o, L.d, L.tmax, t);
        else {
            ok = triangle_test(f3(r0.x, r0.y, r0.z), f3(r0.w, r1.x, r1.y), f3(r1.z, r1.w, r2.x), L.o, L.d, L.tmax, t, b0, b1, b2);
            if (!ANY && (flags & RT_SLOT_REJECT_CLOSEST)) ok = false;
        }
        if (ok) {
            L.found = true;
            if (ANY) { L.mode = RT_LANE_DONE; return; }
            L.tmax = t;                                      // primitive.cpp:39
            L.hitSlot = slot; L.b0 = b0; L.b1 = b1; L.b2 = b2;
        }
    }
    lane_pop_next(sc, L, st);
}

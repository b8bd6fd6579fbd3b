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
//   * Quad nodes.  HBM serves a random 128-byte line as fast as a random 64-byte half line (the request rate, not the byte
//     rate, is the ceiling for scattered reads), so the batch kernels walk TWO tree levels per round trip: one 128-byte
//     record per even-depth interior node holds the boxes of its four grandchildren.  The children's boxes are rebuilt from
//     them (a parent's box is the exact componentwise min/max of its children's, bvh.cpp:30-34), so every slab test the
//     reference evaluates is evaluated here on the same numbers, and the visit order is the reference's.
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
    int cap;                        // entries the spill area holds per lane
    int sp, lo;                     // entries [lo, sp) are in the ring
    __device__ __forceinline__ void reset() { sp = 0; lo = 0; }
    __device__ __forceinline__ void push(int info, float t) {
        if (sp - lo == RT_PAIR_STACK) {
            if (lo < cap) {
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
            if (sp < cap) { info = spill_i[(size_t)sp * spill_stride]; t = spill_t[(size_t)sp * spill_stride]; }
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
// ---- quad nodes ----
// Record layout (8 x float4): boxes of LL, LR, RL, RR (6 floats each: min xyz, max xyz) | info[4] | meta | 3 spare words.
// A child that is itself a leaf occupies the first box of its side; the second info of that side is RT_QUAD_NONE.
// meta: bits 0-1 split axis of the node, 2-3 of its left child, 4-5 of its right child, bit 6 left child is a leaf,
// bit 7 right child is a leaf.
#define RT_QUAD_NONE 0xFFFFFFFFu
struct QuadCand { uint32_t info; float t; bool ok; };

// One side of a quad: child box C = union(A, B) (or A when the child is a leaf).  Produces the side's candidates in visit
// order.  A candidate is kept when every tMax-independent condition of the reference's tests on its ancestors inside the
// record and on itself holds and its own tMin is below the ray's current tMax; tMin travels with the candidate so that
// the comparison against the ray's tMax of the moment the reference would test it is made at pop time.  The child's own
// comparison needs no replay at pop time: tMin(child) <= tMin(grandchild) because the boxes nest and the slab arithmetic is
// monotonic, and tMax only shrinks; a NaN tMin(child) rejects the child for every tMax, here as there.
__device__ __forceinline__ void quad_side(const LaneTrav &L, float a0, float a1, float a2, float a3, float a4, float a5, float b0, float b1, float b2, float b3,
                                          float b4, float b5, uint32_t iA, uint32_t iB, bool leaf, int axis, QuadCand &c0, QuadCand &c1) {
    // Union(b1, b2) with std::min / std::max argument order (geometry.h:833-840)
    float cx0 = b0 < a0 ? b0 : a0, cy0 = b1 < a1 ? b1 : a1, cz0 = b2 < a2 ? b2 : a2;
    float cx1 = a3 < b3 ? b3 : a3, cy1 = a4 < b4 ? b4 : a4, cz1 = a5 < b5 ? b5 : a5;
    float tA, tB, tC;
    bool okA = slab_partial(a0, a1, a2, a3, a4, a5, L.o, L.invDir, L.negx, L.negy, L.negz, tA);
    bool okB = slab_partial(b0, b1, b2, b3, b4, b5, L.o, L.invDir, L.negx, L.negy, L.negz, tB);
    bool okC = slab_partial(cx0, cy0, cz0, cx1, cy1, cz1, L.o, L.invDir, L.negx, L.negy, L.negz, tC);
    okA = okA && tA < L.tmax; okB = okB && tB < L.tmax; okC = okC && tC < L.tmax;
    if (leaf) { c0.info = iA; c0.t = tA; c0.ok = okA; c1.info = RT_QUAD_NONE; c1.t = 0.f; c1.ok = false; return; }
    int neg = axis == 0 ? L.negx : (axis == 1 ? L.negy : L.negz);
    c0.info = neg ? iB : iA; c0.t = neg ? tB : tA; c0.ok = okC && (neg ? okB : okA);
    c1.info = neg ? iA : iB; c1.t = neg ? tA : tB; c1.ok = okC && (neg ? okA : okB);
}

// One interior step over quad node L.cur: the node's two children and four grandchildren.
__device__ __forceinline__ void lane_walk_quad(const DevScene &sc, LaneTrav &L, PairStack &st) {
    const float4 *Q = sc.quads + 8 * (size_t)L.cur;
    float4 q0 = Q[0], q1 = Q[1], q2 = Q[2], q3 = Q[3], q4 = Q[4], q5 = Q[5], q6 = Q[6];
    uint32_t meta = ((const uint32_t *)Q)[28];
    QuadCand l0, l1, r0, r1;
    quad_side(L, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, __float_as_uint(q6.x), __float_as_uint(q6.y), (meta & 64u) != 0,
              (int)((meta >> 2) & 3u), l0, l1);
    quad_side(L, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w, __float_as_uint(q6.z), __float_as_uint(q6.w), (meta & 128u) != 0,
              (int)((meta >> 4) & 3u), r0, r1);
    int axis = (int)(meta & 3u);
    int neg = axis == 0 ? L.negx : (axis == 1 ? L.negy : L.negz);
    // visit order: near child's candidates, then the far child's (bvh.cpp:266-273)
    QuadCand c0 = neg ? r0 : l0, c1 = neg ? r1 : l1, c2 = neg ? l0 : r0, c3 = neg ? l1 : r1;
    int first = c0.ok ? 0 : (c1.ok ? 1 : (c2.ok ? 2 : (c3.ok ? 3 : 4)));
    if (c3.ok && first < 3) st.push((int)c3.info, c3.t);
    if (c2.ok && first < 2) st.push((int)c2.info, c2.t);
    if (c1.ok && first < 1) st.push((int)c1.info, c1.t);
    if (first == 4) lane_pop_next(sc, L, st);
    else lane_enter(sc, L, first == 0 ? c0.info : (first == 1 ? c1.info : (first == 2 ? c2.info : c3.info)));
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

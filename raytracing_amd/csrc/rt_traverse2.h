// rt_traverse2.h — persistent-lane BVH traversal over "pair" and "quad" nodes.
//
// Same visit order, slab arithmetic and primitive tests as BVHAccel::Intersect / IntersectP (bvh.cpp:247-320), arranged
// for HBM latency and 64-wide waves:
//   * Pair nodes.  The device keeps one 64-byte record per INTERIOR node holding the boxes of both children, so one
//     memory round trip serves the two slab tests the reference performs for that node's children.  The reference
//     tests the near child at once and the far child when it is popped, against the ray's tMax at that moment; its test
//     is  tMin < ray.tMax && tMax > 0  (geometry.h:913) where tMin and tMax do not depend on ray.tMax.  The far child's
//     tMin is therefore computed now and kept on the stack, and only  tMin < ray.tMax  is evaluated at pop time — the same
//     comparison on the same numbers.  (It cannot be dropped for failing against the present tMax: a later hit may raise
//     tMax by an ulp, see lane_walk.)
//   * Quad nodes.  HBM serves a random 128-byte line as fast as a random 64-byte half line (the request rate, not the byte
//     rate, is the ceiling for scattered reads), so the batch kernels walk TWO tree levels per round trip: one 128-byte
//     record per even-depth interior node holds the boxes of its four grandchildren.  The children's boxes are rebuilt from
//     them (a parent's box is the exact componentwise min/max of its children's, bvh.cpp:30-34), so every slab test the
//     reference evaluates is evaluated here on the same numbers, and the visit order is the reference's.
//   * Persistent lanes.  A wave owns a contiguous range of rays; a lane whose ray is finished takes the next one, so a wave
//     never idles 63 lanes behind its longest ray.
//   * Leaf postponing.  Lanes that reach a leaf wait until enough lanes hold one (or nobody can walk), then the
//     primitive tests run together.
// The per-lane state is written so that every field is assigned on one control path per step: stores to different fields
// on different branches get merged by the compiler into one store with a computed address, which pins the whole struct in
// scratch memory.
#pragma once
#include "rt_device.h"

#define RT_PAIR_LEAF 0x80000000u        // child is a leaf with one primitive: low 30 bits = slot
#define RT_PAIR_LEAF_MULTI 0xC0000000u  // child is a leaf with several primitives: bit 29 clear: low 29 bits = linear node index;
#define RT_PAIR_LEAF_INLINE 0x20000000u //   bit 29 set: bits 22-28 = count - 1, bits 0-21 = first slot (no node read on entering)
#define RT_PAIR_STACK 16                // LDS ring entries per lane (index + tMin), pair kernels
#define RT_QUAD_STACK 8                 // same, quad kernels (half the levels; a smaller ring lets more blocks share a CU)

// SPILL = false: the caller guarantees that no more than RING entries are ever live (the tree is at most RING interior levels
// deep), and the stack never touches global memory — which also keeps vector-memory waits out of the loops that use it.
template <int RING, bool SPILL = true>
struct PairStackT {
    int *lds_i; float *lds_t;       // &shared[0][threadIdx.x], stride RT_BLOCK
    int *spill_i; float *spill_t;   // &spill[0][global thread], stride spill_stride
    size_t spill_stride;
    int cap;                        // entries the spill area holds per lane
    int sp, lo;                     // entries [lo, sp) are in the ring
    __device__ __forceinline__ void reset() { sp = 0; lo = 0; }
    __device__ __forceinline__ void push(int info, float t) {
        if (SPILL && sp - lo == RING) {
            if (lo < cap) {
                int k = (lo & (RING - 1)) * RT_BLOCK;
                spill_i[(size_t)lo * spill_stride] = lds_i[k]; spill_t[(size_t)lo * spill_stride] = lds_t[k];
            }
            lo++;
        }
        int k = (sp & (RING - 1)) * RT_BLOCK;
        lds_i[k] = info; lds_t[k] = t;
        sp++;
    }
    // Several pushes at once: reserve(n) moves the oldest ring entries to the spill area until n more fit (n <= RING), put() then writes
    // entry number `at` (sp .. sp + n - 1) without looking at the ring's fill, and the caller sets sp.
    __device__ __forceinline__ void reserve(int n) {
        while (SPILL && sp + n - lo > RING) {
            if (lo < cap) {
                int k = (lo & (RING - 1)) * RT_BLOCK;
                spill_i[(size_t)lo * spill_stride] = lds_i[k]; spill_t[(size_t)lo * spill_stride] = lds_t[k];
            }
            lo++;
        }
    }
    __device__ __forceinline__ void put(int at, int info, float t) {
        int k = (at & (RING - 1)) * RT_BLOCK;
        lds_i[k] = info; lds_t[k] = t;
    }
    __device__ __forceinline__ void pop(int &info, float &t) {
        sp--;
        if (SPILL && sp < lo) {
            lo = sp;
            if (sp < cap) { info = spill_i[(size_t)sp * spill_stride]; t = spill_t[(size_t)sp * spill_stride]; }
            else { info = 0; t = RT_INF; }
            return;
        }
        int k = (sp & (RING - 1)) * RT_BLOCK;
        info = lds_i[k]; t = lds_t[k];
    }
};

// Bounds3::IntersectP(ray, invDir, dirIsNeg) (geometry.h:895-914) without its final  tMin < ray.tMax :
// returns every other condition and hands back tMin.  Written without early returns (a failed test keeps computing and
// reports false), which changes no result and keeps the wave on one path.
__device__ __forceinline__ bool slab_partial(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, F3 o, F3 invDir, bool negx, bool negy,
                                             bool negz, float &tMinOut) {
    const float widen = 1 + 2 * ((3 * 5.9604644775390625e-08f) / (1 - 3 * 5.9604644775390625e-08f));   // 1 + 2*gamma(3)
    float tMin = ((negx ? bmaxx : bminx) - o.x) * invDir.x;
    float tMax = ((negx ? bminx : bmaxx) - o.x) * invDir.x;
    float tyMin = ((negy ? bmaxy : bminy) - o.y) * invDir.y;
    float tyMax = ((negy ? bminy : bmaxy) - o.y) * invDir.y;
    tMax *= widen;
    tyMax *= widen;
    bool ok = !(tMin > tyMax || tyMin > tMax);
    tMin = tyMin > tMin ? tyMin : tMin;
    tMax = tyMax < tMax ? tyMax : tMax;
    float tzMin = ((negz ? bmaxz : bminz) - o.z) * invDir.z;
    float tzMax = ((negz ? bminz : bmaxz) - o.z) * invDir.z;
    tzMax *= widen;
    ok = ok && !(tMin > tzMax || tzMin > tMax);
    tMin = tzMin > tMin ? tzMin : tMin;
    tMax = tzMax < tMax ? tzMax : tMax;
    tMinOut = tMin;
    return ok && tMax > 0;
}
// The same test for a ray whose origin is finite and whose reciprocal direction is finite and non-zero in every component.
// No NaN can then arise (NaN needs 0 * inf), selecting the near/far plane by the direction's sign equals taking the min/max
// of the two plane distances, and the chain of pairwise rejections equals  max(near) <= min(far)  together with
// min(far) > 0  (a same-axis near > far only happens when far < 0, which fails the last condition anyway).
__device__ __forceinline__ bool slab_fast(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, F3 o, F3 invDir, float &tMinOut) {
    const float widen = 1 + 2 * ((3 * 5.9604644775390625e-08f) / (1 - 3 * 5.9604644775390625e-08f));
    float x0 = (bminx - o.x) * invDir.x, x1 = (bmaxx - o.x) * invDir.x;
    float y0 = (bminy - o.y) * invDir.y, y1 = (bmaxy - o.y) * invDir.y;
    float z0 = (bminz - o.z) * invDir.z, z1 = (bmaxz - o.z) * invDir.z;
    // the far planes are widened after the minimum is taken: x -> fl(x * widen) is monotonic, so min(fl(a w), fl(b w), fl(c w)) is
    // fl(min(a, b, c) w), bit for bit, and two multiplications are saved
    float nx = fminf(x0, x1), fx = fmaxf(x0, x1);
    float ny = fminf(y0, y1), fy = fmaxf(y0, y1);
    float nz = fminf(z0, z1), fz = fmaxf(z0, z1);
    float tMin = fmaxf(fmaxf(nx, ny), nz), tMax = fminf(fminf(fx, fy), fz) * widen;
    tMinOut = tMin;
    return tMin <= tMax && tMax > 0;
}

// Per-lane traversal state.
struct LaneTrav {
    F3 o, d, invDir;
    float tmax;
    bool negx, negy, negz;
    uint32_t sign;       // bit k: invDir[k] < 0 (one word, so that picking the bit of a split axis is arithmetic, not an indexed read)
    bool plain;          // slab_fast applies to this ray
    int mode;            // 0 walk (cur = record index), 1 leaf pending (cur = first slot, count), 2 finished
    int cur, count;
    int hitSlot; float b0, b1, b2;
    bool found;
    RayShear shear;      // per-ray constants of the triangle test
    // order-free closest-hit walks (lane_walk_quad_free / lane_leaf_free) only:
    float t0;            // the ray's tMax as it came in (tmax holds the best t found so far once found is set)
    float entry;         // box entry distance of the leaf that is entered / being tested (at least the smallest positive number)
    bool amb;            // the answer so far is not safe against the reference's visit order: replay the ray in that order
};

enum { RT_LANE_WALK = 0, RT_LANE_LEAF = 1, RT_LANE_DONE = 2 };

// Enter the node an info word names (or finish when there is none).
__device__ __forceinline__ void lane_enter(const DevScene &sc, LaneTrav &L, uint32_t info, bool have) {
    bool leaf = (info & RT_PAIR_LEAF) != 0, multi = (info & RT_PAIR_LEAF_MULTI) == RT_PAIR_LEAF_MULTI;
    const bool inl = multi && (info & RT_PAIR_LEAF_INLINE) != 0;
    int cur = leaf ? (int)(info & 0x3FFFFFFFu) : (int)info, count = 1;
    if (inl) { count = (int)((info >> 22) & 127u) + 1; cur = (int)(info & 0x3FFFFFu); }
    else if (have && multi) {
        float4 n1 = sc.nodes[2 * (size_t)cur + 1];
        cur = __float_as_int(n1.z); count = (int)(__float_as_uint(n1.w) & 0xFFFF);
    }
    L.mode = !have ? RT_LANE_DONE : (leaf ? RT_LANE_LEAF : RT_LANE_WALK);
    L.cur = cur; L.count = count;
}
// Pop until an entry whose tMin is still below the ray's tMax turns up.
template <class Stack>
__device__ __forceinline__ bool stack_next(const LaneTrav &L, Stack &st, uint32_t &info) {
    bool have = false;
    info = 0;
    while (!have && st.sp > 0) {
        int i; float t;
        st.pop(i, t);
        if (t < L.tmax) { info = (uint32_t)i; have = true; }
    }
    return have;
}
__device__ __forceinline__ bool finite_nonzero(float v) { float a = fabsf(v); return a > 0.f && a < RT_INF; }
// The per-ray constants of a walk (bvh.cpp:250-252, triangle.cpp:62-72).
__device__ __forceinline__ void lane_setup(LaneTrav &L, F3 o, F3 d, float tmax) {
    L.o = o; L.d = d; L.tmax = tmax;
    L.invDir = f3(1.f / d.x, 1.f / d.y, 1.f / d.z);
    L.negx = L.invDir.x < 0; L.negy = L.invDir.y < 0; L.negz = L.invDir.z < 0;
    L.sign = (L.negx ? 1u : 0u) | (L.negy ? 2u : 0u) | (L.negz ? 4u : 0u);
    L.plain = finite_nonzero(L.invDir.x) && finite_nonzero(L.invDir.y) && finite_nonzero(L.invDir.z) && fabsf(o.x) < RT_INF && fabsf(o.y) < RT_INF &&
              fabsf(o.z) < RT_INF;
    L.hitSlot = -1; L.b0 = L.b1 = L.b2 = 0.f; L.found = false;
    L.shear = ray_shear(d, L.invDir);
    L.t0 = tmax; L.entry = 0.f; L.amb = false;
}
// Start a ray: slab-test the root (the reference's first loop iteration) and enter it.
// Returns whether the root is to be entered (info word sc.root_info).
// For a plain ray and an interior root the test is left out: whatever the walk does next is decided by the boxes of the root's
// children (pair records) or grandchildren (quad records), and one of those can only pass if the root's box — their exact
// componentwise union, under monotonic slab arithmetic — passes too, tMin < tMax included; a ray that misses the root then ends
// after one step instead of none, with the same result.  A tree handed in by the caller may carry a root box that is not that
// union (a tighter one, say): sc.root_implied says whether it is, and without it the test is made.  (The counting variant keeps
// the test: it counts the reference's slab tests.)
template <bool COUNT, class Stack>
__device__ __forceinline__ bool lane_begin(const DevScene &sc, LaneTrav &L, Stack &st, F3 o, F3 d, float tmax, uint32_t &nNodes) {
    lane_setup(L, o, d, tmax);
    st.reset();
    bool have = false;
    if (sc.n_nodes != 0) {
        if (COUNT) nNodes++;
        if (!COUNT && L.plain && sc.root_implied && (sc.root_info & RT_PAIR_LEAF) == 0) have = true;
        else {
            float tMin;
            have = slab_partial(sc.root_box[0], sc.root_box[1], sc.root_box[2], sc.root_box[3], sc.root_box[4], sc.root_box[5], o, L.invDir, L.negx, L.negy, L.negz, tMin) &&
                   tMin < tmax;
        }
    }
    return have;
}
// One interior step: both children of pair node L.cur.  Hands back the info word of the node to enter next (have = false:
// the ray is finished).
// KSTRIDE = 1: records stored one after the other (4 float4 each).  KSTRIDE = n: the k-th float4 of record r is at
// pairs[k * n + r] — the layout of the LDS copy, where consecutive records then fall into different banks.
// FIXED_TMAX: the ray's tMax never changes (any-hit rays: the first accepted primitive ends the walk), so the far child's test against
// it is final now: a child that fails is not pushed, and a popped entry needs no second look.
template <bool COUNT, bool PLAIN, int KSTRIDE = 1, bool FIXED_TMAX = false, class Stack>
__device__ __forceinline__ void lane_walk(const float4 *pairs, const LaneTrav &L, Stack &st, uint32_t &nNodes, uint32_t &info, bool &have) {
    const float4 *P = pairs + (KSTRIDE == 1 ? 4 : 1) * (size_t)L.cur;
    float4 q0 = P[0], q1 = P[KSTRIDE], q2 = P[2 * KSTRIDE], q3 = P[3 * KSTRIDE];
    uint32_t infoL = __float_as_uint(q3.x), infoR = __float_as_uint(q3.y);
    int axis = __float_as_int(q3.z);
    bool neg = ((L.sign >> axis) & 1u) != 0;
    float tL, tR;
    bool okL, okR;
    // (Packed two-float arithmetic, v_pk_add_f32 / v_pk_mul_f32 on an x/y-major record layout, was measured here: 15 packed
    // instructions in place of 30 scalar ones per step, and the frame ran 2.7 % slower.)
    if (PLAIN) {
        okL = slab_fast(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, L.o, L.invDir, tL);
        okR = slab_fast(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, L.o, L.invDir, tR);
    } else {
        okL = slab_partial(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, L.o, L.invDir, L.negx, L.negy, L.negz, tL);
        okR = slab_partial(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, L.o, L.invDir, L.negx, L.negy, L.negz, tR);
    }
    if (COUNT) nNodes += 2;
    // bvh.cpp:266-273: negative direction along the split axis visits the second child first
    bool okN = neg ? okR : okL, okF = neg ? okL : okR;
    uint32_t infoN = neg ? infoR : infoL, infoF = neg ? infoL : infoR;
    float tN = neg ? tR : tL, tF = neg ? tL : tR;
    // The near child is tested now, against the ray's present tMax.  The far child is tested when popped, against the tMax
    // of that moment, so it is pushed whenever its tMax-independent part holds: a hit may RAISE tMax by an ulp (the
    // reference accepts  tScaled <= tMax * det  and then stores the rounded  tScaled / det ), so "fails now" does not imply
    // "fails later".  If the near child is rejected the far child is what the reference pops next, at the same tMax.
    bool nearNow = okN && tN < L.tmax, farNow = okF && tF < L.tmax;
    if (nearNow && (FIXED_TMAX ? farNow : okF)) st.push((int)infoF, tF);
    info = nearNow ? infoN : infoF;
    have = nearNow || farNow;
    if (!have) {
        if (FIXED_TMAX) { have = st.sp > 0; info = 0; if (have) { int i; float t; st.pop(i, t); info = (uint32_t)i; } }
        else have = stack_next(L, st, info);
    }
}

// ---- quad nodes ----
// Record layout (8 x float4): boxes of LL, LR, RL, RR (6 floats each: min xyz, max xyz) | info[4] | meta | 3 spare words.
// A child that is itself a leaf occupies both boxes of its side (so their union is the child's box); its second info is
// RT_QUAD_NONE.  meta: bits 0-1 split axis of the node, 2-3 of its left child, 4-5 of its right child, bit 6 left child
// is a leaf, bit 7 right child is a leaf.
#define RT_QUAD_NONE 0xFFFFFFFFu
struct QuadCand { uint32_t info; float t; bool indep, now; };

// One side of a quad: child box C = union(A, B).  Produces the side's two candidates in visit order.  For each:
//   indep — every tMax-independent condition of the reference's tests on the child and on the candidate holds;
//   now   — indep, and both tMin values are below the ray's present tMax (the tests the reference would make at this
//           moment if it reached the candidate without any primitive test in between).
// The candidate's own tMin travels with it to the stack so that the comparison against the ray's tMax of the moment the
// reference would test it is made at pop time.  The child's own comparison needs no replay then: tMin(child) <=
// tMin(grandchild) because the boxes nest and the slab arithmetic is monotonic, so a child the reference rejects at some tMax
// has both grandchildren rejected at that same tMax (nothing is tested in between), and an accepted child leaves the
// decision to the grandchild's own comparison; a NaN tMin(child) rejects the child for every tMax, here as there.  For a
// plain ray (slab_fast) the child's test is implied by the grandchild's for the same reason and is skipped.
template <bool PLAIN>
__device__ __forceinline__ void quad_side(const LaneTrav &L, float a0, float a1, float a2, float a3, float a4, float a5, float b0, float b1, float b2, float b3,
                                          float b4, float b5, uint32_t iA, uint32_t iB, bool leaf, int axis, QuadCand &c0, QuadCand &c1) {
    float tA, tB;
    bool okA, okB, nowC = true;
    if (PLAIN) {
        okA = slab_fast(a0, a1, a2, a3, a4, a5, L.o, L.invDir, tA);
        okB = slab_fast(b0, b1, b2, b3, b4, b5, L.o, L.invDir, tB);
    } else {
        // Union(b1, b2) with std::min / std::max argument order (geometry.h:833-840)
        float cx0 = b0 < a0 ? b0 : a0, cy0 = b1 < a1 ? b1 : a1, cz0 = b2 < a2 ? b2 : a2;
        float cx1 = a3 < b3 ? b3 : a3, cy1 = a4 < b4 ? b4 : a4, cz1 = a5 < b5 ? b5 : a5;
        float tC;
        okA = slab_partial(a0, a1, a2, a3, a4, a5, L.o, L.invDir, L.negx, L.negy, L.negz, tA);
        okB = slab_partial(b0, b1, b2, b3, b4, b5, L.o, L.invDir, L.negx, L.negy, L.negz, tB);
        bool okC = slab_partial(cx0, cy0, cz0, cx1, cy1, cz1, L.o, L.invDir, L.negx, L.negy, L.negz, tC) && tC == tC;
        okA = okA && okC; okB = okB && okC;
        nowC = tC < L.tmax;
    }
    okB = okB && !leaf;
    bool nowA = okA && nowC && tA < L.tmax, nowB = okB && nowC && tB < L.tmax;
    bool neg = ((L.sign >> axis) & 1u) != 0;
    c0.info = neg ? iB : iA; c0.t = neg ? tB : tA; c0.indep = neg ? okB : okA; c0.now = neg ? nowB : nowA;
    c1.info = neg ? iA : iB; c1.t = neg ? tA : tB; c1.indep = neg ? okA : okB; c1.now = neg ? nowA : nowB;
}

// One interior step over quad node L.cur: the node's two children and four grandchildren.
template <bool PLAIN, bool FIXED_TMAX = false, class Stack>
__device__ __forceinline__ void lane_walk_quad(const float4 *quads, const LaneTrav &L, Stack &st, uint32_t &info, bool &have) {
    const float4 *Q = quads + 8 * (size_t)L.cur;
    float4 q0 = Q[0], q1 = Q[1], q2 = Q[2], q3 = Q[3], q4 = Q[4], q5 = Q[5], q6 = Q[6];
    uint32_t meta = ((const uint32_t *)Q)[28];
    QuadCand l0, l1, r0, r1;
    quad_side<PLAIN>(L, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, __float_as_uint(q6.x), __float_as_uint(q6.y), (meta & 64u) != 0,
                     (int)((meta >> 2) & 3u), l0, l1);
    quad_side<PLAIN>(L, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w, __float_as_uint(q6.z), __float_as_uint(q6.w), (meta & 128u) != 0,
                     (int)((meta >> 4) & 3u), r0, r1);
    int axis = (int)(meta & 3u);
    bool neg = ((L.sign >> axis) & 1u) != 0;
    // visit order: near child's candidates, then the far child's (bvh.cpp:266-273).  The reference reaches the candidates
    // one after the other with no primitive test in between until one is accepted, so up to the first accepted candidate
    // the tests are made now; the candidates after it are tested when popped (a hit in between may lower tMax, or raise
    // it by an ulp), so they are pushed on their tMax-independent part alone.
    QuadCand c0 = neg ? r0 : l0, c1 = neg ? r1 : l1, c2 = neg ? l0 : r0, c3 = neg ? l1 : r1;
    int first = c0.now ? 0 : (c1.now ? 1 : (c2.now ? 2 : (c3.now ? 3 : 4)));
    // (FIXED_TMAX, as in lane_walk: a candidate's test against the ray's unchanging tMax is final, so only those that pass are pushed)
    if ((FIXED_TMAX ? c3.now : c3.indep) && first < 3) st.push((int)c3.info, c3.t);
    if ((FIXED_TMAX ? c2.now : c2.indep) && first < 2) st.push((int)c2.info, c2.t);
    if ((FIXED_TMAX ? c1.now : c1.indep) && first < 1) st.push((int)c1.info, c1.t);
    info = first == 0 ? c0.info : (first == 1 ? c1.info : (first == 2 ? c2.info : c3.info));
    have = first < 4;
    if (!have) {
        if (FIXED_TMAX) { have = st.sp > 0; info = 0; if (have) { int i; float t; st.pop(i, t); info = (uint32_t)i; } }
        else have = stack_next(L, st, info);
    }
}

// Test the primitives of the pending leaf (bvh.cpp:258-261 / 294-299) and pick the next node.
struct LeafOut { bool found; float tmax; int hitSlot; float b0, b1, b2; uint32_t info; bool have; };
// first: the three float4 of the leaf's first slot when the caller has loaded them already (so that the load could be issued ahead of
// other work), else nullptr.
template <bool ANY, bool COUNT, bool FIXED_POP = false, class Stack>
__device__ __forceinline__ LeafOut lane_leaf(const DevScene &sc, const float4 *slots, const LaneTrav &L, Stack &st, uint32_t &nTests, const float4 *first = nullptr) {
    LeafOut r; r.found = L.found; r.tmax = L.tmax; r.hitSlot = L.hitSlot; r.b0 = L.b0; r.b1 = L.b1; r.b2 = L.b2;
    bool stop = false;
    for (int i = 0; i < L.count && !stop; ++i) {
        int slot = L.cur + i;
        float4 r0, r1, r2;
        if (first && i == 0) { r0 = first[0]; r1 = first[1]; r2 = first[2]; }
        else { r0 = slots[3 * (size_t)slot]; r1 = slots[3 * (size_t)slot + 1]; r2 = slots[3 * (size_t)slot + 2]; }
        uint32_t flags = __float_as_uint(r2.y);
        if (COUNT) nTests++;
        float t, b0 = 0.f, b1 = 0.f, b2 = 0.f;
        bool ok;
        if (flags & RT_SLOT_SPHERE) ok = sphere_test(sc.spheres + 32 * (size_t)__float_as_int(r2.w), L.o, L.d, r.tmax, t);
        else {
            ok = triangle_test_sheared(f3(r0.x, r0.y, r0.z), f3(r0.w, r1.x, r1.y), f3(r1.z, r1.w, r2.x), L.o, L.shear, r.tmax, t, b0, b1, b2);
            if (!ANY && (flags & RT_SLOT_REJECT_CLOSEST)) ok = false;
        }
        if (ok) {
            r.found = true;
            if (ANY) stop = true;
            else { r.tmax = t; r.hitSlot = slot; r.b0 = b0; r.b1 = b1; r.b2 = b2; }     // primitive.cpp:39
        }
    }
    r.info = 0; r.have = false;
    if (!stop && ANY && FIXED_POP) {      // every entry passed its test against the (fixed) tMax when it was pushed
        if (st.sp > 0) { int i; float t; st.pop(i, t); r.info = (uint32_t)i; r.have = true; }
    } else if (!stop) {          // stack entries are compared with the shrunken tMax
        while (!r.have && st.sp > 0) {
            int i; float t;
            st.pop(i, t);
            if (t < r.tmax) { r.info = (uint32_t)i; r.have = true; }
        }
    }
    return r;
}

// ---- closest hit without the reference's visit order (quad records) ----
// BVHAccel::Intersect (bvh.cpp:247-282) returns the primitive with the smallest t; the order of its visits matters only when two
// candidates are nearly tied — at equal t the later-tested primitive wins (triangle.cpp:104-111 rejects only '>'), an accepted hit may
// raise tMax by an ulp (tScaled <= tMax * det is tested, the rounded tScaled / det is stored), and a box entry distance next to a hit
// distance can fall on either side of '<' (geometry.h:913).  Every one of these needs a second candidate — another primitive's t, or the
// entry distance of the accepted primitive's own leaf box — within a few ulps of the smallest t.  A primitive's t and barycentrics do
// not depend on the ray's tMax, only its acceptance does.  So this walk visits nearest boxes first, whatever the split axes say, with bands:
//   * a box is entered unless its entry distance is at or beyond the reach, min(best t so far * (1 + RT_FREE_CULL_BAND), the ray's own
//     tMax) — the ray's own tMax is never widened: what the reference would not accept against it is no candidate;
//   * primitives are tested against that reach; the best t is kept, and a flag says whether any other candidate lies within
//     RT_FREE_TIE_BAND (relative) of it: a new best clears the flag unless the old best is within the band of it (every earlier
//     candidate is at or beyond the old best), a candidate that is not better sets it when it is within the band of the best;
//   * the flag is also set when the accepted primitive's own leaf box is entered beyond its t (the reference might have culled that
//     leaf against another candidate's t), and from the start for rays that are not plain (zero direction components: NaN slabs);
//   * a flagged ray is walked again in the reference's order by the caller.
// Exactness rests on nested boxes (a leaf's box passes the tMax-independent part of Bounds3::IntersectP exactly when its ancestors'
// do, for a plain ray — the quad records are only built for such trees), on monotonic acceptance (accepted against some tMax,
// accepted against any larger), and on one assumption: no primitive's computed t lies below the entry distance of a box that holds it
// by more than the difference of the bands (0.1 %), so that a subtree skipped here cannot hold a hit the reference, arriving there
// with a larger tMax, would have accepted below the best t found here.
#ifndef RT_FREE_CULL_BAND
#define RT_FREE_CULL_BAND 1e-3f
#endif
#ifndef RT_FREE_TIE_BAND
#define RT_FREE_TIE_BAND 1e-5f
#endif
__device__ __forceinline__ float free_reach(const LaneTrav &L) { return L.found ? fminf(L.tmax * (1.f + RT_FREE_CULL_BAND), L.t0) : L.t0; }
// Pop until an entry whose entry distance is within reach turns up.
template <class Stack>
__device__ __forceinline__ bool stack_next_free(float reach, Stack &st, uint32_t &info, float &entry) {
    bool have = false;
    info = 0; entry = 0.f;
    while (!have && st.sp > 0) {
        int i; float t;
        st.pop(i, t);
        if (t < reach) { info = (uint32_t)i; entry = t; have = true; }
    }
    return have;
}
// Start a ray of the order-free walk: as lane_begin, except that a ray that is not plain is flagged and not walked at all.
template <class Stack>
__device__ __forceinline__ bool lane_begin_free(const DevScene &sc, LaneTrav &L, Stack &st, F3 o, F3 d, float tmax) {
    uint32_t dummy = 0;
    bool have = lane_begin<false>(sc, L, st, o, d, tmax, dummy);
    if (!L.plain) { L.amb = true; have = false; }
    return have;
}
// One interior step over quad node L.cur for a plain ray: the four grandchildren within reach, nearest first.
// node: the record to expand (L.cur for a walking lane).  push_all: the nearest grandchild goes on the stack with the others instead of
// being entered (a lane that holds a leaf still to be tested walks ahead this way, see trace_share).
template <class Stack>
__device__ __forceinline__ void quad_step_free(const float4 *quads, const LaneTrav &L, uint32_t node, bool push_all, Stack &st, uint32_t &info, bool &have, float &entry) {
    const float4 *Q = quads + 8 * (size_t)node;
    float4 q0 = Q[0], q1 = Q[1], q2 = Q[2], q3 = Q[3], q4 = Q[4], q5 = Q[5], q6 = Q[6];
    const uint32_t meta = ((const uint32_t *)Q)[28];
    const float reach = free_reach(L);
    float t0, t1, t2, t3;
    bool ok0 = slab_fast(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, L.o, L.invDir, t0);
    bool ok1 = slab_fast(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, L.o, L.invDir, t1) && (meta & 64u) == 0;      // a leaf child fills both boxes of its side
    bool ok2 = slab_fast(q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, L.o, L.invDir, t2);
    bool ok3 = slab_fast(q4.z, q4.w, q5.x, q5.y, q5.z, q5.w, L.o, L.invDir, t3) && (meta & 128u) == 0;
    // sort the four (entry distance, info word) pairs by distance with a five-comparator network on the distances' bit patterns (never
    // negative: max with the smallest positive number; a candidate out of reach sorts last as all ones).  Compare-and-select on both
    // halves of a pair: selecting the info words through the position instead makes the compiler build a lookup table in scratch memory.
    const float tiny = __uint_as_float(1u);
    uint32_t ka = (ok0 && t0 < reach) ? __float_as_uint(fmaxf(t0, tiny)) : ~0u, ia = __float_as_uint(q6.x);
    uint32_t kb = (ok1 && t1 < reach) ? __float_as_uint(fmaxf(t1, tiny)) : ~0u, ib = __float_as_uint(q6.y);
    uint32_t kc = (ok2 && t2 < reach) ? __float_as_uint(fmaxf(t2, tiny)) : ~0u, ic = __float_as_uint(q6.z);
    uint32_t kd = (ok3 && t3 < reach) ? __float_as_uint(fmaxf(t3, tiny)) : ~0u, id = __float_as_uint(q6.w);
    auto order = [](uint32_t &k0_, uint32_t &i0_, uint32_t &k1_, uint32_t &i1_) {
        const bool sw = k1_ < k0_;
        const uint32_t lk = sw ? k1_ : k0_, hk = sw ? k0_ : k1_, li = sw ? i1_ : i0_, hi = sw ? i0_ : i1_;
        k0_ = lk; k1_ = hk; i0_ = li; i1_ = hi;
    };
    order(ka, ia, kb, ib); order(kc, ic, kd, id); order(ka, ia, kc, ic); order(kb, ib, kd, id); order(kb, ib, kc, ic);
    if (kd != ~0u) st.push((int)id, __uint_as_float(kd));
    if (kc != ~0u) st.push((int)ic, __uint_as_float(kc));
    if (kb != ~0u) st.push((int)ib, __uint_as_float(kb));
    if (push_all) {
        if (ka != ~0u) st.push((int)ia, __uint_as_float(ka));
        have = false; info = 0; entry = 0.f;
        return;
    }
    have = ka != ~0u;
    info = ia; entry = __uint_as_float(ka);
    if (!have) have = stack_next_free(reach, st, info, entry);
}
template <class Stack>
__device__ __forceinline__ void lane_walk_quad_free(const float4 *quads, const LaneTrav &L, Stack &st, uint32_t &info, bool &have, float &entry) {
    quad_step_free(quads, L, (uint32_t)L.cur, false, st, info, have, entry);
}
// Test the primitives of the pending leaf (entered at distance <= L.entry) and pick the next node.
struct LeafOutFree { bool found, amb; float tmax; int hitSlot; float b0, b1, b2; uint32_t info; bool have; float entry; };
template <class Stack>
__device__ __forceinline__ LeafOutFree lane_leaf_free(const DevScene &sc, const float4 *slots, const LaneTrav &L, Stack &st, const float4 *first = nullptr) {
    LeafOutFree r; r.found = L.found; r.amb = L.amb; r.tmax = L.tmax; r.hitSlot = L.hitSlot; r.b0 = L.b0; r.b1 = L.b1; r.b2 = L.b2;
    for (int i = 0; i < L.count; ++i) {
        int slot = L.cur + i;
        float4 r0, r1, r2;
        if (first && i == 0) { r0 = first[0]; r1 = first[1]; r2 = first[2]; }
        else { r0 = slots[3 * (size_t)slot]; r1 = slots[3 * (size_t)slot + 1]; r2 = slots[3 * (size_t)slot + 2]; }
        const uint32_t flags = __float_as_uint(r2.y);
        const float reach = r.found ? fminf(r.tmax * (1.f + RT_FREE_CULL_BAND), L.t0) : L.t0;
        float t, b0 = 0.f, b1 = 0.f, b2 = 0.f;
        bool ok;
        if (flags & RT_SLOT_SPHERE) ok = sphere_test(sc.spheres + 32 * (size_t)__float_as_int(r2.w), L.o, L.d, reach, t);
        else ok = triangle_test_sheared(f3(r0.x, r0.y, r0.z), f3(r0.w, r1.x, r1.y), f3(r1.z, r1.w, r2.x), L.o, L.shear, reach, t, b0, b1, b2) &&
                  !(flags & RT_SLOT_REJECT_CLOSEST);
        if (ok) {
            if (!r.found || t < r.tmax) {
                r.amb = (r.found && r.tmax <= t * (1.f + RT_FREE_TIE_BAND)) || L.entry > t * (1.f + RT_FREE_TIE_BAND);
                r.found = true; r.tmax = t; r.hitSlot = slot; r.b0 = b0; r.b1 = b1; r.b2 = b2;
            } else if (t <= r.tmax * (1.f + RT_FREE_TIE_BAND)) r.amb = true;
        }
    }
    const float reach = r.found ? fminf(r.tmax * (1.f + RT_FREE_CULL_BAND), L.t0) : L.t0;
    r.have = stack_next_free(reach, st, r.info, r.entry);
    return r;
}

// (test builds: -DRT_ANY_LIST_SQUEEZE=4 leaves room for almost no noted leaves, so that nearly every ray goes through the "list full:
// test, then walk on" path of the deferred walks below)
#ifndef RT_ANY_LIST_SQUEEZE
#define RT_ANY_LIST_SQUEEZE 0
#endif
// Any-hit walk of a shadow ray over pair records in LDS, leaves deferred.
// A shadow ray's tMax never changes and only the answer "is anything hit" is wanted, so the order of the visits is free: a leaf is
// reached by BVHAccel::IntersectP (bvh.cpp:284-320) exactly when its box and the boxes of all its ancestors pass
// Bounds3::IntersectP against that fixed tMax (unless an earlier hit has already ended the walk), and the answer is whether any
// primitive of any such leaf is hit.  This walk makes the same box tests on the same numbers and the same primitive tests, in two
// phases: first the interior nodes alone — every lane of the wave takes interior steps in lockstep until its stack is empty, and
// the leaves it reaches are only noted — then the noted leaves, one primitive per lane per step, until a lane has a hit or its
// list is empty.  A wave waits for its slowest ray either way, so the early exit of an occluded ray in the middle of the walk saved
// nothing, while the alternation between interior steps and leaf tests cost idle lanes on every change of phase.
// The stack is the index half of `st` (RT_FRAME_RING entries: the caller guarantees that many interior levels at most); the list of
// leaves is its tMin half, which a fixed-tMax walk does not need.  A lane whose list is nearly full stops walking; the phases then
// repeat for it.
template <int KSTRIDE, int RING, class Stack>
__device__ __forceinline__ bool trace_any_deferred_from(const DevScene &sc, const float4 *pairs, const float4 *slots, bool has_cur, Stack &st, LaneTrav &L) {
    int *nodes = st.lds_i, *leaves = (int *)st.lds_t;       // [entry * RT_BLOCK], this lane's column
    int sp = 0, nl = 0;
    uint32_t cur = sc.root_info;
    if (has_cur && (cur & RT_PAIR_LEAF)) { leaves[0] = (int)cur; nl = 1; has_cur = false; }     // a tree that is one leaf
    bool found = false;
    int slot = 0, left = 0;                                   // primitives of the leaf under test: next slot, how many are left
    while (true) {
        // ---- interior nodes
        while (has_cur && nl <= RING - 2 - RT_ANY_LIST_SQUEEZE) {
            const float4 *P = pairs + (KSTRIDE == 1 ? 4 : 1) * (size_t)cur;
            float4 q0 = P[0], q1 = P[KSTRIDE], q2 = P[2 * KSTRIDE], q3 = P[3 * KSTRIDE];
            const uint32_t iL = __float_as_uint(q3.x), iR = __float_as_uint(q3.y);
            float tL, tR;
            bool okL, okR;
            if (L.plain) {
                okL = slab_fast(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, L.o, L.invDir, tL);
                okR = slab_fast(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, L.o, L.invDir, tR);
            } else {
                okL = slab_partial(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, L.o, L.invDir, L.negx, L.negy, L.negz, tL);
                okR = slab_partial(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, L.o, L.invDir, L.negx, L.negy, L.negz, tR);
            }
            okL = okL && tL < L.tmax; okR = okR && tR < L.tmax;      // geometry.h:913
            const bool leafL = okL && (iL & RT_PAIR_LEAF) != 0, leafR = okR && (iR & RT_PAIR_LEAF) != 0;
            if (leafL) { leaves[nl * RT_BLOCK] = (int)iL; nl++; }
            if (leafR) { leaves[nl * RT_BLOCK] = (int)iR; nl++; }
            const bool intL = okL && !leafL, intR = okR && !leafR;
            if (intL && intR) { nodes[sp * RT_BLOCK] = (int)iR; sp++; }
            has_cur = intL || intR;
            cur = intL ? iL : iR;
            if (!has_cur && sp > 0) { sp--; cur = (uint32_t)nodes[sp * RT_BLOCK]; has_cur = true; }
        }
        // ---- the leaves noted so far, one primitive per step
        while (!found && (left > 0 || nl > 0)) {
            if (left == 0) {
                nl--;
                LaneTrav E;                                   // lane_enter decodes the info word (first slot, count)
                lane_enter(sc, E, (uint32_t)leaves[nl * RT_BLOCK], true);
                slot = E.cur; left = E.count;
            }
            float4 r0 = slots[3 * (size_t)slot], r1 = slots[3 * (size_t)slot + 1], r2 = slots[3 * (size_t)slot + 2];
            float t, b0, b1, b2;
            if (__float_as_uint(r2.y) & RT_SLOT_SPHERE) found = sphere_test(sc.spheres + 32 * (size_t)__float_as_int(r2.w), L.o, L.d, L.tmax, t);
            else found = triangle_test_sheared(f3(r0.x, r0.y, r0.z), f3(r0.w, r1.x, r1.y), f3(r1.z, r1.w, r2.x), L.o, L.shear, L.tmax, t, b0, b1, b2);
            slot++; left--;
        }
        if (found) has_cur = false;
        if (__ballot(has_cur) == 0) break;
    }
    L.found = found;
    return found;
}
template <int KSTRIDE, int RING, class Stack>
__device__ __forceinline__ bool trace_any_deferred(const DevScene &sc, const float4 *pairs, const float4 *slots, F3 o, F3 d, float tmax, Stack &st, LaneTrav &L) {
    uint32_t dummy = 0;
    bool has_cur = lane_begin<false>(sc, L, st, o, d, tmax, dummy);       // the root's box (the reference's first loop iteration)
    return trace_any_deferred_from<KSTRIDE, RING>(sc, pairs, slots, has_cur, st, L);
}

// (No walk at all was also measured for the 19-record Cornell tree: for a plain ray a leaf is reached exactly when its own box
// passes — nested boxes and monotonic slab arithmetic make every ancestor pass too — so every lane can test all 20 leaf boxes in
// lockstep and keep a bit mask.  It gives the same images and is 9 % slower than this walk: the tree rejects most leaves after six
// or seven steps, and the uniform reads of every record double the LDS traffic.)

// The deferred any-hit walk over quad records in global memory (trees too large for LDS), for kernels that run one ray per lane to
// completion.  Same reasoning as trace_any_deferred: every candidate of a quad step is accepted or rejected for good when the step is
// made (the candidate's `now` of quad_side: all of the reference's conditions on the child and on the grandchild against the
// fixed tMax), accepted leaves are noted, accepted interior nodes are walked in any order.  The node stack keeps indices only
// (ring of RING entries in LDS, older entries spilled to global memory: InfoStack); the leaf list is the other half of the
// kernel's LDS stack area.
template <int RING>
struct InfoStack {
    int *lds;                       // &shared[0][threadIdx.x], stride RT_BLOCK
    int *spill; size_t spill_stride; int cap;
    int sp, lo;                     // entries [lo, sp) are in the ring
    __device__ __forceinline__ void reset() { sp = 0; lo = 0; }
    __device__ __forceinline__ void push(int info) {
        if (sp - lo == RING) {
            if (lo < cap) spill[(size_t)lo * spill_stride] = lds[(lo & (RING - 1)) * RT_BLOCK];
            lo++;
        }
        lds[(sp & (RING - 1)) * RT_BLOCK] = info;
        sp++;
    }
    __device__ __forceinline__ bool pop(int &info) {      // false: the entry was beyond the spill area's capacity (trees that deep are refused at scene creation)
        sp--;
        if (sp < lo) {
            lo = sp;
            if (sp < cap) { info = spill[(size_t)sp * spill_stride]; return true; }
            info = 0; return false;
        }
        info = lds[(sp & (RING - 1)) * RT_BLOCK];
        return true;
    }
};
// Per-lane state of a deferred any-hit walk over quad records, beyond the ray's constants in LaneTrav.
struct AnyWalk {
    uint32_t cur; bool has_cur;     // interior node to visit next
    int nl;                         // leaves noted and not yet tested (info words in the lane's column of `leaves`)
    int slot, left;                 // primitives of the leaf under test: next slot, how many are left
    bool found;
};
// Ray constants and the root (lane_begin's reasoning: an interior root's own test is implied for a plain ray).
template <int RING>
__device__ __forceinline__ void any_walk_begin(const DevScene &sc, LaneTrav &L, AnyWalk &A, InfoStack<RING> &st, int *leaves, F3 o, F3 d, float tmax) {
    lane_setup(L, o, d, tmax);
    st.reset();
    A.has_cur = false; A.nl = 0; A.slot = 0; A.left = 0; A.found = false; A.cur = sc.root_info;
    if (sc.n_nodes != 0) {
        if (L.plain && sc.root_implied && (sc.root_info & RT_PAIR_LEAF) == 0) A.has_cur = true;
        else {
            float tMin;
            A.has_cur = slab_partial(sc.root_box[0], sc.root_box[1], sc.root_box[2], sc.root_box[3], sc.root_box[4], sc.root_box[5], o, L.invDir, L.negx, L.negy, L.negz, tMin) &&
                        tMin < tmax;
        }
    }
    if (A.has_cur && (A.cur & RT_PAIR_LEAF)) { leaves[0] = (int)A.cur; A.nl = 1; A.has_cur = false; }     // a tree that is one leaf
}
// One quad record: the candidates that pass are noted (leaves) or walked / pushed (interior nodes); at most four leaves are added.
template <bool PLAIN, int RING>
__device__ __forceinline__ void any_walk_step(const float4 *quads, const LaneTrav &L, AnyWalk &A, InfoStack<RING> &st, int *leaves) {
    const float4 *Q = quads + 8 * (size_t)A.cur;
    float4 q0 = Q[0], q1 = Q[1], q2 = Q[2], q3 = Q[3], q4 = Q[4], q5 = Q[5], q6 = Q[6];
    uint32_t meta = ((const uint32_t *)Q)[28];
    QuadCand c[4];
    quad_side<PLAIN>(L, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, __float_as_uint(q6.x), __float_as_uint(q6.y), (meta & 64u) != 0,
                     (int)((meta >> 2) & 3u), c[0], c[1]);
    quad_side<PLAIN>(L, q3.x, q3.y, q3.z, q3.w, q4.x, q4.y, q4.z, q4.w, q5.x, q5.y, q5.z, q5.w, __float_as_uint(q6.z), __float_as_uint(q6.w), (meta & 128u) != 0,
                     (int)((meta >> 4) & 3u), c[2], c[3]);
    A.has_cur = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!c[k].now) continue;
        if (c[k].info & RT_PAIR_LEAF) { leaves[A.nl * RT_BLOCK] = (int)c[k].info; A.nl++; }
        else if (!A.has_cur) { A.cur = c[k].info; A.has_cur = true; }
        else st.push((int)c[k].info);
    }
    if (!A.has_cur && st.sp > 0) { int i; A.has_cur = st.pop(i); A.cur = (uint32_t)i; }
}
// One primitive of the noted leaves (the caller checks that there is one: A.left > 0 || A.nl > 0).
__device__ __forceinline__ void any_walk_test(const DevScene &sc, const float4 *slots, const LaneTrav &L, AnyWalk &A, const int *leaves) {
    if (A.left == 0) {
        A.nl--;
        LaneTrav E;                                           // lane_enter decodes the info word (first slot, count)
        lane_enter(sc, E, (uint32_t)leaves[A.nl * RT_BLOCK], true);
        A.slot = E.cur; A.left = E.count;
    }
    float4 r0 = slots[3 * (size_t)A.slot], r1 = slots[3 * (size_t)A.slot + 1], r2 = slots[3 * (size_t)A.slot + 2];
    float t, b0, b1, b2;
    if (__float_as_uint(r2.y) & RT_SLOT_SPHERE) A.found = sphere_test(sc.spheres + 32 * (size_t)__float_as_int(r2.w), L.o, L.d, L.tmax, t);
    else A.found = triangle_test_sheared(f3(r0.x, r0.y, r0.z), f3(r0.w, r1.x, r1.y), f3(r1.z, r1.w, r2.x), L.o, L.shear, L.tmax, t, b0, b1, b2);
    A.slot++; A.left--;
}
// One ray per lane, run to completion: interior nodes until the stack is empty or the list is nearly full, then the noted leaves.
template <int RING>
__device__ __forceinline__ bool trace_any_deferred_quads(const DevScene &sc, const float4 *quads, const float4 *slots, F3 o, F3 d, float tmax, InfoStack<RING> &st,
                                                         int *leaves, LaneTrav &L) {
    AnyWalk A;
    any_walk_begin(sc, L, A, st, leaves, o, d, tmax);
    while (true) {
        while (A.has_cur && A.nl <= RING - 4 - RT_ANY_LIST_SQUEEZE) {
            if (L.plain) any_walk_step<true>(quads, L, A, st, leaves);
            else any_walk_step<false>(quads, L, A, st, leaves);
        }
        while (!A.found && (A.left > 0 || A.nl > 0)) any_walk_test(sc, slots, L, A, leaves);
        if (A.found) A.has_cur = false;
        if (__ballot(A.has_cur) == 0) break;
    }
    L.found = A.found;
    return A.found;
}

// One ray per lane, run to completion (the round-based frame kernels): interior steps until every lane of the wave that is
// still walking has reached a leaf or finished, then the leaf tests together — the same pair records, slab tests and
// per-ray constants as the persistent walk, without its scheduling.
// QUADS: `pairs` holds quad records (KSTRIDE ignored).
template <bool ANY, int KSTRIDE, bool QUADS = false, class Stack>
__device__ __forceinline__ void trace_one(const DevScene &sc, const float4 *pairs, const float4 *slots, F3 o, F3 d, float tmax, Stack &st, LaneTrav &L) {
    uint32_t dummy = 0;
    bool have = lane_begin<false>(sc, L, st, o, d, tmax, dummy);
    lane_enter(sc, L, sc.root_info, have);
    while (L.mode != RT_LANE_DONE) {
        while (L.mode == RT_LANE_WALK) {
            uint32_t info; bool hv;
            if (QUADS) {
                if (L.plain) lane_walk_quad<true, ANY>(pairs, L, st, info, hv);
                else lane_walk_quad<false, ANY>(pairs, L, st, info, hv);
            } else if (L.plain) lane_walk<false, true, KSTRIDE, ANY>(pairs, L, st, dummy, info, hv);
            else lane_walk<false, false, KSTRIDE, ANY>(pairs, L, st, dummy, info, hv);
            lane_enter(sc, L, info, hv);
        }
        if (L.mode == RT_LANE_LEAF) {
            LeafOut r = lane_leaf<ANY, false, ANY>(sc, slots, L, st, dummy);
            L.found = r.found; L.tmax = r.tmax; L.hitSlot = r.hitSlot; L.b0 = r.b0; L.b1 = r.b1; L.b2 = r.b2;
            lane_enter(sc, L, r.info, r.have);
        }
    }
}

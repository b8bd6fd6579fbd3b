// rt_kernels.hip — HIP kernels of the wavefront path tracer and the GPU half of the C ABI (rt_amd.h).
//
// Pipeline for one pass over N = pixels x samples-in-pass camera paths (all state stays in HBM):
//   k_raygen            camera rays, Philox film jitter                       camera.cpp:52-58, stratified.cpp:4-29
//   per bounce b = 0..maxDepth:
//     k_extend          closest-hit BVH traversal + path logic (emission, sky, termination), appends the path to the
//                       index list of its material class                      bvh.cpp:247-282, integrator.cpp:241-257
//                       variants: k_extend (one ray per lane per round), k_extend_carry (small scenes walked out of LDS:
//                       finished lanes take new rays every two rounds), k_extend_pl (persistent lanes, large trees)
//     k_shade<MAT>      one launch per material class present in the scene: BSDF build, next-event estimation, BSDF
//                       sampling, Russian roulette (-> compacted next-bounce queue).  The shadow ray goes to a compacted
//                       shadow queue, or — scenes that fit LDS — is traced inside the kernel: a wave collects its shadow
//                       rays until it holds 64 of them and walks those together.
//                                                                             integrator.cpp:258-280, material.cpp, reflection.cpp
//     k_shadow          any-hit traversal of the shadow queue, adds unoccluded light (k_shadow / k_shadow_pl; not launched when
//                       the shading kernels trace their own shadow rays)       bvh.cpp:284-320, light.cpp:19-21
//                       A shadow ray's tMax is fixed and only "is anything hit" is asked, so the kernels that run one ray per
//                       lane to completion visit the interior nodes first and test the leaves they reach afterwards
//                       (trace_any_deferred, rt_traverse2.h); closest-hit walks keep the reference's order throughout.
//   k_accumulate        per-pixel sum of the pass's path radiances into the frame's sum buffer  integrator.cpp:324-337
// Queues are segmented (one segment per producer block, block-local LDS counters, no global atomics on the data path); no
// scattered float atomics are used: every path owns one float4 slot of radiance that only its own lane updates.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rt_amd.h"
#include "rt_device.h"
#include "rt_traverse2.h"
#include "rt_oct.h"

#define HIP_OK(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) { set_error(ctx_of(this_), std::string(#expr) + ": " + hipGetErrorString(e_)); return -2; } \
    } while (0)

#define RT_NUM_CLASSES 7
#define RT_BATCH_STACK 32       // LDS ring depth of the batch trace kernels (deep trees)
#define RT_SEG_BLOCKS 1280      // most queue segments = largest grid of the wavefront kernels (five blocks per CU)
#define RT_SEG_PER_THREAD (RT_SEG_BLOCKS / RT_BLOCK)
#define RT_FRAME_RING 8         // LDS ring entries per lane (node, tMin) of the frame kernels
#define RT_FLAT_MAX 32          // most leaves of a tree that k_extend_flat tests leaf by leaf
// The number of segments in use is the grid size of the frame kernels (gridDim.x, a multiple of RT_BLOCK): 1280 when every
// kernel of the frame fits five blocks on a CU, 1024 when one of them (the metal / plastic shading kernels on large trees) does not.

// Queues are segmented: block b of a producer kernel appends only to segment b (positions
// [b*segcap, b*segcap + count[b])) using a block-local LDS counter, and publishes count[b] with a plain store when it
// is done.  A consumer kernel turns the RT_SEG_BLOCKS counts into a prefix sum, takes an equal share of the
// concatenated items and walks the segments.  No global atomics, no contended cache line, and the result does not
// depend on scheduling.
struct BounceCounts {
    uint32_t queue[RT_SEG_BLOCKS];                    // paths entering this bounce (written by the previous bounce's shading)
    uint32_t shadow[RT_SEG_BLOCKS];                   // shadow rays produced by this bounce
    uint32_t cls[RT_NUM_CLASSES][RT_SEG_BLOCKS];      // paths per material class
};
struct BounceStats { uint32_t n_queue, n_shadow, n_replay, pad1; };      // n_replay: rays k_extend_flat walked again in the reference's order

struct PathQueue {   // SoA, capacity = cap
    float *ox, *oy, *oz, *dx, *dy, *dz, *tr, *tg, *tb;
    uint32_t *id, *state;   // state: depth | specular << 16
    float *es;              // etaScale of PathIntegrator::Li (integrator.cpp:181,216-219); touched only under RenderParams::li
};
struct HitBuf { int *slot; float *t, *b0, *b1, *b2; };
struct ShadowQueue { uint32_t *id; float *ox, *oy, *oz, *dx, *dy, *dz, *cr, *cg, *cb; };

struct RenderParams {
    float r2c[16], c2w[16];
    int width, height;
    int strata_x, strata_y;
    int spb;            // samples of each pixel in this pass
    int sample0;        // first sample index of this pass
    int max_depth;
    uint32_t k0, k1;
    uint32_t n_paths;
    int rr_shade;       // shading kernels take their chunks round robin (block_chunks)
    FastDiv d_spb, d_width, d_strata_x, d_nstrata;      // divisions by spb, width, strata_x, strata_x * strata_y
    int pack_shadow;    // fused shading kernels: a wave collects its shadow rays and walks them 64 at a time (k_shade)
    int merge_mirror;   // 2: k_shade<RT_MAT_MATTE> shades the mirror list in a second loop of the same launch; 1: mirror paths are binned with the
                        // matte ones and shaded by a branch of the matte code; 0: a launch of their own
    // The estimators next to Li_re (only the kernels built with GEN read these, and the emission step of the extend kernels reads li):
    int li;             // PathIntegrator::Li (integrator.cpp:175-229)
    float rr_threshold; // its rrThreshold
    int nee_mode;       // 0: UniformSampleOneLight and BSDF sampling in one launch; UniformSampleAllLights (integrator.cpp:26-51) takes one launch per
    int nee_light;      // light with nee_mode 1 (direct light of light nee_light only) and a last one with nee_mode 2 (BSDF sampling only)
    uint32_t class_mask; // bit c: the scene has a material of class c (the binning loops skip the others without a ballot)
};

// ---------------------------------------------------------------- helpers --
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t prefix_in_mask(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// Exclusive prefix sum of the gridDim.x segment counts into s_prefix[0..gridDim.x] (last entry = total).
__device__ __forceinline__ uint32_t segment_prefix(const uint32_t *__restrict__ counts, uint32_t *s_prefix, uint32_t *s_wave) {
    uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t per = gridDim.x / RT_BLOCK;     // counts per thread, at most RT_SEG_PER_THREAD
    uint32_t v[RT_SEG_PER_THREAD], sum = 0;
#pragma unroll
    for (int k = 0; k < RT_SEG_PER_THREAD; ++k) { v[k] = (uint32_t)k < per ? counts[t * per + k] : 0u; sum += v[k]; }
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { uint32_t y = __shfl_up(incl, off); if (lane >= (uint32_t)off) incl += y; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; ++w) base += s_wave[w];
    uint32_t run = base + incl - sum;
#pragma unroll
    for (int k = 0; k < RT_SEG_PER_THREAD; ++k) if ((uint32_t)k < per) { s_prefix[t * per + k] = run; run += v[k]; }
    if (t == RT_BLOCK - 1) s_prefix[gridDim.x] = run;
    __syncthreads();
    return s_prefix[gridDim.x];
}
// How consumers divide a queue.  Round robin (block_chunks with rr set, and per wave in k_extend_carry): chunks of consecutive
// positions dealt in turn, so every block sees paths from all over the image and the blocks, which all run at once, finish
// together — with one contiguous share per block the CUs that drew cheap image regions idle at the end.  Contiguous (block_share):
// one share [lo, hi) per block (multiples of the wave size); a block then stays within one image region, which keeps its L1
// working set small when the scene is read from global memory.  The host picks per kernel (rt_render_device).
struct ChunkLoop { uint32_t start, end, stride; };
__device__ __forceinline__ void block_share(uint32_t total, uint32_t &lo, uint32_t &hi);
// Loop bounds of a round-based consumer: round-robin chunks (rr != 0) or the block's contiguous share.
__device__ __forceinline__ ChunkLoop block_chunks(uint32_t total, int rr) {
    ChunkLoop c;
    if (rr) { c.start = blockIdx.x * RT_BLOCK; c.end = total; c.stride = gridDim.x * RT_BLOCK; }
    else { uint32_t lo, hi; block_share(total, lo, hi); c.start = lo; c.end = hi; c.stride = RT_BLOCK; }
    return c;
}
__device__ __forceinline__ void block_share(uint32_t total, uint32_t &lo, uint32_t &hi) {
    uint32_t share = (total + gridDim.x - 1) / gridDim.x;
    share = (share + RT_WAVE - 1) & ~(uint32_t)(RT_WAVE - 1);
    uint64_t l = (uint64_t)blockIdx.x * share;
    lo = (uint32_t)(l < total ? l : total);
    hi = (uint32_t)((l + share) < total ? (l + share) : total);
}
// The waves of the grid take the chunks of RT_WAVE consecutive queue positions round robin (wave w: chunks w, w + W, ...; see
// block_share for why).  WaveChunks maps a wave's k-th chunk to its first queue position.
struct WaveChunks {
    uint32_t n_waves, gwave, n_mine, total;
    __device__ __forceinline__ WaveChunks(uint32_t total_) : total(total_) {
        n_waves = gridDim.x * (RT_BLOCK / RT_WAVE); gwave = blockIdx.x * (RT_BLOCK / RT_WAVE) + threadIdx.x / RT_WAVE;
        const uint32_t n_chunks = (total + RT_WAVE - 1) / RT_WAVE;
        n_mine = gwave < n_chunks ? (n_chunks - gwave + n_waves - 1) / n_waves : 0;
    }
    __device__ __forceinline__ uint32_t first(uint32_t k) const { return (k * n_waves + gwave) * RT_WAVE; }
};

// Segment that holds concatenated index g (s_prefix[s] <= g < s_prefix[s+1]); g < total.
__device__ __forceinline__ uint32_t segment_of(const uint32_t *s_prefix, uint32_t g) {
    uint32_t lo = 0, hi = gridDim.x;   // invariant: s_prefix[lo] <= g < s_prefix[hi]
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (s_prefix[mid] <= g) lo = mid; else hi = mid; }
    return lo;
}
// Append within this block's segment: wave ballot + one LDS atomic per wave.  Returns the position inside the segment.
__device__ __forceinline__ uint32_t block_append(bool want, uint32_t *s_counter) {
    uint64_t mask = __ballot(want);
    if (mask == 0) return 0;
    uint32_t n = __popcll(mask), base = 0;
    int leader = __ffsll((unsigned long long)mask) - 1;
    if ((int)lane_id() == leader) base = atomicAdd(s_counter, n);
    base = __shfl(base, leader);
    return base + prefix_in_mask(mask);
}

// ----------------------------------------------------- batch trace kernels --
template <bool ANY, bool COUNT>
__global__ __launch_bounds__(RT_BLOCK) void k_trace_batch(DevScene sc, size_t n, const float *__restrict__ rays, int aos, int *out_prim, float *out_t,
                                                           float *out_b1, float *out_b2, uint8_t *out_any, int *spill, unsigned long long *work) {
    __shared__ int s_stack[RT_BATCH_STACK * RT_BLOCK];
    size_t nthreads = (size_t)gridDim.x * blockDim.x, gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    TravStackT<RT_BATCH_STACK> st; st.lds = &s_stack[threadIdx.x]; st.spill = spill + gtid; st.spill_stride = nthreads; st.reset();
    uint32_t nNodes = 0, nTests = 0;
    for (size_t i = gtid; i < n; i += nthreads) {
        F3 o, d; float tmax;
        if (aos) { o = f3(rays[7 * i], rays[7 * i + 1], rays[7 * i + 2]); d = f3(rays[7 * i + 3], rays[7 * i + 4], rays[7 * i + 5]); tmax = rays[7 * i + 6]; }
        else { o = f3(rays[i], rays[n + i], rays[2 * n + i]); d = f3(rays[3 * n + i], rays[4 * n + i], rays[5 * n + i]); tmax = rays[6 * n + i]; }
        HitRec h;
        GlobalFetch gf; gf.nodes = sc.nodes; gf.slots = sc.slots;
        bool found = traverse<ANY, COUNT>(sc, gf, o, d, tmax, st, h, nNodes, nTests);
        if (ANY) out_any[i] = found ? 1 : 0;
        else {
            int prim = -1; float b1 = h.b1, b2 = h.b2;
            if (found) {
                float4 r2 = sc.slots[3 * h.slot + 2];
                prim = __float_as_int(r2.z);
                if (__float_as_uint(r2.y) & RT_SLOT_SPHERE) {   // (u, v) of the hit, sphere.cpp:51-58
                    const float *sp = sc.spheres + 32 * (size_t)__float_as_int(r2.w);
                    F3 oo = xf_point(sp + 4, o), dd = xf_vector(sp + 4, d);
                    F3 ph = oo + h.t * dd;
                    float phi = atan2f(ph.y, ph.x);
                    if (phi < 0) phi += 2 * RT_PI;
                    b1 = phi / (2 * RT_PI);
                    b2 = acosf(clampf(ph.z / sp[0], -1.f, 1.f)) / RT_PI;
                }
            }
            out_prim[i] = prim; out_t[i] = h.t;
            if (out_b1) { out_b1[i] = b1; out_b2[i] = b2; }
        }
    }
    if (COUNT) {
        atomicAdd(&work[0], (unsigned long long)nNodes);
        atomicAdd(&work[1], (unsigned long long)nTests);
    }
}

// Batch trace over pair nodes with persistent lanes (rt_traverse2.h).  Each wave owns a contiguous range of rays.
#define RT_LEAF_MIN_LANES 20
#ifndef RT_LEAF_LOAD_AHEAD
#define RT_LEAF_LOAD_AHEAD 1
#endif
#ifndef RT_TRACE_RUN
#define RT_TRACE_RUN 256        // rays a wave of the order-free batch trace draws at a time
#endif
#ifndef RT_TRACE_OCC
#define RT_TRACE_OCC 4          // waves per SIMD the batch trace kernels are compiled for (they are launched four blocks to a CU: five lose to L2 thrash)
#endif
// FREE (closest hit over quad records only): the order-free walk of rt_traverse2.h; a ray whose answer is not safe against the
// reference's visit order is walked again at once, by the lane that holds it, in that order over the quad records (`ord`; replay[0]
// counts such rays).  The wave then runs the ordered step next to the order-free one for as long as one of its lanes needs it — a few
// hundred of a wave's many thousand iterations, for a handful of rays in a million.  (A second launch over a list of the flagged rays
// cost 0.75 - 1.2 ms of a 21 ms launch: a hundred rays, each a chain of some 150 dependent fetches, with the chip otherwise idle.)
// OCT (with FREE): the order-free walk over the eight-wide records with quantised boxes (rt_oct.h) instead of the quad records.
template <bool ANY, bool COUNT, bool QUAD, bool FREE = false, bool OCT = false>
__global__ __launch_bounds__(RT_BLOCK, RT_TRACE_OCC) void k_trace_pairs(DevScene sc, size_t n, const float *__restrict__ rays, int aos, int *out_prim, float *out_t,
                                                           float *out_b1, float *out_b2, uint8_t *out_any, int *spill, unsigned long long *work, int leaf_min,
                                                           uint32_t *replay = nullptr) {
    static_assert(!FREE || (!ANY && !COUNT && QUAD), "the order-free walk is a closest-hit walk over quad records");
    static_assert(!OCT || FREE, "the eight-wide records serve the order-free walk only");
    constexpr int RING = OCT ? RT_OCT_STACK : (QUAD ? RT_QUAD_STACK : RT_PAIR_STACK);
    __shared__ int s_si[RING * RT_BLOCK];
    __shared__ float s_st[RING * RT_BLOCK];
    size_t nthreads = (size_t)gridDim.x * blockDim.x, gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    PairStackT<RING> st; st.lds_i = &s_si[threadIdx.x]; st.lds_t = &s_st[threadIdx.x];
    st.cap = OCT ? max(sc.oct_stack_cap, sc.quad_stack_cap) : (QUAD ? sc.quad_stack_cap : RT_MAX_STACK);      // (OCT: the second, ordered walk is over the quad records)
    st.spill_i = spill + gtid; st.spill_t = (float *)(spill + (size_t)st.cap * nthreads) + gtid; st.spill_stride = nthreads; st.reset();
    size_t nwaves = nthreads / RT_WAVE, gw = gtid / RT_WAVE;
    size_t per = ((n + nwaves - 1) / nwaves + RT_WAVE - 1) & ~(size_t)(RT_WAVE - 1);
    size_t w_lo = gw * per < n ? gw * per : n, w_hi = w_lo + per < n ? w_lo + per : n;
    size_t next = w_lo;
    // FREE: no fixed share; the waves draw runs of RT_TRACE_RUN rays from a counter (replay[2..3], zeroed by the host) until the batch is
    // used up, so that a wave held up at the end of its work — by a ray on its second, ordered walk, a chain of dependent fetches — holds up
    // nothing but itself.  One atomic per run and wave; the answers do not depend on which lane walks which ray.
    bool dry = false;
    if (FREE) { next = 0; w_hi = 0; }
    long long ri = -1;
    LaneTrav L = {};
    L.mode = RT_LANE_DONE; L.hitSlot = -1;
    uint32_t nNodes = 0, nTests = 0, nReplayed = 0;
    const bool exact_slabs = (leaf_min >> 8) & 1;    // knob: every wave takes the general slab path
    leaf_min &= 255;
    bool ord = false;                                // FREE: this lane's ray is on its second walk, in the reference's order
    while (true) {
        bool again = false;
        if (FREE && ri >= 0 && L.mode == RT_LANE_DONE && L.amb && !ord) { ord = true; again = true; nReplayed++; }
        else if (ri >= 0 && L.mode == RT_LANE_DONE) {   // retire
            ord = false;
            size_t i = (size_t)ri;
            if (ANY) out_any[i] = L.found ? 1 : 0;
            else {
                int prim = -1; float b1 = L.b1, b2 = L.b2;
                if (L.found) {
                    float4 r2 = sc.slots[3 * (size_t)L.hitSlot + 2];
                    prim = __float_as_int(r2.z);
                    if (__float_as_uint(r2.y) & RT_SLOT_SPHERE) {   // (u, v) of the hit, sphere.cpp:51-58
                        const float *sp = sc.spheres + 32 * (size_t)__float_as_int(r2.w);
                        F3 oo = xf_point(sp + 4, L.o), dd = xf_vector(sp + 4, L.d);
                        F3 ph = oo + L.tmax * dd;
                        float phi = atan2f(ph.y, ph.x);
                        if (phi < 0) phi += 2 * RT_PI;
                        b1 = phi / (2 * RT_PI);
                        b2 = acosf(clampf(ph.z / sp[0], -1.f, 1.f)) / RT_PI;
                    }
                }
                out_prim[i] = prim; out_t[i] = L.tmax;
                if (out_b1) { out_b1[i] = b1; out_b2[i] = b2; }
            }
            ri = -1;
        }
        // refill: idle lanes take the next rays of the wave's range; the loads are consumed at the end of this iteration
        bool fresh = false;
        F3 po = f3(0, 0, 0), pd = f3(0, 0, 1); float ptmax = 0.f;
        uint64_t im = __ballot(ri < 0);
        if (FREE && im != 0 && next >= w_hi && !dry) {
            unsigned long long base = 0;
            if (lane_id() == 0) base = atomicAdd((unsigned long long *)replay + 1, (unsigned long long)RT_TRACE_RUN);
            base = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            dry = base >= n;
            next = dry ? n : (size_t)base; w_hi = next + RT_TRACE_RUN < n ? next + RT_TRACE_RUN : n;
        }
        if (im != 0 && next < w_hi) {
            size_t cand = next + prefix_in_mask(im);
            if (ri < 0 && cand < w_hi) {
                size_t i = cand;
                ri = (long long)i;
                if (aos) { po = f3(rays[7 * i], rays[7 * i + 1], rays[7 * i + 2]); pd = f3(rays[7 * i + 3], rays[7 * i + 4], rays[7 * i + 5]); ptmax = rays[7 * i + 6]; }
                else { po = f3(rays[i], rays[n + i], rays[2 * n + i]); pd = f3(rays[3 * n + i], rays[4 * n + i], rays[5 * n + i]); ptmax = rays[6 * n + i]; }
                fresh = true;
            }
            next += __popcll(im);
        }
        if (__ballot(ri >= 0) == 0) break;
        // a fresh lane still shows the finished state of its previous ray, so it neither walks nor tests a leaf below
        uint32_t info = 0; bool have = false;
        bool walk = L.mode == RT_LANE_WALK, leaf = L.mode == RT_LANE_LEAF;
        uint64_t lm = __ballot(leaf), wm = __ballot(walk);
        const bool leafrun = leaf && (__popcll(lm) >= leaf_min || wm == 0);
#if RT_LEAF_LOAD_AHEAD
        // the leaf lanes' triangle loads go out before the walkers' step, so that the two fetches of this iteration are in flight together
        float4 pre[3];
        if (leafrun) { const float4 *sl = sc.slots + 3 * (size_t)L.cur; pre[0] = sl[0]; pre[1] = sl[1]; pre[2] = sl[2]; }
#endif
        float entry = 0.f;
        if (FREE) {
            if (walk && !ord) { if (OCT) oct_step_free(sc.octs, L, (uint32_t)L.cur, false, st, info, have, entry); else lane_walk_quad_free(sc.quads, L, st, info, have, entry); }
            if (__ballot(walk && ord) != 0) { if (walk && ord) lane_walk_quad<false>(sc.quads, L, st, info, have); }
        } else if (QUAD) {
            if (__ballot(walk && !L.plain) == 0 && !exact_slabs) { if (walk) lane_walk_quad<true>(sc.quads, L, st, info, have); }     // the whole wave takes one version
            else if (walk) lane_walk_quad<false>(sc.quads, L, st, info, have);
        } else if (walk) lane_walk<COUNT, false>(sc.pairs, L, st, nNodes, info, have);
        if (FREE) {
            if (leafrun && !ord) {
#if RT_LEAF_LOAD_AHEAD
                LeafOutFree r = OCT ? oct_leaf_free(sc, sc.slots, L, st, pre) : lane_leaf_free(sc, sc.slots, L, st, pre);
#else
                LeafOutFree r = OCT ? oct_leaf_free(sc, sc.slots, L, st) : lane_leaf_free(sc, sc.slots, L, st);
#endif
                L.found = r.found; L.amb = r.amb; L.tmax = r.tmax; L.hitSlot = r.hitSlot; L.b0 = r.b0; L.b1 = r.b1; L.b2 = r.b2;
                info = r.info; have = r.have; entry = r.entry;
            }
            if (__ballot(leafrun && ord) != 0) {
                if (leafrun && ord) {
                    LeafOut r = lane_leaf<false, false>(sc, sc.slots, L, st, nTests);
                    L.found = r.found; L.tmax = r.tmax; L.hitSlot = r.hitSlot; L.b0 = r.b0; L.b1 = r.b1; L.b2 = r.b2;
                    info = r.info; have = r.have;
                }
            }
        } else if (leafrun) {
#if RT_LEAF_LOAD_AHEAD
            LeafOut r = lane_leaf<ANY, COUNT>(sc, sc.slots, L, st, nTests, pre);
#else
            LeafOut r = lane_leaf<ANY, COUNT>(sc, sc.slots, L, st, nTests);
#endif
            L.found = r.found; L.tmax = r.tmax; L.hitSlot = r.hitSlot; L.b0 = r.b0; L.b1 = r.b1; L.b2 = r.b2;
            info = r.info; have = r.have;
        }
        if (fresh) {
            if (FREE) have = lane_begin_free(sc, L, st, po, pd, ptmax); else have = lane_begin<COUNT>(sc, L, st, po, pd, ptmax, nNodes);
            info = OCT ? 0u : sc.root_info;      // (the root's record is the first of either kind)
            if (OCT && have && !oct_ray_ok(L)) { L.amb = true; have = false; }      // a reciprocal direction too small or too large for the record-space slab test
        }
        if (FREE && again) {      // the same ray from its start, as BVHAccel::Intersect walks it (the ray's own tMax is in L.t0)
            uint32_t dummy = 0;
            have = lane_begin<false>(sc, L, st, L.o, L.d, L.t0, dummy);
            info = sc.root_info;
        }
        if (walk || leafrun || fresh || again) { lane_enter(sc, L, info, have); if (FREE) L.entry = entry; }
    }
    if (COUNT) {
        atomicAdd(&work[0], (unsigned long long)nNodes);
        atomicAdd(&work[1], (unsigned long long)nTests);
    }
    if (FREE && nReplayed) atomicAdd(&replay[0], nReplayed);      // statistics only
}

// ------------------------------------------------------------------ raygen --
// Film sample = stratum (Kensler permutation of the sample index, keyed per pixel) + Philox jitter;
// ray = PerspectiveCamera::GenerateRay (camera.cpp:52-58).
__device__ __forceinline__ F3 xf_point_proj(const float *m, F3 p) {   // Transform::operator()(Point3f) with the w divide, transform.h:135-146
    float xp = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    float yp = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    float zp = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    float wp = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (wp == 1) return f3(xp, yp, zp);
    float inv = 1.f / wp;
    return f3(inv * xp, inv * yp, inv * zp);
}

__global__ __launch_bounds__(RT_BLOCK) void k_raygen(RenderParams rp, PathQueue q, uint32_t segcap, float4 *lbuf, BounceCounts *counts) {
    uint32_t n = rp.n_paths, lo, hi;
    block_share(n, lo, hi);
    for (uint32_t id = lo + threadIdx.x; id < hi; id += RT_BLOCK) {
        uint32_t pixel = fast_div(id, rp.d_spb), sample = (uint32_t)rp.sample0 + (id - pixel * (uint32_t)rp.spb);
        uint32_t c0 = pixel, c1 = sample, c2 = 0xFFFFFFFFu, c3 = 0;
        philox4x32(c0, c1, c2, c3, rp.k0, rp.k1);
        uint32_t p0 = pixel, p1 = 0xFFFFFFFFu, p2 = 0xFFFFFFFFu, p3 = 0;
        philox4x32(p0, p1, p2, p3, rp.k0, rp.k1);
        uint32_t s = kensler_permute(fast_mod(sample, rp.d_nstrata), rp.d_nstrata, p0);
        int sy = (int)fast_div(s, rp.d_strata_x), sx = (int)(s - (uint32_t)sy * (uint32_t)rp.strata_x);
        float ddx = 1.f / rp.strata_x, ddy = 1.f / rp.strata_y;
        float fx = fminf((sx + u32_to_unit(c0)) * ddx, RT_ONE_MINUS_EPS), fy = fminf((sy + u32_to_unit(c1)) * ddy, RT_ONE_MINUS_EPS);
        int py = (int)fast_div(pixel, rp.d_width), px = (int)(pixel - (uint32_t)py * (uint32_t)rp.width);
        F3 pCamera = xf_point_proj(rp.r2c, f3((float)px + fx, (float)py + fy, 0.f));
        F3 dir = normalize3(pCamera);
        F3 o = xf_point_proj(rp.c2w, f3(0.f, 0.f, 0.f));
        F3 d = xf_vector(rp.c2w, dir);
        uint32_t pos = blockIdx.x * segcap + (id - lo);
        q.ox[pos] = o.x; q.oy[pos] = o.y; q.oz[pos] = o.z; q.dx[pos] = d.x; q.dy[pos] = d.y; q.dz[pos] = d.z;
        q.tr[pos] = 1.f; q.tg[pos] = 1.f; q.tb[pos] = 1.f; q.id[pos] = id; q.state[pos] = 0;
        if (rp.li) q.es[pos] = 1.f;
        lbuf[id] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (threadIdx.x == 0) counts[0].queue[blockIdx.x] = hi - lo;
}

// ------------------------------------------------------------------ extend --
// Copy a small scene (nodes + leaf slots) into LDS; all threads of the block take part.
__device__ __forceinline__ void stage_scene_lds(const DevScene &sc, float4 *s_nodes, float4 *s_slots) {
    for (int i = threadIdx.x; i < 2 * sc.n_nodes; i += RT_BLOCK) s_nodes[i] = sc.nodes[i];
    for (int i = threadIdx.x; i < 3 * sc.n_slots; i += RT_BLOCK) s_slots[i] = sc.slots[i];
}

// Pair records of a small scene into LDS: the k-th float4 of record r goes to s_pairs[k * 64 + r], so that consecutive records
// fall into different banks (lane_walk<.., KSTRIDE = 64>, rt_traverse2.h).
__device__ __forceinline__ void stage_pairs_lds(const DevScene &sc, float4 *s_pairs) {
    for (int i = threadIdx.x; i < 4 * sc.n_pairs; i += RT_BLOCK) s_pairs[(i & 3) * (RT_LDS_SCENE_NODES / 2) + (i >> 2)] = sc.pairs[i];
}

// Shadow ray over the LDS copy of a small scene's pair records (bvh.cpp:284-320; tMax = 1 - ShadowEpsilon, interaction.h:27-32):
// interior nodes first, the leaves they lead to afterwards (trace_any_deferred); RT_ANY_DEFERRED=0 builds the walk in the
// reference's order instead (same answers).
#ifndef RT_ANY_DEFERRED
#define RT_ANY_DEFERRED 1
#endif
template <class Stack>
__device__ __forceinline__ void trace_shadow_lds(const DevScene &sc, const float4 *s_pairs, const float4 *s_slots, F3 o, F3 d, Stack &st, LaneTrav &L) {
#if RT_ANY_DEFERRED
    trace_any_deferred<RT_LDS_SCENE_NODES / 2, RT_FRAME_RING>(sc, s_pairs, s_slots, o, d, 1.f - 0.0001f, st, L);
#else
    trace_one<true, RT_LDS_SCENE_NODES / 2>(sc, s_pairs, s_slots, o, d, 1.f - 0.0001f, st, L);
#endif
}

template <bool LDS_SCENE>
__global__ __launch_bounds__(RT_BLOCK, 5) void k_extend(DevScene sc, RenderParams rp, int bounce, PathQueue q, HitBuf hb, uint32_t *cls_list, uint32_t cap,
                                                      uint32_t segcap, float4 *lbuf, BounceCounts *counts, BounceStats *stats, int *spill, int fast_walk) {
    __shared__ int s_stack[RT_LDS_STACK * RT_BLOCK];
    __shared__ float4 s_nodes[LDS_SCENE ? 2 * RT_LDS_SCENE_NODES : 1], s_slots[LDS_SCENE ? 3 * RT_LDS_SCENE_SLOTS : 1];
    // fast_walk on an LDS scene: the LDS copy holds pair records (k-th float4 of record r at s_nodes[k * 64 + r]) instead of nodes; the int
    // stack area then carries (node, tMin) entries for 8 levels, which the host guarantees to be enough
    if (LDS_SCENE && (fast_walk & 1)) {
        stage_pairs_lds(sc, s_nodes);
        for (int i = threadIdx.x; i < 3 * sc.n_slots; i += RT_BLOCK) s_slots[i] = sc.slots[i];
    } else if (LDS_SCENE) stage_scene_lds(sc, s_nodes, s_slots);
    PairStackT<8, false> pst; pst.lds_i = &s_stack[threadIdx.x]; pst.lds_t = (float *)&s_stack[8 * RT_BLOCK + threadIdx.x]; pst.reset();
    __shared__ uint32_t s_prefix[RT_SEG_BLOCKS + 1], s_wave[RT_BLOCK / RT_WAVE], s_cls[RT_NUM_CLASSES];
    BounceCounts *bc = counts + bounce;
    if (threadIdx.x < RT_NUM_CLASSES) s_cls[threadIdx.x] = 0;
    const uint32_t total = segment_prefix(bc->queue, s_prefix, s_wave);
    if (blockIdx.x == 0 && threadIdx.x == 0) stats[bounce].n_queue = total;
    // the shading kernels of this bounce append to segment blockIdx.x of the shadow queue and of the next queue
    if (threadIdx.x == 0) { bc->shadow[blockIdx.x] = 0; counts[bounce + 1].queue[blockIdx.x] = 0; }
    size_t nthreads = (size_t)gridDim.x * blockDim.x, gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    TravStack st; st.lds = &s_stack[threadIdx.x]; st.spill = spill + gtid; st.spill_stride = nthreads; st.reset();
    PairStackT<8, true> gst; gst.lds_i = pst.lds_i; gst.lds_t = pst.lds_t; gst.cap = sc.quad_stack_cap;
    gst.spill_i = spill + gtid; gst.spill_t = (float *)(spill + (size_t)gst.cap * nthreads) + gtid; gst.spill_stride = nthreads; gst.reset();
    const ChunkLoop cl = block_chunks(total, fast_walk & 2);
    uint32_t seg = cl.start < cl.end ? segment_of(s_prefix, cl.start) : 0;
    const uint32_t seg_base = blockIdx.x * segcap;
    for (uint32_t base = cl.start; base < cl.end; base += cl.stride) {
        uint32_t g = base + threadIdx.x;
        bool active = g < cl.end;
        int cls = -1;
        uint32_t i = 0;
        if (active) {
            while (g >= s_prefix[seg + 1]) seg++;
            i = seg * segcap + (g - s_prefix[seg]);
            F3 o = f3(q.ox[i], q.oy[i], q.oz[i]), d = f3(q.dx[i], q.dy[i], q.dz[i]);
            uint32_t state = q.state[i];
            int depth = state & 0xFFFF; bool spec = (state >> 16) & 1;
            HitRec h; uint32_t c0 = 0, c1 = 0;
            bool found;
            if (LDS_SCENE && (fast_walk & 1)) {
                LaneTrav L = {};
                trace_one<false, RT_LDS_SCENE_NODES / 2>(sc, s_nodes, s_slots, o, d, RT_INF, pst, L);
                found = L.found; h.slot = L.found ? L.hitSlot : -1; h.t = L.tmax; h.b0 = L.b0; h.b1 = L.b1; h.b2 = L.b2;
            } else if (!LDS_SCENE && (fast_walk & 1)) {
                LaneTrav L = {};
                trace_one<false, 1, true>(sc, sc.quads, sc.slots, o, d, RT_INF, gst, L);
                found = L.found; h.slot = L.found ? L.hitSlot : -1; h.t = L.tmax; h.b0 = L.b0; h.b1 = L.b1; h.b2 = L.b2;
            } else if (LDS_SCENE) { LdsFetch lf; lf.nodes = s_nodes; lf.slots = s_slots; found = traverse<false, false>(sc, lf, o, d, RT_INF, st, h, c0, c1); }
            else { GlobalFetch gf; gf.nodes = sc.nodes; gf.slots = sc.slots; found = traverse<false, false>(sc, gf, o, d, RT_INF, st, h, c0, c1); }
            hb.slot[i] = h.slot; hb.t[i] = h.t; hb.b0[i] = h.b0; hb.b1[i] = h.b1; hb.b2[i] = h.b2;
            // integrator.cpp:242-257
            F3 add = f3(0.f, 0.f, 0.f); bool terminal = false;
            if (depth == 0 || spec || (sc.n_lights == 0 && !rp.li)) {      // rp.li: integrator.cpp:185-192, emission does not end the path and there is no sky
                if (found) {
                    int li = sc.slot_light[h.slot];
                    if (li >= 0) {
                        const DevLight &L = sc.lights[li];
                        float4 nf = sc.frames[3 * h.slot];
                        if (L.two_sided || dot3(f3(nf.x, nf.y, nf.z), -d) > 0) add = f3(L.Le[0], L.Le[1], L.Le[2]);   // light.h:73-75
                        if (!black3(add) && !rp.li) terminal = true;
                    }
                } else if (sc.n_lights == 0 && !rp.li) {
                    F3 ud = normalize3(d);
                    double t = 0.5 * (ud.y + 1.0);
                    float a = (float)(1.0 - t), b = (float)t;
                    add = f3(1.0f * a + 0.5f * b, 1.0f * a + 0.7f * b, 1.0f * a + 1.0f * b);
                }
            }
            if (!found || depth >= rp.max_depth) terminal = true;
            if (!black3(add)) {
                uint32_t id = q.id[i];
                float4 L = lbuf[id];
                L.x += q.tr[i] * add.x; L.y += q.tg[i] * add.y; L.z += q.tb[i] * add.z;
                lbuf[id] = L;
            }
            if (!terminal) { cls = sc.materials[sc.slot_material[h.slot]].kind; if (rp.merge_mirror == 1 && cls == RT_MAT_MIRROR) cls = RT_MAT_MATTE; }
        }
        // bin the path into the queue of its material class (one ballot per class present in the wave)
        for (int c = 0; c < RT_NUM_CLASSES; ++c) {
            if (!((rp.class_mask >> c) & 1u)) continue;
            bool mine = (cls == c);
            if (__ballot(mine) == 0) continue;
            uint32_t pos = block_append(mine, &s_cls[c]);
            if (mine) cls_list[(size_t)c * cap + seg_base + pos] = i;
        }
    }
    __syncthreads();
    if (threadIdx.x < RT_NUM_CLASSES) bc->cls[threadIdx.x][blockIdx.x] = s_cls[threadIdx.x];
}

// k_extend for scenes that live in LDS, with carry-over: a lane whose ray is finished after RT_CARRY_ROUNDS rounds of
// (interior steps until every walking lane holds a leaf, then the leaf tests) hands in its result and takes the next ray of
// its wave's range, while lanes with long rays keep theirs.  A wave therefore no longer waits for its slowest ray after every
// 64 rays, and the hand-in / take-over code still runs for most of the wave at once.
#ifndef RT_CARRY_ROUNDS
#define RT_CARRY_ROUNDS 2
#endif
__global__ __launch_bounds__(RT_BLOCK) void k_extend_carry(DevScene sc, RenderParams rp, int bounce, PathQueue q, HitBuf hb, uint32_t *cls_list, uint32_t cap,
                                                            uint32_t segcap, float4 *lbuf, BounceCounts *counts, BounceStats *stats) {
    __shared__ int s_stack[2 * RT_FRAME_RING * RT_BLOCK];
    __shared__ float4 s_nodes[2 * RT_LDS_SCENE_NODES], s_slots[3 * RT_LDS_SCENE_SLOTS];
    stage_pairs_lds(sc, s_nodes);
    for (int i = threadIdx.x; i < 3 * sc.n_slots; i += RT_BLOCK) s_slots[i] = sc.slots[i];
    PairStackT<RT_FRAME_RING, false> pst; pst.lds_i = &s_stack[threadIdx.x]; pst.lds_t = (float *)&s_stack[RT_FRAME_RING * RT_BLOCK + threadIdx.x]; pst.reset();
    __shared__ uint32_t s_prefix[RT_SEG_BLOCKS + 1], s_wave[RT_BLOCK / RT_WAVE], s_cls[RT_NUM_CLASSES];
    BounceCounts *bc = counts + bounce;
    if (threadIdx.x < RT_NUM_CLASSES) s_cls[threadIdx.x] = 0;
    const uint32_t total = segment_prefix(bc->queue, s_prefix, s_wave);
    if (blockIdx.x == 0 && threadIdx.x == 0) stats[bounce].n_queue = total;
    if (threadIdx.x == 0) { bc->shadow[blockIdx.x] = 0; counts[bounce + 1].queue[blockIdx.x] = 0; }
    // Work is dealt out in chunks of 64 queue positions, round robin over all waves of the grid: wave w takes chunks w, w + W,
    // w + 2W, ...  Every wave then sees rays from all over the image, so the waves (and with them the blocks, which all run at
    // once) finish together; contiguous shares left the CUs that drew cheap image regions idle at the end.
    const uint32_t n_waves = gridDim.x * (RT_BLOCK / RT_WAVE), gwave = blockIdx.x * (RT_BLOCK / RT_WAVE) + threadIdx.x / RT_WAVE;
    const uint32_t n_chunks = (total + RT_WAVE - 1) / RT_WAVE;
    const uint32_t w_hi = (gwave < n_chunks ? (n_chunks - gwave + n_waves - 1) / n_waves : 0) * RT_WAVE;   // this wave's positions: logical [0, w_hi)
    uint32_t w_next = 0;
    uint32_t seg = 0;                                                 // per-lane cursor; a lane's positions only grow
    // (a wave with a single chunk has nothing to take over: it runs its rays to the end and hands in once, which keeps the launches
    // of the near-empty late bounces short)
    const int rounds = n_chunks <= n_waves ? 1 << 20 : RT_CARRY_ROUNDS;
    const uint32_t seg_base = blockIdx.x * segcap;
    LaneTrav L = {};
    L.mode = RT_LANE_DONE; L.hitSlot = -1;
    bool busy = false;
    uint32_t cur_i = 0, dummy = 0;
    while (true) {
        // ---- hand in finished rays: hit record, emission / sky / termination (integrator.cpp:242-257), class binning
        const bool fin = busy && L.mode == RT_LANE_DONE;
        int cls = -1;
        if (fin) {
            const uint32_t i = cur_i;
            const bool found = L.found;
            const int slot = found ? L.hitSlot : -1;
            hb.slot[i] = slot; hb.t[i] = L.tmax; hb.b0[i] = L.b0; hb.b1[i] = L.b1; hb.b2[i] = L.b2;
            const uint32_t state = q.state[i];
            const int depth = state & 0xFFFF; const bool spec = (state >> 16) & 1;
            F3 add = f3(0.f, 0.f, 0.f); bool terminal = false;
            if (depth == 0 || spec || (sc.n_lights == 0 && !rp.li)) {      // rp.li: integrator.cpp:185-192, emission does not end the path and there is no sky
                if (found) {
                    int li = sc.slot_light[slot];
                    if (li >= 0) {
                        const DevLight &Lt = sc.lights[li];
                        float4 nf = sc.frames[3 * slot];
                        if (Lt.two_sided || dot3(f3(nf.x, nf.y, nf.z), -L.d) > 0) add = f3(Lt.Le[0], Lt.Le[1], Lt.Le[2]);   // light.h:73-75
                        if (!black3(add) && !rp.li) terminal = true;
                    }
                } else if (sc.n_lights == 0 && !rp.li) {
                    F3 ud = normalize3(L.d);
                    double t = 0.5 * (ud.y + 1.0);
                    float a = (float)(1.0 - t), b = (float)t;
                    add = f3(1.0f * a + 0.5f * b, 1.0f * a + 0.7f * b, 1.0f * a + 1.0f * b);
                }
            }
            if (!found || depth >= rp.max_depth) terminal = true;
            if (!black3(add)) {
                uint32_t id = q.id[i];
                float4 Lr = lbuf[id];
                Lr.x += q.tr[i] * add.x; Lr.y += q.tg[i] * add.y; Lr.z += q.tb[i] * add.z;
                lbuf[id] = Lr;
            }
            if (!terminal) { cls = sc.materials[sc.slot_material[slot]].kind; if (rp.merge_mirror == 1 && cls == RT_MAT_MIRROR) cls = RT_MAT_MATTE; }
            busy = false;
        }
        if (__ballot(fin) != 0)
            for (int c = 0; c < RT_NUM_CLASSES; ++c) {
                if (!((rp.class_mask >> c) & 1u)) continue;
                bool mine = (cls == c);
                if (__ballot(mine) == 0) continue;
                uint32_t pos = block_append(mine, &s_cls[c]);
                if (mine) cls_list[(size_t)c * cap + seg_base + pos] = cur_i;
            }
        // ---- take the next rays of the wave's range
        const uint64_t im = __ballot(!busy);
        const uint32_t left = w_hi - w_next, nidle = (uint32_t)__popcll(im), take = nidle < left ? nidle : left;
        const uint32_t ln = w_next + prefix_in_mask(im);                  // logical position -> queue position
        const uint32_t g = ((ln / RT_WAVE) * n_waves + gwave) * RT_WAVE + (ln & (RT_WAVE - 1));
        if (!busy && prefix_in_mask(im) < take && g < total) {
            while (g >= s_prefix[seg + 1]) seg++;
            cur_i = seg * segcap + (g - s_prefix[seg]);
            F3 o = f3(q.ox[cur_i], q.oy[cur_i], q.oz[cur_i]), d = f3(q.dx[cur_i], q.dy[cur_i], q.dz[cur_i]);
            bool have = lane_begin<false>(sc, L, pst, o, d, RT_INF, dummy);
            lane_enter(sc, L, sc.root_info, have);
            busy = true;
        }
        w_next += take;
        if (__ballot(busy) == 0) break;
        // ---- a few rounds of traversal for every lane that holds a ray
#pragma unroll 1
        for (int k = 0; k < rounds; ++k) {
            while (L.mode == RT_LANE_WALK) {
                uint32_t info; bool hv;
                if (L.plain) lane_walk<false, true, RT_LDS_SCENE_NODES / 2>(s_nodes, L, pst, dummy, info, hv);
                else lane_walk<false, false, RT_LDS_SCENE_NODES / 2>(s_nodes, L, pst, dummy, info, hv);
                lane_enter(sc, L, info, hv);
            }
            if (L.mode == RT_LANE_LEAF) {
                LeafOut r = lane_leaf<false, false>(sc, s_slots, L, pst, dummy);
                L.found = r.found; L.tmax = r.tmax; L.hitSlot = r.hitSlot; L.b0 = r.b0; L.b1 = r.b1; L.b2 = r.b2;
                lane_enter(sc, L, r.info, r.have);
            }
            if (__ballot(L.mode != RT_LANE_DONE) == 0) break;
        }
    }
    __syncthreads();
    if (threadIdx.x < RT_NUM_CLASSES) bc->cls[threadIdx.x][blockIdx.x] = s_cls[threadIdx.x];
}

// ------------------------------------------------------------ extend, flat --
// Closest hit on a tree of at most RT_FLAT_MAX leaves without a walk.  Incoherent rays on a small tree spend their time
// waiting for one another: the lanes of a wave need different numbers of interior steps to reach their leaves and change
// between walking and testing at different moments (lane utilisation 0.42 in k_extend_carry on the Cornell box).  Here every
// lane tests the boxes of ALL leaves, in lockstep, with the boxes coming from scalar loads (the same record for the whole
// wave), keeps the three nearest boxes it pierces, and then tests the primitives of the nearest and of the second nearest leaf.
//
// Why this can give BVHAccel::Intersect's answer (bvh.cpp:247-282) although it ignores that function's visit order:
//   * For a plain ray (slab_fast applies) on a tree whose interior boxes are bit for bit the unions of their children's, a
//     leaf's own box passes the tMax-independent part of Bounds3::IntersectP exactly when every ancestor's does (nested
//     boxes, monotonic slab arithmetic), so the set of leaves the reference can reach at all is the set found here.
//   * A primitive's t and barycentrics do not depend on the ray's tMax; only its acceptance does.  The reference returns the
//     primitive with the smallest t unless two candidates are nearly tied: at equal t the later-tested one wins
//     (triangle.cpp:104-111 rejects only '>'), a hit may raise tMax by an ulp (tScaled <= tMax * det is tested, the rounded
//     tScaled / det is stored), and a box entry distance next to a hit distance can fall on either side of '<'.  All of these
//     need a second candidate — another primitive's t, or the accepted primitive's own leaf-box entry — within a few ulps
//     of the smallest t.
//   * So the lanes work with bands: a leaf is looked at unless its box entry lies beyond the best t so far times
//     (1 + RT_FLAT_CULL_BAND); primitives are tested against that widened bound, the smallest and the second smallest t are kept;
//     and the answer is taken only when the runner-up lies beyond max(best t, entry of the best's leaf) * (1 + RT_FLAT_TIE_BAND).
//     Otherwise — and when a third leaf is still in reach, or the ray is not plain — the ray goes on the block's replay list,
//     which the block walks afterwards in the reference's order (trace_one over the pair records, full waves again).
//   * What the bands assume: no primitive's computed hit distance lies below the entry distance of its own leaf's box (or of an
//     enclosing box) by more than the difference of the two bands, 0.1 % — a leaf skipped here cannot then hold a hit that the
//     reference, arriving there with a larger tMax, would have accepted below the best t found here.
typedef float v2f __attribute__((ext_vector_type(2)));
#ifndef RT_FLAT_CULL_BAND
#define RT_FLAT_CULL_BAND 1e-3f
#endif
#ifndef RT_FLAT_TIE_BAND
#define RT_FLAT_TIE_BAND 1e-5f
#endif
__global__ __launch_bounds__(RT_BLOCK, 5) void k_extend_flat(DevScene sc, RenderParams rp, int bounce, PathQueue q, HitBuf hb, uint32_t *cls_list, uint32_t cap,
                                                           uint32_t segcap, float4 *lbuf, BounceCounts *counts, BounceStats *stats,
                                                           const float4 *__restrict__ flat, int n_flat) {
    __shared__ int s_stack[2 * RT_FRAME_RING * RT_BLOCK];                      // the replay walk's stack
    __shared__ float4 s_nodes[2 * RT_LDS_SCENE_NODES], s_slots[3 * RT_LDS_SCENE_SLOTS];
    __shared__ uint32_t s_info[RT_FLAT_MAX];
    stage_pairs_lds(sc, s_nodes);
    for (int i = threadIdx.x; i < 3 * sc.n_slots; i += RT_BLOCK) s_slots[i] = sc.slots[i];
    if ((int)threadIdx.x < n_flat) s_info[threadIdx.x] = __float_as_uint(flat[2 * threadIdx.x + 1].z);
    PairStackT<RT_FRAME_RING, false> pst; pst.lds_i = &s_stack[threadIdx.x]; pst.lds_t = (float *)&s_stack[RT_FRAME_RING * RT_BLOCK + threadIdx.x]; pst.reset();
    __shared__ uint32_t s_prefix[RT_SEG_BLOCKS + 1], s_wave[RT_BLOCK / RT_WAVE], s_cls[RT_NUM_CLASSES], s_replay;
    BounceCounts *bc = counts + bounce;
    if (threadIdx.x < RT_NUM_CLASSES) s_cls[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_replay = 0;
    const uint32_t total = segment_prefix(bc->queue, s_prefix, s_wave);
    if (blockIdx.x == 0 && threadIdx.x == 0) stats[bounce].n_queue = total;
    if (threadIdx.x == 0) { bc->shadow[blockIdx.x] = 0; counts[bounce + 1].queue[blockIdx.x] = 0; }
    const uint32_t seg_base = blockIdx.x * segcap;
    uint32_t *replay_list = cls_list + (size_t)RT_NUM_CLASSES * cap + seg_base;
    // hit record, emission / sky / termination (integrator.cpp:242-257); returns the material class the path is binned into, or -1
    auto hand_in = [&](uint32_t i, bool found, int slot, float t, float b0, float b1, float b2, F3 d) -> int {
        hb.slot[i] = found ? slot : -1; hb.t[i] = t; hb.b0[i] = b0; hb.b1[i] = b1; hb.b2[i] = b2;
        const uint32_t state = q.state[i];
        const int depth = state & 0xFFFF; const bool spec = (state >> 16) & 1;
        F3 add = f3(0.f, 0.f, 0.f); bool terminal = false;
        if (depth == 0 || spec || (sc.n_lights == 0 && !rp.li)) {      // rp.li: integrator.cpp:185-192, emission does not end the path and there is no sky
            if (found) {
                int li = sc.slot_light[slot];
                if (li >= 0) {
                    const DevLight &Lt = sc.lights[li];
                    float4 nf = sc.frames[3 * slot];
                    if (Lt.two_sided || dot3(f3(nf.x, nf.y, nf.z), -d) > 0) add = f3(Lt.Le[0], Lt.Le[1], Lt.Le[2]);   // light.h:73-75
                    if (!black3(add) && !rp.li) terminal = true;
                }
            } else if (sc.n_lights == 0 && !rp.li) {
                F3 ud = normalize3(d);
                double tt = 0.5 * (ud.y + 1.0);
                float a = (float)(1.0 - tt), b = (float)tt;
                add = f3(1.0f * a + 0.5f * b, 1.0f * a + 0.7f * b, 1.0f * a + 1.0f * b);
            }
        }
        if (!found || depth >= rp.max_depth) terminal = true;
        if (!black3(add)) {
            uint32_t id = q.id[i];
            float4 Lr = lbuf[id];
            Lr.x += q.tr[i] * add.x; Lr.y += q.tg[i] * add.y; Lr.z += q.tb[i] * add.z;
            lbuf[id] = Lr;
        }
        int cls = -1;
        if (!terminal) { cls = sc.materials[sc.slot_material[slot]].kind; if (rp.merge_mirror == 1 && cls == RT_MAT_MIRROR) cls = RT_MAT_MATTE; }
        return cls;
    };
    auto bin = [&](int cls, uint32_t i) {
        for (int c = 0; c < RT_NUM_CLASSES; ++c) {
            if (!((rp.class_mask >> c) & 1u)) continue;
            bool mine = (cls == c);
            if (__ballot(mine) == 0) continue;
            uint32_t pos = block_append(mine, &s_cls[c]);
            if (mine) cls_list[(size_t)c * cap + seg_base + pos] = i;
        }
    };
    const ChunkLoop cl = block_chunks(total, 1);
    uint32_t seg = cl.start < cl.end ? segment_of(s_prefix, cl.start) : 0;
    for (uint32_t base = cl.start; base < cl.end; base += cl.stride) {
        const uint32_t g = base + threadIdx.x;
        const bool active = g < cl.end;
        uint32_t i = 0;
        F3 o = f3(0.f, 0.f, 0.f), d = f3(1.f, 1.f, 1.f);
        if (active) {
            while (g >= s_prefix[seg + 1]) seg++;
            i = seg * segcap + (g - s_prefix[seg]);
            o = f3(q.ox[i], q.oy[i], q.oz[i]); d = f3(q.dx[i], q.dy[i], q.dz[i]);
        }
        LaneTrav L;
        lane_setup(L, o, d, RT_INF);
        const bool run = active && L.plain;
        bool replay = active && !L.plain;
        // ---- every leaf box, the three nearest pierced ones kept as keys: entry distance (low five bits dropped) | leaf number
        // (slab_fast on two-float vectors, which compile to packed subtractions and multiplications — the same IEEE operations on the
        // same numbers: the record's three pairs (min x, min y), (min z, max x), (max y, max z) against the ray's origin and reciprocal
        // direction laid out to match.  max(tMin, smallest positive number) <= tMax is  tMin <= tMax && tMax > 0  in one comparison
        // (denormals are kept), and that maximum, never negative, is also what the key needs.  An entry distance of +Infinity (an
        // overflowing product) makes a key like any other; the comparisons against the reach below then keep the leaf out, as
        // tMin < ray.tMax does in the reference, geometry.h:913.)
        const v2f oA = {L.o.x, L.o.y}, oB = {L.o.z, L.o.x}, oC = {L.o.y, L.o.z};
        const v2f iA = {L.invDir.x, L.invDir.y}, iB = {L.invDir.z, L.invDir.x}, iC = {L.invDir.y, L.invDir.z};
        uint32_t k1 = ~0u, k2 = ~0u, k3 = ~0u;
        for (int k = 0; k < n_flat; ++k) {
            const float4 a = flat[2 * k], b = flat[2 * k + 1];
            const v2f P0 = {a.x, a.y}, P1 = {a.z, a.w}, P2 = {b.x, b.y};
            const v2f T0 = (P0 - oA) * iA, T1 = (P1 - oB) * iB, T2 = (P2 - oC) * iC;      // (x0, y0), (z0, x1), (y1, z1)
            const float widen = 1 + 2 * ((3 * 5.9604644775390625e-08f) / (1 - 3 * 5.9604644775390625e-08f));
            const float nx = fminf(T0.x, T1.y), fx = fmaxf(T0.x, T1.y), ny = fminf(T0.y, T2.x), fy = fmaxf(T0.y, T2.x), nz = fminf(T1.x, T2.y), fz = fmaxf(T1.x, T2.y);
            const float tMin = fmaxf(fmaxf(fmaxf(nx, ny), nz), __uint_as_float(1u)), tMax = fminf(fminf(fx, fy), fz) * widen;
            const uint32_t key = tMin <= tMax ? ((__float_as_uint(tMin) & ~31u) | (uint32_t)k) : ~0u;
            const uint32_t x = max(k1, key); k1 = min(k1, key);
            const uint32_t y = max(k2, x); k2 = min(k2, x);
            k3 = min(k3, y);
        }
        // ---- the primitives of the nearest leaf, then of the second nearest if it is still in reach
        bool found = false;
        float best_t = RT_INF, second = RT_INF, m_up = 0.f, hb0 = 0.f, hb1 = 0.f, hb2 = 0.f;
        int hitSlot = -1;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t key = r == 0 ? k1 : k2;
            const float bound = best_t * (1.f + RT_FLAT_CULL_BAND);
            const bool go = run && key != ~0u && __uint_as_float(key & ~31u) < bound;
            if (__ballot(go) == 0) continue;
            LaneTrav E; E.cur = 0; E.count = 0;
            if (go) lane_enter(sc, E, s_info[key & 31u], true);      // first slot and count of the leaf
            const float entry_up = __uint_as_float(key | 31u);       // not below the box's entry distance
            for (int j = 0; __ballot(go && j < E.count) != 0; ++j) {
                if (!(go && j < E.count)) continue;
                const int slot = E.cur + j;
                const float4 r0 = s_slots[3 * slot], r1 = s_slots[3 * slot + 1], r2 = s_slots[3 * slot + 2];
                const uint32_t flags = __float_as_uint(r2.y);
                float t, b0 = 0.f, b1 = 0.f, b2 = 0.f;
                bool ok;
                if (flags & RT_SLOT_SPHERE) ok = sphere_test(sc.spheres + 32 * (size_t)__float_as_int(r2.w), L.o, L.d, bound, t);
                else ok = triangle_test_sheared(f3(r0.x, r0.y, r0.z), f3(r0.w, r1.x, r1.y), f3(r1.z, r1.w, r2.x), L.o, L.shear, bound, t, b0, b1, b2) &&
                          !(flags & RT_SLOT_REJECT_CLOSEST);
                if (ok) {
                    if (!found || t < best_t) { second = best_t; best_t = t; hitSlot = slot; hb0 = b0; hb1 = b1; hb2 = b2; m_up = entry_up; found = true; }
                    else second = fminf(second, t);
                }
            }
        }
        const float reach = best_t * (1.f + RT_FLAT_CULL_BAND);
        const bool third = run && k3 != ~0u && __uint_as_float(k3 & ~31u) < reach;
        const bool tie = run && found && (second <= fmaxf(best_t, m_up) * (1.f + RT_FLAT_TIE_BAND) || m_up > best_t * (1.f + 0.5f * RT_FLAT_CULL_BAND));
        replay = replay || third || tie;
        {
            const uint32_t pos = block_append(replay, &s_replay);
            if (replay) replay_list[pos] = i;
        }
        int cls = -1;
        if (active && !replay) cls = hand_in(i, found, hitSlot, found ? best_t : RT_INF, hb0, hb1, hb2, d);
        bin(cls, i);
    }
    __threadfence_block();
    __syncthreads();
    // ---- the replay list, in the reference's order
    const uint32_t n_replay = s_replay;
    for (uint32_t base = 0; base < n_replay; base += RT_BLOCK) {
        const uint32_t j = base + threadIdx.x;
        int cls = -1;
        uint32_t i = 0;
        if (j < n_replay) {
            i = replay_list[j];
            const F3 o = f3(q.ox[i], q.oy[i], q.oz[i]), d = f3(q.dx[i], q.dy[i], q.dz[i]);
            LaneTrav L = {};
            trace_one<false, RT_LDS_SCENE_NODES / 2>(sc, s_nodes, s_slots, o, d, RT_INF, pst, L);
            cls = hand_in(i, L.found, L.hitSlot, L.tmax, L.b0, L.b1, L.b2, d);
        }
        bin(cls, i);
    }
    __syncthreads();
    if (threadIdx.x < RT_NUM_CLASSES) bc->cls[threadIdx.x][blockIdx.x] = s_cls[threadIdx.x];
    if (threadIdx.x == 0 && n_replay) atomicAdd(&stats[bounce].n_replay, n_replay);      // statistics only
}

// ------------------------------------------------------------------- shade --
struct SurfacePoint { F3 p, n, ss, ts, wo; float u, v; };

// Interaction at the hit: Triangle::Intersect triangle.cpp:121-146 (frame precomputed per triangle on the host),
// Sphere::Intersect sphere.cpp:51-70 + Transform::operator()(SurfaceInteraction) transform.cpp:205-235.
__device__ __forceinline__ void surface_point(const DevScene &sc, int slot, F3 o, F3 d, float t, float b0, float b1, float b2, bool need_uv, SurfacePoint &sp) {
    float4 r0 = sc.slots[3 * slot], r1 = sc.slots[3 * slot + 1], r2 = sc.slots[3 * slot + 2];
    if (__float_as_uint(r2.y) & RT_SLOT_SPHERE) {
        const float *s = sc.spheres + 32 * (size_t)__float_as_int(r2.w);
        F3 oo = xf_point(s + 4, o), dd = xf_vector(s + 4, d);
        F3 ph = oo + t * dd;
        if (need_uv) {
            float phi = atan2f(ph.y, ph.x);
            if (phi < 0) phi += 2 * RT_PI;
            sp.u = phi / (2 * RT_PI);
            sp.v = acosf(clampf(ph.z / s[0], -1.f, 1.f)) / RT_PI;
        } else { sp.u = sp.v = 0.f; }
        float phiMax = 2 * RT_PI;
        F3 dpdu = f3(-phiMax * ph.y, phiMax * ph.x, 0.f);
        F3 n1 = normalize3(ph);
        sp.p = xf_point(s + 16, ph);
        sp.wo = normalize3(xf_vector(s + 16, -dd));
        sp.n = xf_normal_by_inverse(s + 4, n1);
        sp.ss = normalize3(xf_vector(s + 16, dpdu));
        sp.ts = cross3(sp.n, sp.ss);
    } else {
        F3 p0 = f3(r0.x, r0.y, r0.z), p1 = f3(r0.w, r1.x, r1.y), p2 = f3(r1.z, r1.w, r2.x);
        sp.p = (b0 * p0 + b1 * p1) + b2 * p2;
        float4 fn = sc.frames[3 * slot], fs = sc.frames[3 * slot + 1], ft = sc.frames[3 * slot + 2];
        sp.n = f3(fn.x, fn.y, fn.z); sp.ss = f3(fs.x, fs.y, fs.z); sp.ts = f3(ft.x, ft.y, ft.z);
        sp.wo = -d;
        if (need_uv) {
            const float *uv = sc.slot_uv + 6 * (size_t)slot;
            sp.u = (b0 * uv[0] + b1 * uv[2]) + b2 * uv[4];
            sp.v = (b0 * uv[1] + b1 * uv[3]) + b2 * uv[5];
        } else { sp.u = sp.v = 0.f; }
    }
}
__device__ __forceinline__ F3 to_local(const SurfacePoint &sp, F3 v) { return f3(dot3(v, sp.ss), dot3(v, sp.ts), dot3(v, sp.n)); }
__device__ __forceinline__ F3 to_world(const SurfacePoint &sp, F3 v) {
    return f3(sp.ss.x * v.x + sp.ts.x * v.y + sp.n.x * v.z, sp.ss.y * v.x + sp.ts.y * v.y + sp.n.y * v.z, sp.ss.z * v.x + sp.ts.z * v.y + sp.n.z * v.z);
}

// One scattering lobe of the hit's BSDF (the shipped materials have exactly one, material.cpp:5-82).
struct Lobe {
    int kind;        // 0 none, 1 Lambert, 2 specular reflection, 3 specular transmission, 4 microfacet conductor, 5 metal-roughness,
                     // 6 microfacet reflection with dielectric Fresnel (etaA, etaB) = (1.5, 1) for PlasticMaterial, material.cpp:44-47,
                     // 7 microfacet transmission between etaA and etaB (reflection.cpp:311-355)
    F3 R, eta, k;
    float etaA, etaB, metallic;
    TRDist dist;
};
__device__ __forceinline__ bool lobe_is_specular(int kind) { return kind == 2 || kind == 3; }
__device__ __forceinline__ bool lobe_transmits(int kind) { return kind == 3 || kind == 7; }

// BxDF::f in the local frame: reflection.cpp:135-137, 241-249, 268-284.
__device__ __forceinline__ F3 lobe_f(const Lobe &l, F3 wo, F3 wi) {
    if (l.kind == 1) return f3(l.R.x * RT_INV_PI, l.R.y * RT_INV_PI, l.R.z * RT_INV_PI);
    if (l.kind == 4 || l.kind == 5 || l.kind == 6) {
        float cosO = fabsf(wo.z), cosI = fabsf(wi.z);
        F3 wh = wi + wo;
        if (cosI == 0 || cosO == 0) return f3(0.f, 0.f, 0.f);
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) return f3(0.f, 0.f, 0.f);
        wh = normalize3(wh);
        F3 whf = (wh.z < 0.f) ? -wh : wh;   // Faceforward(wh, (0,0,1))
        float c = dot3(wi, whf);
        float Dv = l.dist.D(wh), Gv = l.dist.G(wo, wi);
        if (l.kind == 4) {
            float ca = fabsf(c);
            F3 F = f3(fr_conductor1(ca, l.eta.x, l.k.x), fr_conductor1(ca, l.eta.y, l.k.y), fr_conductor1(ca, l.eta.z, l.k.z));
            float den = 4 * cosI * cosO;
            return cdiv(f3(((l.R.x * Dv) * Gv) * F.x, ((l.R.y * Dv) * Gv) * F.y, ((l.R.z * Dv) * Gv) * F.z), den);
        }
        if (l.kind == 6) {
            float F = fr_dielectric(c, l.etaA, l.etaB);
            float den = 4 * cosI * cosO;
            return cdiv(f3(((l.R.x * Dv) * Gv) * F, ((l.R.y * Dv) * Gv) * F, ((l.R.z * Dv) * Gv) * F), den);
        }
        float cs = fabsf(c);
        float om = 1 - l.metallic;
        F3 F0 = f3(0.04f * om + l.R.x * l.metallic, 0.04f * om + l.R.y * l.metallic, 0.04f * om + l.R.z * l.metallic);
        double x = 1.0 - (double)cs, x2 = x * x;
        float p5 = (float)(x2 * x2 * x);
        F3 F = f3(F0.x + (1.0f - F0.x) * p5, F0.y + (1.0f - F0.y) * p5, F0.z + (1.0f - F0.z) * p5);
        float dg = Dv * Gv;
        float den = (float)((double)(4 * cosI * cosO) + 0.001);
        F3 spec = cdiv(f3(F.x * dg, F.y * dg, F.z * dg), den);
        float km = (float)(1.0 - (double)l.metallic);
        F3 kD = f3((1.0f - F.x) * km, (1.0f - F.y) * km, (1.0f - F.z) * km);
        return f3(spec.x + (kD.x * l.R.x) * RT_INV_PI, spec.y + (kD.y * l.R.y) * RT_INV_PI, spec.z + (kD.z * l.R.z) * RT_INV_PI);
    }
    if (l.kind == 7) {                                                   // MicrofacetTransmission::f, Radiance mode
        if (same_hemisphere(wo, wi)) return f3(0.f, 0.f, 0.f);
        float cosO = wo.z, cosI = wi.z;
        if (cosI == 0 || cosO == 0) return f3(0.f, 0.f, 0.f);
        float eta = wo.z > 0 ? (l.etaB / l.etaA) : (l.etaA / l.etaB);
        F3 wh = normalize3(wo + eta * wi);
        if (wh.z < 0) wh = -wh;
        float dwo = dot3(wo, wh), dwi = dot3(wi, wh);
        if (dwo * dwi > 0) return f3(0.f, 0.f, 0.f);
        float Fm = 1.f - fr_dielectric(dwo, l.etaA, l.etaB);
        float sq = dwo + eta * dwi;
        float factor = 1 / eta;
        float v = fabsf((((((((l.dist.D(wh) * l.dist.G(wo, wi)) * eta) * eta) * fabsf(dwi)) * fabsf(dwo)) * factor) * factor) / (((cosI * cosO) * sq) * sq));
        return f3((Fm * l.R.x) * v, (Fm * l.R.y) * v, (Fm * l.R.z) * v);
    }
    return f3(0.f, 0.f, 0.f);
}

// BxDF::Pdf for a direction another lobe sampled: reflection.cpp:107-109 (Lambert), 262-266 / 296-300 (microfacet reflection),
// 346-355 (microfacet transmission).
__device__ __forceinline__ float lobe_pdf(const Lobe &l, F3 wo, F3 wi) {
    if (l.kind == 1) return same_hemisphere(wo, wi) ? fabsf(wi.z) * RT_INV_PI : 0.f;
    if (l.kind == 4 || l.kind == 5 || l.kind == 6) {
        if (!same_hemisphere(wo, wi)) return 0.f;
        F3 wh = normalize3(wo + wi);
        return l.dist.Pdf(wo, wh) / (4 * dot3(wo, wh));
    }
    if (l.kind == 7) {
        if (same_hemisphere(wo, wi)) return 0.f;
        float eta = wo.z > 0 ? (l.etaB / l.etaA) : (l.etaA / l.etaB);
        F3 wh = normalize3(wo + eta * wi);
        float dwo = dot3(wo, wh), dwi = dot3(wi, wh);
        if (dwo * dwi > 0) return 0.f;
        float sq = dwo + eta * dwi;
        float dwh_dwi = fabsf(((eta * eta) * dwi) / (sq * sq));
        return l.dist.Pdf(wo, wh) * dwh_dwi;
    }
    return 0.f;
}

// BxDF::Sample_f in the local frame: reflection.cpp:99-105 (Lambert), 62-68 (mirror, FresnelNoOp),
// 75-88 (glass, Radiance mode), 251-260 / 286-294 (microfacet lobes).  pdf == 0 means "no sample".
__device__ __forceinline__ F3 lobe_sample(const Lobe &l, F3 wo, float u0, float u1, F3 &wi, float &pdf) {
    pdf = 0.f;
    if (l.kind == 1) {
        wi = cosine_sample_hemisphere(u0, u1);
        if (wo.z < 0) wi.z *= -1;
        pdf = same_hemisphere(wo, wi) ? fabsf(wi.z) * RT_INV_PI : 0.f;
        return lobe_f(l, wo, wi);
    }
    if (l.kind == 2) {
        wi = f3(-wo.x, -wo.y, wo.z);
        pdf = 1;
        return cdiv_s(f3(1.f * l.R.x, 1.f * l.R.y, 1.f * l.R.z), fabsf(wi.z));
    }
    if (l.kind == 3) {
        bool entering = wo.z > 0;
        float etaI = entering ? 1.f : l.etaB, etaT = entering ? l.etaB : 1.f;
        F3 n = (wo.z < 0.f) ? f3(-0.f, -0.f, -1.f) : f3(0.f, 0.f, 1.f);   // Faceforward(Normal3f(0,0,1), wo)
        float eta = etaI / etaT;
        float cosThetaI = dot3(n, wo);                                   // Refract, reflection.h:52-61
        float sin2ThetaI = fmaxf(0.f, 1 - cosThetaI * cosThetaI);
        float sin2ThetaT = eta * eta * sin2ThetaI;
        if (sin2ThetaT >= 1) return f3(0.f, 0.f, 0.f);
        float cosThetaT = sqrtf(1 - sin2ThetaT);
        wi = eta * (-wo) + (eta * cosThetaI - cosThetaT) * n;
        pdf = 1;
        float fr = fr_dielectric(wi.z, 1.f, l.etaB);
        F3 ft = f3(l.R.x * (1.f - fr), l.R.y * (1.f - fr), l.R.z * (1.f - fr));
        float sc = fdiv(etaI * etaI, etaT * etaT);
        ft = f3(ft.x * sc, ft.y * sc, ft.z * sc);
        return cdiv_s(ft, fabsf(wi.z));
    }
    if (l.kind == 4 || l.kind == 5 || l.kind == 6) {
        if (wo.z == 0) return f3(0.f, 0.f, 0.f);
        F3 wh = l.dist.Sample_wh(wo, u0, u1);
        float dwh = dot3(wo, wh);
        if (dwh < 0) return f3(0.f, 0.f, 0.f);
        wi = -wo + (2 * dwh) * wh;                                       // Reflect, reflection.h:48-50
        if (!same_hemisphere(wo, wi)) return f3(0.f, 0.f, 0.f);
        pdf = l.dist.Pdf(wo, wh) / (4 * dwh);
        return lobe_f(l, wo, wi);
    }
    if (l.kind == 7) {                                                   // MicrofacetTransmission::Sample_f, reflection.cpp:334-344
        if (wo.z == 0) return f3(0.f, 0.f, 0.f);
        F3 wh = l.dist.Sample_wh(wo, u0, u1);
        float cosThetaI = dot3(wh, wo);
        if (cosThetaI < 0) return f3(0.f, 0.f, 0.f);
        float eta = wo.z > 0 ? (l.etaA / l.etaB) : (l.etaB / l.etaA);
        float sin2ThetaI = fmaxf(0.f, 1 - cosThetaI * cosThetaI);        // Refract about wh, reflection.h:52-61
        float sin2ThetaT = eta * eta * sin2ThetaI;
        if (sin2ThetaT >= 1) return f3(0.f, 0.f, 0.f);
        float cosThetaT = sqrtf(1 - sin2ThetaT);
        wi = eta * (-wo) + (eta * cosThetaI - cosThetaT) * wh;
        pdf = lobe_pdf(l, wo, wi);
        return lobe_f(l, wo, wi);
    }
    return f3(0.f, 0.f, 0.f);
}


template <int MAT>
__device__ __forceinline__ bool material_needs_uv(const DevScene &sc, const DevMaterial &m) {
    bool need = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (m.tex[k] >= 0 && sc.textures[m.tex[k]].kind != 0) need = true;
    return need;
}

// Material::ComputeScatteringFunctions, material.cpp:5-82.
template <int MAT>
__device__ __forceinline__ void build_lobe(const DevScene &sc, const DevMaterial &m, float u, float v, Lobe &l) {
    l.kind = 0; l.R = l.eta = l.k = f3(0.f, 0.f, 0.f); l.etaA = l.etaB = 1.f; l.metallic = 0.f; l.dist.ax = l.dist.ay = 1.f;
    if (MAT == RT_MAT_MATTE) { l.kind = 1; l.R = clamp0(eval_texture(sc.textures[m.tex[0]], u, v)); }
    else if (MAT == RT_MAT_MIRROR) {
        F3 R = clamp0(eval_texture(sc.textures[m.tex[0]], u, v));
        if (!black3(R)) { l.kind = 2; l.R = R; }
    } else if (MAT == RT_MAT_GLASS) {
        float eta = eval_texture(sc.textures[m.tex[2]], u, v).x;
        F3 R = clamp0(eval_texture(sc.textures[m.tex[0]], u, v)), T = clamp0(eval_texture(sc.textures[m.tex[1]], u, v));
        if (!black3(R) && !black3(T)) { l.kind = 3; l.R = T; l.etaB = eta; }
    } else if (MAT == RT_MAT_METAL) {
        float ur = eval_texture(sc.textures[m.tex[2]], u, v).x, vr = eval_texture(sc.textures[m.tex[3]], u, v).x;
        if (m.remap) { ur = roughness_to_alpha(ur); vr = roughness_to_alpha(vr); }
        l.kind = 4; l.R = f3(1.f, 1.f, 1.f);
        l.eta = eval_texture(sc.textures[m.tex[0]], u, v); l.k = eval_texture(sc.textures[m.tex[1]], u, v);
        l.dist.ax = fmaxf(0.001f, ur); l.dist.ay = fmaxf(0.001f, vr);
    } else if (MAT == RT_MAT_PLASTIC) {          // first lobe only: Lambert(kd) if kd is not black; the glossy lobe is build_plastic_gloss
        F3 kd = clamp0(eval_texture(sc.textures[m.tex[0]], u, v));
        if (!black3(kd)) { l.kind = 1; l.R = kd; }
    } else if (MAT == RT_MAT_ROUGHGLASS) {       // first lobe only: the reflection, if Kr is not black; the transmission is build_second_lobe
        F3 R = clamp0(eval_texture(sc.textures[m.tex[0]], u, v));
        if (!black3(R)) {
            float rough = eval_texture(sc.textures[m.tex[3]], u, v).x;
            if (m.remap) rough = roughness_to_alpha(rough);
            l.kind = 6; l.R = R; l.etaA = 1.f; l.etaB = eval_texture(sc.textures[m.tex[2]], u, v).x;
            l.dist.ax = fmaxf(0.001f, rough); l.dist.ay = fmaxf(0.001f, rough);
        }
    } else if (MAT == RT_MAT_METALROUGH) {
        F3 c = clamp0(eval_texture(sc.textures[m.tex[0]], u, v));
        if (m.tex[1] < 0) { l.kind = 1; l.R = c; }
        else {
            F3 tmp = clamp0(eval_texture(sc.textures[m.tex[1]], u, v));
            float rough = fmaxf(0.05f, tmp.y);
            l.kind = 5; l.R = c; l.metallic = tmp.z;
            l.dist.ax = fmaxf(0.001f, rough); l.dist.ay = fmaxf(0.001f, rough);
        }
    }
}

// PlasticMaterial's MicrofacetReflection(ks, TrowbridgeReitz(rough, rough), FresnelDielectric(1.5, 1)), material.cpp:41-48.
__device__ __forceinline__ void build_plastic_gloss(const DevScene &sc, const DevMaterial &m, float u, float v, Lobe &l) {
    l.kind = 0; l.R = l.eta = l.k = f3(0.f, 0.f, 0.f); l.etaA = l.etaB = 1.f; l.metallic = 0.f; l.dist.ax = l.dist.ay = 1.f;
    F3 ks = clamp0(eval_texture(sc.textures[m.tex[1]], u, v));
    if (black3(ks)) return;
    float rough = eval_texture(sc.textures[m.tex[2]], u, v).x;
    if (m.remap) rough = roughness_to_alpha(rough);
    l.kind = 6; l.R = ks; l.etaA = 1.5f; l.etaB = 1.f;
    l.dist.ax = fmaxf(0.001f, rough); l.dist.ay = fmaxf(0.001f, rough);
}

// The rough dielectric's MicrofacetTransmission(Kt, TrowbridgeReitz(rough, rough), 1, index, Radiance): the second lobe after the reflection.
__device__ __forceinline__ void build_roughglass_trans(const DevScene &sc, const DevMaterial &m, float u, float v, Lobe &l) {
    l.kind = 0; l.R = l.eta = l.k = f3(0.f, 0.f, 0.f); l.etaA = l.etaB = 1.f; l.metallic = 0.f; l.dist.ax = l.dist.ay = 1.f;
    F3 T = clamp0(eval_texture(sc.textures[m.tex[1]], u, v));
    if (black3(T)) return;
    float rough = eval_texture(sc.textures[m.tex[3]], u, v).x;
    if (m.remap) rough = roughness_to_alpha(rough);
    l.kind = 7; l.R = T; l.etaB = eval_texture(sc.textures[m.tex[2]], u, v).x;
    l.dist.ax = fmaxf(0.001f, rough); l.dist.ay = fmaxf(0.001f, rough);
}

// BSDF::f over the non-specular lobes (reflection.cpp:140-152): the geometric side test picks the reflection or the transmission lobes.
// Only the rough dielectric has a transmission lobe here, so every other material keeps the plain 'if (reflect)' form.
template <int MAT>
__device__ __forceinline__ F3 bsdf_f(const Lobe &lobe, const Lobe &lobe2, const int nlobes, bool reflect, F3 wol, F3 wil) {
    F3 f = f3(0.f, 0.f, 0.f);
    if (MAT != RT_MAT_ROUGHGLASS) {
        if (reflect) {
            f = lobe_f(lobe, wol, wil);
            if (nlobes == 2) f = f + lobe_f(lobe2, wol, wil);
        }
    } else {
        if (reflect != lobe_transmits(lobe.kind)) f = lobe_f(lobe, wol, wil);
        if (nlobes == 2 && reflect != lobe_transmits(lobe2.kind)) f = f + lobe_f(lobe2, wol, wil);
    }
    return f;
}

// Small scenes: every table the shading code follows a slot through (slot -> material -> textures, slot -> vertices /
// frame / uv, lights) is copied to LDS, so the chain of dependent reads per path stays on the CU.
#define RT_LDS_TAB_SLOTS RT_LDS_SCENE_SLOTS
#define RT_LDS_TAB_MATERIALS 16
#define RT_LDS_TAB_TEXTURES 32
#define RT_LDS_TAB_LIGHTS 8
__device__ __forceinline__ void copy_words_to_lds(void *dst, const void *src, int bytes) {
    uint32_t *d = (uint32_t *)dst; const uint32_t *g = (const uint32_t *)src;
    for (int i = threadIdx.x; i < bytes / 4; i += RT_BLOCK) d[i] = g[i];
}

// What shading one path hands on: the shadow ray of next-event estimation with the radiance it carries if unoccluded, and the
// next ray of the path with its throughput and state (depth | specular << 16).
struct ShadeOut { bool has_shadow, has_next; uint32_t nstate; F3 so, sd, sc3, no, nd, nT; float nes; };

// Shading of one path at its hit (integrator.cpp:258-280): BSDF of material class MAT, next-event estimation, BSDF sampling,
// Russian roulette.  `sc` may point its tables at LDS copies.
// ENV: the scene may hold infinite area lights (a compile-time switch, so that the kernels of scenes without one do not carry the code).
// GEN: the estimators next to Li_re, chosen at run time through rp.li / rp.nee_mode (a compile-time switch, so that the kernels of the
// reference's own estimator carry none of it): PathIntegrator::Li's light choice, roulette and etaScale, and the one-light-per-launch
// steps of UniformSampleAllLights.
template <int MAT, bool ENV = false, bool GEN = false>
__device__ __forceinline__ void shade_path(const DevScene &sc, const RenderParams &rp, F3 o, F3 d, F3 T, uint32_t id, uint32_t state, int slot, float hit_t,
                                           float hit_b0, float hit_b1, float hit_b2, ShadeOut &out, float eta_scale = 1.f) {
    out.has_shadow = false; out.has_next = false; out.nstate = 0; out.nes = eta_scale;
    out.so = out.sd = out.sc3 = out.no = out.nd = out.nT = f3(0.f, 0.f, 0.f);
    int depth = state & 0xFFFF;
    const DevMaterial &m = sc.materials[sc.slot_material[slot]];
    SurfacePoint sp;
    surface_point(sc, slot, o, d, hit_t, hit_b0, hit_b1, hit_b2, material_needs_uv<MAT>(sc, m), sp);
    uint32_t pixel = fast_div(id, rp.d_spb), sample = (uint32_t)rp.sample0 + (id - pixel * (uint32_t)rp.spb);
    uint32_t a0 = pixel, a1 = sample, a2 = (uint32_t)depth, a3 = 0;   // {light pick, uLight.x, uLight.y, roulette}
    philox4x32(a0, a1, a2, a3, rp.k0, rp.k1);
    // ---- BSDF sampling, integrator.cpp:265-279 + BSDF::Sample_f reflection.cpp:173-217 (as a lambda: the matte kernel runs it
    // a second time, specialised for a mirror lobe, when mirror paths ride in its list)
    auto sample_bsdf = [&](const Lobe &lobe, const Lobe &lobe2, const int nlobes) {
        uint32_t b0 = pixel, b1 = sample, b2 = (uint32_t)depth, b3 = 1;   // {bsdf.x, bsdf.y, -, -}
        philox4x32(b0, b1, b2, b3, rp.k0, rp.k1);
        // BSDF::Sample_f: component min(floor(u0 * n), n - 1), u0 remapped (reflection.cpp:176-190)
        float ub = u32_to_unit(b0), u1 = u32_to_unit(b1);
        int comp = nlobes == 1 ? 0 : min((int)floorf(ub * nlobes), nlobes - 1);
        float u0 = fminf(ub * nlobes - comp, RT_ONE_MINUS_EPS);
        Lobe chosen = lobe, other = lobe2;
        if (nlobes == 2 && comp == 1) { chosen = lobe2; other = lobe; }
        F3 woW = -d;
        F3 wol = to_local(sp, woW), wil; float pdf = 0.f;
        F3 f = f3(0.f, 0.f, 0.f);
        if (wol.z != 0) {
            f = lobe_sample(chosen, wol, u0, u1, wil, pdf);
            if (pdf == 0) f = f3(0.f, 0.f, 0.f);
            else {
                F3 wiW = to_world(sp, wil);
                if (nlobes == 2) { pdf += lobe_pdf(other, wol, wil); pdf /= nlobes; }      // reflection.cpp:199-204
                if (!lobe_is_specular(chosen.kind)) {
                    bool reflect = dot3(wiW, sp.n) * dot3(woW, sp.n) > 0;
                    f = bsdf_f<MAT>(lobe, lobe2, nlobes, reflect, wol, wil);
                }
                if (!black3(f) && pdf != 0.f) {
                    float ad = fabsf(dot3(wiW, sp.n));
                    F3 beta = cdiv_s(f3(f.x * ad, f.y * ad, f.z * ad), pdf);
                    bool alive = true;
                    F3 nT;
                    if (GEN && rp.li) {            // integrator.cpp:214-226: roulette on the whole path's throughput times etaScale
                        nT = T * beta;
                        float es = eta_scale;
                        if (chosen.kind == 3) {    // specular transmission: BSDF::eta is the glass's index
                            float eta = chosen.etaB;
                            es *= (dot3(woW, sp.n) > 0) ? (eta * eta) : 1 / (eta * eta);
                        }
                        out.nes = es;
                        float mx = maxcomp3(f3(nT.x * es, nT.y * es, nT.z * es));
                        if (mx < rp.rr_threshold && depth > 3) {
                            float qq = fmaxf(.05f, 1 - mx);
                            if (u32_to_unit(a3) < qq) alive = false;
                            nT = cdiv_s(nT, 1 - qq);
                        }
                    } else {
                        if (depth > 3) {
                            float qq = fmaxf(.05f, 1 - maxcomp3(beta));
                            if (u32_to_unit(a3) < qq) alive = false;
                            beta = cdiv_s(beta, 1 - qq);
                        }
                        nT = T * beta;
                    }
                    if (alive) {
                        out.has_next = true;
                        out.no = offset_ray_origin(sp.p, sp.n, wiW); out.nd = wiW; out.nT = nT;
                        out.nstate = (uint32_t)(depth + 1) | (lobe_is_specular(chosen.kind) ? (1u << 16) : 0u);
                    }
                }
            }
        }
    };
    // Mirror paths that k_extend binned with the matte ones (RenderParams::merge_mirror): MirrorMaterial has one specular lobe,
    // out.so there is out.no next-event estimation and the whole shading is one reflection (material.cpp:12-20, reflection.cpp:62-68).
    // Their own launch read its short index list with sparse gathers and cost 4 % of a Cornell frame.
    if (MAT == RT_MAT_MATTE && rp.merge_mirror == 1 && m.kind == RT_MAT_MIRROR) {
        Lobe lm, none; none.kind = 0;
        build_lobe<RT_MAT_MIRROR>(sc, m, sp.u, sp.v, lm);
        if (lm.kind != 0 && (!GEN || rp.nee_mode != 1)) sample_bsdf(lm, none, 1);
    } else {
        Lobe lobe, lobe2;       // lobe2: PlasticMaterial (BSDF::Add order: Lambert, then microfacet) and the rough dielectric (reflection, then transmission)
        build_lobe<MAT>(sc, m, sp.u, sp.v, lobe);
        lobe2.kind = 0;
        if (MAT == RT_MAT_PLASTIC) {
            build_plastic_gloss(sc, m, sp.u, sp.v, lobe2);
            if (lobe.kind == 0) { lobe = lobe2; lobe2.kind = 0; }
        }
        if (MAT == RT_MAT_ROUGHGLASS) {
            build_roughglass_trans(sc, m, sp.u, sp.v, lobe2);
            if (lobe.kind == 0) { lobe = lobe2; lobe2.kind = 0; }
        }
        const int nlobes = ((MAT == RT_MAT_PLASTIC || MAT == RT_MAT_ROUGHGLASS) && lobe2.kind != 0) ? 2 : 1;
        // ---- next-event estimation: UniformSampleOneLight + EstimateDirect, integrator.cpp:53-102,259-264
        const bool do_nee = !GEN || rp.nee_mode != 2, do_bsdf = !GEN || rp.nee_mode != 1;
        if (MAT != RT_MAT_MIRROR && do_nee && lobe.kind != 0 && !lobe_is_specular(lobe.kind) && sc.n_lights > 0) {
            int nL = sc.n_lights;
            int ln = min((int)(u32_to_unit(a0) * nL), nL - 1);
            float selPdf = frcp((float)nL);
            float u0 = u32_to_unit(a1), u1 = u32_to_unit(a2);
            if (GEN && rp.nee_mode == 1) {         // UniformSampleAllLights, this launch's light: EstimateDirect as it is (dividing by 1 changes no bit)
                ln = rp.nee_light; selPdf = 1.f;
                uint32_t c0 = pixel, c1 = sample, c2 = (uint32_t)depth, c3 = 2u + (uint32_t)ln;     // {uLight.x, uLight.y, -, -} of light ln
                philox4x32(c0, c1, c2, c3, rp.k0, rp.k1);
                u0 = u32_to_unit(c0); u1 = u32_to_unit(c1);
            } else if (GEN && rp.li) {             // Distribution1D::SampleDiscrete over the uniform light distribution (sampling.h:40-48, pbrt.h:213-226)
                float u = u32_to_unit(a0);
                int first = 0, len = nL + 1;
                while (len > 0) {
                    int half = len >> 1, middle = first + half;
                    if (sc.light_cdf[middle] <= u) { first = middle + 1; len -= half + 1; }
                    else len = half;
                }
                ln = min(max(first - 1, 0), nL - 1);
                selPdf = 1.f / (sc.light_func_int * nL);
            }
            const DevLight &L = sc.lights[ln];
            if (ENV && L.kind == RT_LIGHT_INFINITE) {
                // EstimateDirect (integrator.cpp:74-102) around InfiniteAreaLight::Sample_Li (infinite.cpp:83-106); the visibility
                // segment ends at p + wi * 2 * worldRadius, a point without a normal, so that end gets no offset (interaction.h:27-36)
                const DevEnv &E = sc.envs[L.env];
                F3 wi; float pdf;
                F3 Li = env_sample_li(E, u0, u1, wi, pdf);
                if (pdf > 0 && !black3(Li)) {
                    F3 wol = to_local(sp, sp.wo), wil = to_local(sp, wi);
                    F3 f = f3(0.f, 0.f, 0.f);
                    if (wol.z != 0) {
                        bool reflect = dot3(wi, sp.n) * dot3(sp.wo, sp.n) > 0;
                        f = bsdf_f<MAT>(lobe, lobe2, nlobes, reflect, wol, wil);
                    }
                    float ad = fabsf(dot3(wi, sp.n));
                    f = f3(f.x * ad, f.y * ad, f.z * ad);
                    if (!black3(f)) {
                        F3 far_end = sp.p + (2 * E.world_radius) * wi;
                        F3 origin = offset_ray_origin(sp.p, sp.n, far_end - sp.p);
                        F3 target = offset_ray_origin(far_end, f3(0.f, 0.f, 0.f), origin - far_end);
                        F3 Ld = cdiv_s(cdiv_s(f * Li, pdf), selPdf);
                        out.has_shadow = true; out.so = origin; out.sd = target - origin; out.sc3 = T * Ld;
                    }
                }
            } else {
            F3 p0 = f3(L.p0[0], L.p0[1], L.p0[2]), p1 = f3(L.p1[0], L.p1[1], L.p1[2]), p2 = f3(L.p2[0], L.p2[1], L.p2[2]);
            float su0 = fsqrt(u0);
            float lb0 = 1 - su0, lb1 = u1 * su0;                       // UniformSampleTriangle, sampling.cpp:108-111
            F3 lp = (lb0 * p0 + lb1 * p1) + (1 - lb0 - lb1) * p2;      // Triangle::Sample, triangle.cpp:229-248
            F3 lnrm = f3(L.n[0], L.n[1], L.n[2]);
            if (L.has_n) {
                float w2 = 1 - lb0 - lb1;
                F3 ns = (lb0 * f3(L.vn[0], L.vn[1], L.vn[2]) + lb1 * f3(L.vn[3], L.vn[4], L.vn[5])) + w2 * f3(L.vn[6], L.vn[7], L.vn[8]);
                if (dot3(lnrm, ns) < 0.f) lnrm = -lnrm;
            }
            float pdf = frcp(L.area);
            F3 wi = lp - sp.p;                                          // Shape::Sample(ref,u,pdf), shape.cpp:20-31
            if (lensq(wi) == 0) pdf = 0;
            else {
                wi = normalize3_s(wi);
                pdf *= fdiv(lensq(sp.p - lp), fabsf(dot3(lnrm, -wi)));
                if (isinf(pdf)) pdf = 0.f;
            }
            F3 Li = f3(0.f, 0.f, 0.f);                                  // DiffuseAreaLight::Sample_Li, light.cpp:43-54
            if (pdf == 0 || lensq(lp - sp.p) == 0) pdf = 0;
            else {
                wi = normalize3_s(lp - sp.p);
                if (L.two_sided || dot3(lnrm, -wi) > 0) Li = f3(L.Le[0], L.Le[1], L.Le[2]);
            }
            if (pdf > 0 && !black3(Li)) {
                // BSDF::f(wo, wi, ALL & ~SPECULAR) * AbsDot(wi, ns), reflection.cpp:140-152
                F3 wol = to_local(sp, sp.wo), wil = to_local(sp, wi);
                F3 f = f3(0.f, 0.f, 0.f);
                if (wol.z != 0) {
                    bool reflect = dot3(wi, sp.n) * dot3(sp.wo, sp.n) > 0;
                    f = bsdf_f<MAT>(lobe, lobe2, nlobes, reflect, wol, wil);
                }
                float ad = fabsf(dot3(wi, sp.n));
                f = f3(f.x * ad, f.y * ad, f.z * ad);
                if (!black3(f)) {
                    F3 origin = offset_ray_origin(sp.p, sp.n, lp - sp.p);     // Interaction::SpawnRayTo, interaction.h:27-32
                    F3 target = offset_ray_origin(lp, lnrm, origin - lp);
                    F3 Ld = cdiv_s(cdiv_s(f * Li, pdf), selPdf);
                    out.has_shadow = true; out.so = origin; out.sd = target - origin; out.sc3 = T * Ld;
                }
            }
            }
        }
        if (do_bsdf && lobe.kind != 0) sample_bsdf(lobe, lobe2, nlobes);
    }
}

// TABLES: 0 = every table read from global memory, 1 = all of them in LDS (scenes of at most 64 slots), 2 = materials, textures and
// lights in LDS, the per-slot tables in global memory.
// FUSE (TABLES == 1 scenes whose tree also fits the LDS walk): the shadow ray of next-event estimation is traced here, out of an
// LDS copy of the pair records, instead of going through the shadow queue to k_shadow; the unoccluded light is added to
// the path's radiance slot at once.  The tables live in dynamic LDS sized to the scene (shade_lds_bytes).
// (matte and mirror are held at five waves per SIMD, which is also what their LDS allows; the heavier materials take the registers they need)
template <int MAT, int TABLES, bool FUSE, bool ENV = false, bool GEN = false>
__global__ __launch_bounds__(RT_BLOCK, (MAT == RT_MAT_MATTE || MAT == RT_MAT_MIRROR || MAT == RT_MAT_METALROUGH) ? 5 : 4) void k_shade(DevScene sc_in, RenderParams rp, int bounce, PathQueue q, PathQueue qn, HitBuf hb, const uint32_t *cls_list,
                                                     uint32_t cap, uint32_t segcap, ShadowQueue sq, BounceCounts *counts, float4 *lbuf, BounceStats *stats) {
    constexpr bool LDS_TABLES = TABLES == 1;
    static_assert(!FUSE || LDS_TABLES, "the fused shadow walk needs the LDS tables");
    __shared__ uint32_t s_prefix[RT_SEG_BLOCKS + 1], s_wave[RT_BLOCK / RT_WAVE], s_out[2];
    extern __shared__ float4 s_dyn[];
    __shared__ float4 s_pairs[FUSE ? 2 * RT_LDS_SCENE_NODES : 1];
    __shared__ int s_stack[FUSE ? 2 * RT_FRAME_RING * RT_BLOCK : 1];
    DevScene sc = sc_in;
    const float4 *s_slots = s_dyn;
    if (LDS_TABLES) {      // the barriers of segment_prefix below order these copies before their use
        const int ns = sc_in.n_slots;
        float4 *t_slots = s_dyn, *t_frames = t_slots + 3 * ns;
        DevTexture *t_tex = (DevTexture *)(t_frames + 3 * ns);
        DevLight *t_lights = (DevLight *)(t_tex + sc_in.n_textures);
        DevMaterial *t_mats = (DevMaterial *)(t_lights + sc_in.n_lights);
        float *t_uv = (float *)(t_mats + sc_in.n_materials);
        int *t_slot_mat = (int *)(t_uv + 6 * ns);
        copy_words_to_lds(t_slots, sc_in.slots, 48 * ns); copy_words_to_lds(t_frames, sc_in.frames, 48 * ns);
        copy_words_to_lds(t_uv, sc_in.slot_uv, 24 * ns); copy_words_to_lds(t_slot_mat, sc_in.slot_material, 4 * ns);
        copy_words_to_lds(t_mats, sc_in.materials, (int)sizeof(DevMaterial) * sc_in.n_materials);
        copy_words_to_lds(t_tex, sc_in.textures, (int)sizeof(DevTexture) * sc_in.n_textures);
        copy_words_to_lds(t_lights, sc_in.lights, (int)sizeof(DevLight) * sc_in.n_lights);
        sc.slots = t_slots; sc.frames = t_frames; sc.slot_uv = t_uv; sc.slot_material = t_slot_mat; sc.materials = t_mats; sc.textures = t_tex;
        sc.lights = t_lights;
    }
    if (TABLES == 2) {     // larger scenes: only the tables that do not grow with the geometry (materials, textures, lights)
        DevTexture *t_tex = (DevTexture *)s_dyn;
        DevLight *t_lights = (DevLight *)(t_tex + sc_in.n_textures);
        DevMaterial *t_mats = (DevMaterial *)(t_lights + sc_in.n_lights);
        copy_words_to_lds(t_mats, sc_in.materials, (int)sizeof(DevMaterial) * sc_in.n_materials);
        copy_words_to_lds(t_tex, sc_in.textures, (int)sizeof(DevTexture) * sc_in.n_textures);
        copy_words_to_lds(t_lights, sc_in.lights, (int)sizeof(DevLight) * sc_in.n_lights);
        sc.materials = t_mats; sc.textures = t_tex; sc.lights = t_lights;
    }
    if (FUSE)              // pair records, k-th float4 of record r at s_pairs[k * 64 + r] (the layout k_extend / k_shadow walk)
        stage_pairs_lds(sc_in, s_pairs);
    PairStackT<RT_FRAME_RING, false> pst; pst.lds_i = &s_stack[FUSE ? threadIdx.x : 0]; pst.lds_t = (float *)&s_stack[FUSE ? RT_FRAME_RING * RT_BLOCK + threadIdx.x : 0]; pst.reset();
    uint32_t n_shadow_rays = 0;
    BounceCounts *bc = counts + bounce;
    // continue this block's segments where the previous class kernel of this bounce left them
    if (threadIdx.x == 0) { s_out[0] = bc->shadow[blockIdx.x]; s_out[1] = counts[bounce + 1].queue[blockIdx.x]; }
    const uint32_t seg_base = blockIdx.x * segcap;
    // The matte kernel also shades the mirror list (merge_mirror == 2), in a second loop: one specular lobe, no next-event estimation
    // (material.cpp:12-20, reflection.cpp:62-68).  A launch of its own over that short list cost 4 % of a Cornell frame, and one loop
    // over both classes makes nearly every wave run both branches with most lanes idle in one of them.
    // (two instances of the loop, so that each gets its own register allocation)
    auto shade_list = [&](auto mat_tag) {
    constexpr int M = decltype(mat_tag)::value;
    const uint32_t *list = cls_list + (size_t)M * cap;
    const uint32_t total = segment_prefix(bc->cls[M], s_prefix, s_wave);
    const ChunkLoop cl = block_chunks(total, rp.rr_shade);
    uint32_t seg = cl.start < cl.end ? segment_of(s_prefix, cl.start) : 0;
    // Shadow-ray packing (FUSE, rp.pack_shadow): about six paths in ten produce a shadow ray, so a wave that walks the rays of its own 64
    // paths at once starts every walk with four lanes in ten idle.  Instead the wave parks its shadow rays (origin, direction, radiance,
    // path id: ten words) in its own columns of the stack area, which is idle between walks, and walks them when 64 have come
    // together; the rays that do not fit stay in their lanes' registers during the walk and are parked after it.  No barrier is
    // involved: only the wave itself touches its columns.  One more trip through the loop, with no path in it, walks what is left.
    // An any-hit answer does not depend on the lane that computes it and a path has one shadow ray per bounce, so the image is the same
    // to the last bit.
    constexpr bool WALKS = FUSE && M != RT_MAT_MIRROR;
    const bool pack = WALKS && rp.pack_shadow != 0;
    const uint32_t wave_col = threadIdx.x & ~(uint32_t)(RT_WAVE - 1), lane = threadIdx.x & (RT_WAVE - 1);
    uint32_t n_parked = 0;
    for (uint32_t base = cl.start; ; base += cl.stride) {
        const bool flush = base >= cl.end;
        if (flush && (!pack || n_parked == 0)) break;
        uint32_t j = base + threadIdx.x;
        bool active = j < cl.end;
        bool has_shadow = false, has_next = false;
        uint32_t id = 0, nstate = 0;
        float nes = 1.f;
        F3 so = f3(0, 0, 0), sd = f3(0, 0, 0), sc3 = f3(0, 0, 0), no = f3(0, 0, 0), nd = f3(0, 0, 0), nT = f3(0, 0, 0);
        if (active) {
            while (j >= s_prefix[seg + 1]) seg++;
            uint32_t i = list[seg * segcap + (j - s_prefix[seg])];
            F3 o = f3(q.ox[i], q.oy[i], q.oz[i]), d = f3(q.dx[i], q.dy[i], q.dz[i]);
            F3 T = f3(q.tr[i], q.tg[i], q.tb[i]);
            id = q.id[i];
            uint32_t state = q.state[i];
            ShadeOut out;
            if (GEN) {
                shade_path<M, ENV, true>(sc, rp, o, d, T, id, state, hb.slot[i], hb.t[i], hb.b0[i], hb.b1[i], hb.b2[i], out, rp.li ? q.es[i] : 1.f);
                nes = out.nes;
            } else
                shade_path<M, ENV>(sc, rp, o, d, T, id, state, hb.slot[i], hb.t[i], hb.b0[i], hb.b1[i], hb.b2[i], out);
            has_shadow = out.has_shadow; has_next = out.has_next; nstate = out.nstate;
            so = out.so; sd = out.sd; sc3 = out.sc3; no = out.no; nd = out.nd; nT = out.nT;
        }
        // the next ray goes out first, so that its registers are free during the shadow walk
        if (!flush) {
            uint32_t npos = seg_base + block_append(has_next, &s_out[1]);
            if (has_next) {
                qn.ox[npos] = no.x; qn.oy[npos] = no.y; qn.oz[npos] = no.z; qn.dx[npos] = nd.x; qn.dy[npos] = nd.y; qn.dz[npos] = nd.z;
                qn.tr[npos] = nT.x; qn.tg[npos] = nT.y; qn.tb[npos] = nT.z; qn.id[npos] = id; qn.state[npos] = nstate;
                if (GEN && rp.li) qn.es[npos] = nes;
            }
        }
        if (WALKS) {            // bvh.cpp:284-320 on the spot; tMax = 1 - ShadowEpsilon (interaction.h:27-32)
            // (Packing the BLOCK's shadow rays into full waves — three barriers per chunk, walkers rotated over the waves — was measured
            // 4 % slower than no packing, because the waves of a block then move in lockstep.)
            const uint64_t hm = __ballot(has_shadow);
            const uint32_t nn = (uint32_t)__popcll(hm);
            n_shadow_rays += nn;
            bool walker = has_shadow;                    // this lane holds a ray to walk now (so, sd, sc3, id)
            uint32_t rank = 0, room = 0;
            bool walk_now = true;
            if (pack) {
                rank = prefix_in_mask(hm); room = RT_WAVE - n_parked;
                walk_now = n_parked + nn >= RT_WAVE || flush;
                int *col = &s_stack[wave_col];
                if (has_shadow && rank < room) {         // park (column n_parked + rank)
                    int *p = col + n_parked + rank;
                    p[0] = __float_as_int(so.x); p[RT_BLOCK] = __float_as_int(so.y); p[2 * RT_BLOCK] = __float_as_int(so.z);
                    p[3 * RT_BLOCK] = __float_as_int(sd.x); p[4 * RT_BLOCK] = __float_as_int(sd.y); p[5 * RT_BLOCK] = __float_as_int(sd.z);
                    p[6 * RT_BLOCK] = __float_as_int(sc3.x); p[7 * RT_BLOCK] = __float_as_int(sc3.y); p[8 * RT_BLOCK] = __float_as_int(sc3.z);
                    p[9 * RT_BLOCK] = (int)id;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            if (walk_now) {
                F3 wo_ = so, wd_ = sd, wc_ = sc3; uint32_t wid = id;
                if (pack) {
                    const uint32_t have = n_parked + nn < RT_WAVE ? n_parked + nn : RT_WAVE;
                    const int *p = &s_stack[threadIdx.x];
                    walker = lane < have;
                    wo_ = f3(__int_as_float(p[0]), __int_as_float(p[RT_BLOCK]), __int_as_float(p[2 * RT_BLOCK]));
                    wd_ = f3(__int_as_float(p[3 * RT_BLOCK]), __int_as_float(p[4 * RT_BLOCK]), __int_as_float(p[5 * RT_BLOCK]));
                    wc_ = f3(__int_as_float(p[6 * RT_BLOCK]), __int_as_float(p[7 * RT_BLOCK]), __int_as_float(p[8 * RT_BLOCK]));
                    wid = (uint32_t)p[9 * RT_BLOCK];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                bool lit = false;
                if (walker) {
                    LaneTrav L = {};
                    trace_shadow_lds(sc_in, s_pairs, s_slots, wo_, wd_, pst, L);
                    lit = !L.found;
                }
                if (lit) {
                    float4 Lr = lbuf[wid];
                    Lr.x += wc_.x; Lr.y += wc_.y; Lr.z += wc_.z;
                    lbuf[wid] = Lr;
                }
                if (pack) {                              // the rays that did not fit are parked now, from column 0
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (has_shadow && rank >= room) {
                        int *p = &s_stack[wave_col] + (rank - room);
                        p[0] = __float_as_int(so.x); p[RT_BLOCK] = __float_as_int(so.y); p[2 * RT_BLOCK] = __float_as_int(so.z);
                        p[3 * RT_BLOCK] = __float_as_int(sd.x); p[4 * RT_BLOCK] = __float_as_int(sd.y); p[5 * RT_BLOCK] = __float_as_int(sd.z);
                        p[6 * RT_BLOCK] = __float_as_int(sc3.x); p[7 * RT_BLOCK] = __float_as_int(sc3.y); p[8 * RT_BLOCK] = __float_as_int(sc3.z);
                        p[9 * RT_BLOCK] = (int)id;
                    }
                    n_parked = n_parked + nn >= RT_WAVE ? n_parked + nn - RT_WAVE : 0;
                }
            } else n_parked += nn;
        } else if (M != RT_MAT_MIRROR) {
            uint32_t spos = seg_base + block_append(has_shadow, &s_out[0]);
            if (has_shadow) {
                sq.id[spos] = id; sq.ox[spos] = so.x; sq.oy[spos] = so.y; sq.oz[spos] = so.z; sq.dx[spos] = sd.x; sq.dy[spos] = sd.y; sq.dz[spos] = sd.z;
                sq.cr[spos] = sc3.x; sq.cg[spos] = sc3.y; sq.cb[spos] = sc3.z;
            }
        }
    }
    };
    shade_list(std::integral_constant<int, MAT>{});
    if (MAT == RT_MAT_MATTE && rp.merge_mirror == 2) {
        __syncthreads();        // every thread is done with the first list's prefix sums
        shade_list(std::integral_constant<int, RT_MAT_MIRROR>{});
    }
    __syncthreads();
    if (threadIdx.x == 0) { bc->shadow[blockIdx.x] = s_out[0]; counts[bounce + 1].queue[blockIdx.x] = s_out[1]; }
    if (FUSE && (threadIdx.x & (RT_WAVE - 1)) == 0 && n_shadow_rays) atomicAdd(&stats[bounce].n_shadow, n_shadow_rays);   // statistics only
}

// PathIntegrator::Li only (integrator.cpp:188-191): a camera ray or a ray after a specular bounce that leaves the scene collects
// Le of every infinite light, through the path's throughput.  A pass of its own over the bounce's queue after the extend kernel,
// so that the extend kernels stay as they are.
__global__ __launch_bounds__(RT_BLOCK) void k_env_escape(DevScene sc, int bounce, PathQueue q, HitBuf hb, uint32_t segcap, float4 *lbuf, BounceCounts *counts) {
    __shared__ uint32_t s_prefix[RT_SEG_BLOCKS + 1], s_wave[RT_BLOCK / RT_WAVE];
    const uint32_t total = segment_prefix(counts[bounce].queue, s_prefix, s_wave);
    const ChunkLoop cl = block_chunks(total, 0);
    uint32_t seg = cl.start < cl.end ? segment_of(s_prefix, cl.start) : 0;
    for (uint32_t base = cl.start; base < cl.end; base += cl.stride) {
        uint32_t g = base + threadIdx.x;
        if (g >= cl.end) continue;
        while (g >= s_prefix[seg + 1]) seg++;
        uint32_t i = seg * segcap + (g - s_prefix[seg]);
        if (hb.slot[i] >= 0) continue;
        uint32_t state = q.state[i];
        if ((state & 0xFFFF) != 0 && !((state >> 16) & 1)) continue;
        F3 d = f3(q.dx[i], q.dy[i], q.dz[i]), T = f3(q.tr[i], q.tg[i], q.tb[i]);
        uint32_t id = q.id[i];
        float4 L = lbuf[id];
        for (int l = 0; l < sc.n_lights; ++l) {
            if (sc.lights[l].kind != RT_LIGHT_INFINITE) continue;
            F3 le = env_le(sc.envs[sc.lights[l].env], d);
            L.x += T.x * le.x; L.y += T.y * le.y; L.z += T.z * le.z;
        }
        lbuf[id] = L;
    }
}

// ------------------------------------------------------------------ finish --
// The late bounces of a pass hold a fraction of a percent of its paths, but every bounce still costs two or three launches of
// 10-60 us each (one chunk per wave: the whole latency chain of a kernel for almost no work).  k_finish takes the queue of one
// bounce and follows every path in it to its end inside one launch — closest hit, emission /
// termination, shading, shadow ray, next ray — with the same device functions in the same order as the kernels it stands in
// for, so the radiance of every path is the same to the last bit.  A lane whose path has ended takes the next one at once.  The host switches to it at the bounce where, going by the
// survival fractions of an earlier pass over the same scene, no more than two chunks per wave are left (rt_render_device).
template <bool LDS_SCENE>
__global__ __launch_bounds__(RT_BLOCK) void k_finish(DevScene sc_in, RenderParams rp, int bounce, PathQueue q, uint32_t segcap, float4 *lbuf, BounceCounts *counts,
                                                      BounceStats *stats, int *spill) {
    __shared__ uint32_t s_prefix[RT_SEG_BLOCKS + 1], s_wave[RT_BLOCK / RT_WAVE];
    extern __shared__ float4 s_dyn[];
    __shared__ float4 s_pairs[LDS_SCENE ? 2 * RT_LDS_SCENE_NODES : 1];
    __shared__ int s_stack[2 * RT_FRAME_RING * RT_BLOCK];
    DevScene sc = sc_in;
    const float4 *t_slots = sc_in.slots;
    if (LDS_SCENE) {
        const int ns = sc_in.n_slots;                              // the table layout of k_shade<.., TABLES = 1, ..>
        float4 *l_slots = s_dyn, *t_frames = l_slots + 3 * ns;
        DevTexture *t_tex = (DevTexture *)(t_frames + 3 * ns);
        DevLight *t_lights = (DevLight *)(t_tex + sc_in.n_textures);
        DevMaterial *t_mats = (DevMaterial *)(t_lights + sc_in.n_lights);
        float *t_uv = (float *)(t_mats + sc_in.n_materials);
        int *t_slot_mat = (int *)(t_uv + 6 * ns);
        copy_words_to_lds(l_slots, sc_in.slots, 48 * ns); copy_words_to_lds(t_frames, sc_in.frames, 48 * ns);
        copy_words_to_lds(t_uv, sc_in.slot_uv, 24 * ns); copy_words_to_lds(t_slot_mat, sc_in.slot_material, 4 * ns);
        copy_words_to_lds(t_mats, sc_in.materials, (int)sizeof(DevMaterial) * sc_in.n_materials);
        copy_words_to_lds(t_tex, sc_in.textures, (int)sizeof(DevTexture) * sc_in.n_textures);
        copy_words_to_lds(t_lights, sc_in.lights, (int)sizeof(DevLight) * sc_in.n_lights);
        sc.slots = l_slots; sc.frames = t_frames; sc.slot_uv = t_uv; sc.slot_material = t_slot_mat; sc.materials = t_mats; sc.textures = t_tex; sc.lights = t_lights;
        t_slots = l_slots;
        stage_pairs_lds(sc_in, s_pairs);
    }
    // LDS scenes: pair records in LDS, eight stack levels are enough; otherwise quad records in global memory and a stack that spills
    PairStackT<RT_FRAME_RING, false> pst; pst.lds_i = &s_stack[threadIdx.x]; pst.lds_t = (float *)&s_stack[RT_FRAME_RING * RT_BLOCK + threadIdx.x]; pst.reset();
    const size_t nthreads = (size_t)gridDim.x * blockDim.x, gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    PairStackT<RT_FRAME_RING, true> gst; gst.lds_i = pst.lds_i; gst.lds_t = pst.lds_t; gst.cap = sc_in.quad_stack_cap;
    gst.spill_i = spill + gtid; gst.spill_t = (float *)(spill + (size_t)gst.cap * nthreads) + gtid; gst.spill_stride = nthreads; gst.reset();
    auto trace_closest = [&](F3 o, F3 d, LaneTrav &L) {
        if (LDS_SCENE) trace_one<false, RT_LDS_SCENE_NODES / 2>(sc_in, s_pairs, t_slots, o, d, RT_INF, pst, L);
        else trace_one<false, 1, true>(sc_in, sc_in.quads, sc_in.slots, o, d, RT_INF, gst, L);
    };
    auto trace_shadow = [&](F3 o, F3 d, LaneTrav &L) {            // tMax = 1 - ShadowEpsilon (interaction.h:27-32)
        if (LDS_SCENE) trace_shadow_lds(sc_in, s_pairs, t_slots, o, d, pst, L);
        else trace_one<true, 1, true>(sc_in, sc_in.quads, sc_in.slots, o, d, 1.f - 0.0001f, gst, L);
    };
    const uint32_t total = segment_prefix(counts[bounce].queue, s_prefix, s_wave);     // its barriers order the copies above before their use
    if (blockIdx.x == 0 && threadIdx.x == 0) stats[bounce].n_queue = total;
    const WaveChunks wc(total);
    uint32_t seg = 0, n_closest = 0, n_shadow = 0;                 // rays beyond the first of each path, shadow rays (statistics)
    // One iteration = one bounce for every lane that holds a path; a lane whose path has ended takes the next one of the wave's
    // chunks at the top of the next iteration, so the wave does not wait for the longest of 64 paths.
    uint32_t w_chunk = 0, c_next = RT_WAVE, c_first = 0, c_count = 0;   // chunks begun; read cursor / first position / size of the current one
    bool busy = false, first = true;
    F3 o = f3(0, 0, 0), d = f3(0, 0, 1), T = f3(0, 0, 0);
    uint32_t id = 0, state = 0;
    float4 Lr = make_float4(0.f, 0.f, 0.f, 0.f);
    while (true) {
        // ---- idle lanes take paths from the current chunk, then from the next ones
        uint64_t im = __ballot(!busy);
        while (im != 0) {
            if (c_next >= c_count) {
                if (w_chunk >= wc.n_mine) break;
                c_first = wc.first(w_chunk++); c_next = 0; c_count = total - c_first < RT_WAVE ? total - c_first : RT_WAVE;
            }
            const uint32_t nidle = (uint32_t)__popcll(im), left = c_count - c_next, take = nidle < left ? nidle : left;
            const uint32_t rank = prefix_in_mask(im);
            if (!busy && rank < take) {
                const uint32_t g = c_first + c_next + rank;
                while (g >= s_prefix[seg + 1]) seg++;
                const uint32_t i = seg * segcap + (g - s_prefix[seg]);
                o = f3(q.ox[i], q.oy[i], q.oz[i]); d = f3(q.dx[i], q.dy[i], q.dz[i]); T = f3(q.tr[i], q.tg[i], q.tb[i]);
                id = q.id[i]; state = q.state[i];
                Lr = lbuf[id];
                busy = true; first = true;
            }
            c_next += take;
            im = __ballot(!busy);
        }
        if (__ballot(busy) == 0) break;
        if (busy) {
            if (!first) n_closest++;
            first = false;
            bool ended = true;
            LaneTrav L = {};
            trace_closest(o, d, L);
            const bool found = L.found;
            const int slot = found ? L.hitSlot : -1;
            const int depth = state & 0xFFFF; const bool spec = (state >> 16) & 1;
            // integrator.cpp:242-257, as in k_extend
            F3 add = f3(0.f, 0.f, 0.f); bool terminal = false;
            if (depth == 0 || spec || sc.n_lights == 0) {
                if (found) {
                    int li = sc_in.slot_light[slot];
                    if (li >= 0) {
                        const DevLight &Lt = sc.lights[li];
                        float4 nf = sc.frames[3 * slot];
                        if (Lt.two_sided || dot3(f3(nf.x, nf.y, nf.z), -d) > 0) add = f3(Lt.Le[0], Lt.Le[1], Lt.Le[2]);   // light.h:73-75
                        if (!black3(add)) terminal = true;
                    }
                } else if (sc.n_lights == 0) {
                    F3 ud = normalize3(d);
                    double t = 0.5 * (ud.y + 1.0);
                    float a = (float)(1.0 - t), b = (float)t;
                    add = f3(1.0f * a + 0.5f * b, 1.0f * a + 0.7f * b, 1.0f * a + 1.0f * b);
                }
            }
            if (!found || depth >= rp.max_depth) terminal = true;
            if (!black3(add)) { Lr.x += T.x * add.x; Lr.y += T.y * add.y; Lr.z += T.z * add.z; }
            if (!terminal) {
                ShadeOut out;
                switch (sc.materials[sc.slot_material[slot]].kind) {
                    case RT_MAT_MATTE: shade_path<RT_MAT_MATTE>(sc, rp, o, d, T, id, state, slot, L.tmax, L.b0, L.b1, L.b2, out); break;
                    case RT_MAT_MIRROR: shade_path<RT_MAT_MIRROR>(sc, rp, o, d, T, id, state, slot, L.tmax, L.b0, L.b1, L.b2, out); break;
                    case RT_MAT_GLASS: shade_path<RT_MAT_GLASS>(sc, rp, o, d, T, id, state, slot, L.tmax, L.b0, L.b1, L.b2, out); break;
                    case RT_MAT_METAL: shade_path<RT_MAT_METAL>(sc, rp, o, d, T, id, state, slot, L.tmax, L.b0, L.b1, L.b2, out); break;
                    case RT_MAT_PLASTIC: shade_path<RT_MAT_PLASTIC>(sc, rp, o, d, T, id, state, slot, L.tmax, L.b0, L.b1, L.b2, out); break;
                    case RT_MAT_ROUGHGLASS: shade_path<RT_MAT_ROUGHGLASS>(sc, rp, o, d, T, id, state, slot, L.tmax, L.b0, L.b1, L.b2, out); break;
                    default: shade_path<RT_MAT_METALROUGH>(sc, rp, o, d, T, id, state, slot, L.tmax, L.b0, L.b1, L.b2, out); break;
                }
                if (out.has_shadow) {                               // bvh.cpp:284-320; tMax = 1 - ShadowEpsilon (interaction.h:27-32)
                    n_shadow++;
                    LaneTrav S = {};
                    trace_shadow(out.so, out.sd, S);
                    if (!S.found) { Lr.x += out.sc3.x; Lr.y += out.sc3.y; Lr.z += out.sc3.z; }
                }
                if (out.has_next) { o = out.no; d = out.nd; T = out.nT; state = out.nstate; ended = false; }
            }
            if (ended) { lbuf[id] = Lr; busy = false; }
        }
    }
    if (n_closest) atomicAdd(&stats[bounce + 1].n_queue, n_closest);      // statistics only
    if (n_shadow) atomicAdd(&stats[bounce].n_shadow, n_shadow);
}

// ------------------------------------------------------------------ shadow --
template <bool LDS_SCENE>
__global__ __launch_bounds__(RT_BLOCK) void k_shadow(DevScene sc, int bounce, ShadowQueue sq, uint32_t segcap, float4 *lbuf, BounceCounts *counts,
                                                      BounceStats *stats, int *spill, int fast_walk) {
    __shared__ int s_stack[RT_LDS_STACK * RT_BLOCK];
    __shared__ float4 s_nodes[LDS_SCENE ? 2 * RT_LDS_SCENE_NODES : 1], s_slots[LDS_SCENE ? 3 * RT_LDS_SCENE_SLOTS : 1];
    // fast_walk on an LDS scene: the LDS copy holds pair records (k-th float4 of record r at s_nodes[k * 64 + r]) instead of nodes; the int
    // stack area then carries (node, tMin) entries for 8 levels, which the host guarantees to be enough
    if (LDS_SCENE && (fast_walk & 1)) {
        stage_pairs_lds(sc, s_nodes);
        for (int i = threadIdx.x; i < 3 * sc.n_slots; i += RT_BLOCK) s_slots[i] = sc.slots[i];
    } else if (LDS_SCENE) stage_scene_lds(sc, s_nodes, s_slots);
    PairStackT<8, false> pst; pst.lds_i = &s_stack[threadIdx.x]; pst.lds_t = (float *)&s_stack[8 * RT_BLOCK + threadIdx.x]; pst.reset();
    __shared__ uint32_t s_prefix[RT_SEG_BLOCKS + 1], s_wave[RT_BLOCK / RT_WAVE];
    BounceCounts *bc = counts + bounce;
    const uint32_t total = segment_prefix(bc->shadow, s_prefix, s_wave);
    if (blockIdx.x == 0 && threadIdx.x == 0) stats[bounce].n_shadow += total;      // (one launch per light and bounce when every light is sampled)
    size_t nthreads = (size_t)gridDim.x * blockDim.x, gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    TravStack st; st.lds = &s_stack[threadIdx.x]; st.spill = spill + gtid; st.spill_stride = nthreads; st.reset();
    PairStackT<8, true> gst; gst.lds_i = pst.lds_i; gst.lds_t = pst.lds_t; gst.cap = sc.quad_stack_cap;
    gst.spill_i = spill + gtid; gst.spill_t = (float *)(spill + (size_t)gst.cap * nthreads) + gtid; gst.spill_stride = nthreads; gst.reset();
    InfoStack<8> ist; ist.lds = pst.lds_i; ist.spill = spill + gtid; ist.spill_stride = nthreads; ist.cap = gst.cap; ist.reset();
    const ChunkLoop cl = block_chunks(total, fast_walk & 2);
    uint32_t seg = cl.start < cl.end ? segment_of(s_prefix, cl.start) : 0;
    for (uint32_t base = cl.start; base < cl.end; base += cl.stride) {
        uint32_t g = base + threadIdx.x;
        if (g >= cl.end) break;
        while (g >= s_prefix[seg + 1]) seg++;
        uint32_t i = seg * segcap + (g - s_prefix[seg]);
        F3 o = f3(sq.ox[i], sq.oy[i], sq.oz[i]), d = f3(sq.dx[i], sq.dy[i], sq.dz[i]);
        HitRec h; uint32_t c0 = 0, c1 = 0;
        bool occluded;   // tMax = 1 - ShadowEpsilon
        if (LDS_SCENE && (fast_walk & 1)) {
            LaneTrav L = {};
            trace_shadow_lds(sc, s_nodes, s_slots, o, d, pst, L);
            occluded = L.found;
        } else if (!LDS_SCENE && (fast_walk & 1)) {
            LaneTrav L = {};
#if RT_ANY_DEFERRED
            trace_any_deferred_quads<8>(sc, sc.quads, sc.slots, o, d, 1.f - 0.0001f, ist, (int *)pst.lds_t, L);
#else
            trace_one<true, 1, true>(sc, sc.quads, sc.slots, o, d, 1.f - 0.0001f, gst, L);
#endif
            occluded = L.found;
        } else if (LDS_SCENE) { LdsFetch lf; lf.nodes = s_nodes; lf.slots = s_slots; occluded = traverse<true, false>(sc, lf, o, d, 1.f - 0.0001f, st, h, c0, c1); }
        else { GlobalFetch gf; gf.nodes = sc.nodes; gf.slots = sc.slots; occluded = traverse<true, false>(sc, gf, o, d, 1.f - 0.0001f, st, h, c0, c1); }
        if (!occluded) {
            uint32_t id = sq.id[i];
            float4 L = lbuf[id];
            L.x += sq.cr[i]; L.y += sq.cg[i]; L.z += sq.cb[i];
            lbuf[id] = L;
        }
    }
}

// k_shadow for scenes that live in LDS, with the carry-over of k_extend_carry: any-hit walks end after very different numbers of
// steps (first hit ends the ray), so a wave that runs 64 shadow rays to completion spends most of its time with a handful of
// lanes.  Here a lane whose ray is finished after RT_CARRY_ROUNDS rounds adds the unoccluded light to its path's radiance slot and
// takes the next shadow ray of the wave's chunks.  Results are those of k_shadow bit for bit: the same walk, and each path's slot is
// touched once per bounce by exactly one lane.
__global__ __launch_bounds__(RT_BLOCK) void k_shadow_carry(DevScene sc, int bounce, ShadowQueue sq, uint32_t segcap, float4 *lbuf, BounceCounts *counts,
                                                            BounceStats *stats) {
    __shared__ int s_stack[2 * RT_FRAME_RING * RT_BLOCK];
    __shared__ float4 s_nodes[2 * RT_LDS_SCENE_NODES], s_slots[3 * RT_LDS_SCENE_SLOTS];
    stage_pairs_lds(sc, s_nodes);
    for (int i = threadIdx.x; i < 3 * sc.n_slots; i += RT_BLOCK) s_slots[i] = sc.slots[i];
    PairStackT<RT_FRAME_RING, false> pst; pst.lds_i = &s_stack[threadIdx.x]; pst.lds_t = (float *)&s_stack[RT_FRAME_RING * RT_BLOCK + threadIdx.x]; pst.reset();
    __shared__ uint32_t s_prefix[RT_SEG_BLOCKS + 1], s_wave[RT_BLOCK / RT_WAVE];
    BounceCounts *bc = counts + bounce;
    const uint32_t total = segment_prefix(bc->shadow, s_prefix, s_wave);
    if (blockIdx.x == 0 && threadIdx.x == 0) stats[bounce].n_shadow += total;      // (one launch per light and bounce when every light is sampled)
    const uint32_t n_waves = gridDim.x * (RT_BLOCK / RT_WAVE), gwave = blockIdx.x * (RT_BLOCK / RT_WAVE) + threadIdx.x / RT_WAVE;
    const uint32_t n_chunks = (total + RT_WAVE - 1) / RT_WAVE;
    const uint32_t w_hi = (gwave < n_chunks ? (n_chunks - gwave + n_waves - 1) / n_waves : 0) * RT_WAVE;   // this wave's positions: logical [0, w_hi)
    uint32_t w_next = 0, seg = 0;
    const int rounds = n_chunks <= n_waves ? 1 << 20 : RT_CARRY_ROUNDS;
    LaneTrav L = {};
    L.mode = RT_LANE_DONE; L.hitSlot = -1;
    bool busy = false;
    uint32_t cur_i = 0, dummy = 0;
    while (true) {
        // ---- hand in finished rays: an unoccluded ray delivers its light (light.cpp:19-21, integrator.cpp:97-98)
        if (busy && L.mode == RT_LANE_DONE) {
            if (!L.found) {
                const uint32_t id = sq.id[cur_i];
                float4 Lr = lbuf[id];
                Lr.x += sq.cr[cur_i]; Lr.y += sq.cg[cur_i]; Lr.z += sq.cb[cur_i];
                lbuf[id] = Lr;
            }
            busy = false;
        }
        // ---- take the next rays of the wave's chunks
        const uint64_t im = __ballot(!busy);
        const uint32_t left = w_hi - w_next, nidle = (uint32_t)__popcll(im), take = nidle < left ? nidle : left;
        const uint32_t ln = w_next + prefix_in_mask(im);                  // logical position -> queue position
        const uint32_t g = ((ln / RT_WAVE) * n_waves + gwave) * RT_WAVE + (ln & (RT_WAVE - 1));
        if (!busy && prefix_in_mask(im) < take && g < total) {
            while (g >= s_prefix[seg + 1]) seg++;
            cur_i = seg * segcap + (g - s_prefix[seg]);
            F3 o = f3(sq.ox[cur_i], sq.oy[cur_i], sq.oz[cur_i]), d = f3(sq.dx[cur_i], sq.dy[cur_i], sq.dz[cur_i]);
            bool have = lane_begin<false>(sc, L, pst, o, d, 1.f - 0.0001f, dummy);     // tMax = 1 - ShadowEpsilon (interaction.h:27-32)
            lane_enter(sc, L, sc.root_info, have);
            busy = true;
        }
        w_next += take;
        if (__ballot(busy) == 0) break;
#pragma unroll 1
        for (int k = 0; k < rounds; ++k) {
            while (L.mode == RT_LANE_WALK) {
                uint32_t info; bool hv;
                if (L.plain) lane_walk<false, true, RT_LDS_SCENE_NODES / 2, true>(s_nodes, L, pst, dummy, info, hv);
                else lane_walk<false, false, RT_LDS_SCENE_NODES / 2, true>(s_nodes, L, pst, dummy, info, hv);
                lane_enter(sc, L, info, hv);
            }
            if (L.mode == RT_LANE_LEAF) {
                LeafOut r = lane_leaf<true, false, true>(sc, s_slots, L, pst, dummy);
                L.found = r.found;
                lane_enter(sc, L, r.info, r.have);
            }
            if (__ballot(L.mode != RT_LANE_DONE) == 0) break;
        }
    }
}

// ------------------------------------------------ persistent-lane frame tracing --
// k_extend / k_shadow above give every lane one ray per round and the wave waits for its slowest lane: fine while rays
// take similar numbers of steps, wasteful on deep trees with incoherent rays.  The kernels below walk the same queues with
// persistent lanes over records in HBM (rt_traverse2.h): a lane that finishes its ray takes the next one of its wave's range
// at once.  Rays are staged through LDS a batch of 64 at a time (coalesced loads issued one batch ahead), so a refill never
// waits on global memory; results go out with plain stores; everything that needs a global read-modify-write or a
// wave-wide append (emission, class binning, adding the unoccluded light) runs afterwards as a second, coalesced sweep
// over the block's share.
enum { RT_WALK_QUADS = 1, RT_WALK_PAIRS = 2 };
#ifndef RT_FREE_SPECULATE
#define RT_FREE_SPECULATE 0         // order-free persistent walk: lanes that hold an untested leaf walk ahead (trace_share); measured slower, see there
#endif
#ifndef RT_FREE_LEAF_MIN
#define RT_FREE_LEAF_MIN 32         // ... and the leaves are tested once this many lanes of the wave hold one
#endif

struct RayArrays { const float *ox, *oy, *oz, *dx, *dy, *dz; };
#ifdef RT_PHASE_STATS       // diagnostic build only: rounds and busy lanes per phase of trace_share ([0..5] closest hit, [6..11] any hit)
__device__ unsigned long long g_phase[12];
extern "C" void rt_debug_phase(unsigned long long *out, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 12);
    if (reset) { unsigned long long z[12] = {}; hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof z); }
}
#endif

// Traverse this wave's chunks of the `total` concatenated positions of a segmented queue; sink(i, L) receives each finished ray
// (i = its position in the queue arrays).  recs = pair or quad records (LDS or global), slots = leaf slots.
// FREE (closest hit over quad records): the order-free walk of rt_traverse2.h; the sink sees L.amb on rays that have to be walked again
// in the reference's order.
// OCT (with FREE): over the eight-wide records with quantised boxes (rt_oct.h); recs then points at those.
template <bool ANY, int MODE, bool FREE = false, bool OCT = false, class Stack, class Sink>
__device__ __forceinline__ void trace_share(const DevScene &sc, const float4 *recs, const float4 *slots, RayArrays ra, float tmax0, uint32_t total,
                                            const uint32_t *s_prefix, uint32_t segcap, Stack &st, float *s_ray, Sink sink) {
    const uint32_t lane = lane_id();
    const WaveChunks wc(total);
    uint32_t w_chunk = 0;                                             // chunks taken so far
    uint32_t seg = 0;                                                 // per-lane cursor; a lane's positions only grow
    float p_ox = 0.f, p_oy = 0.f, p_oz = 0.f, p_dx = 0.f, p_dy = 0.f, p_dz = 1.f; uint32_t p_i = 0;   // the batch being prefetched
    uint32_t pf_count = 0;                                            // rays in the prefetch registers (wave-uniform)
    auto prefetch = [&]() {
        pf_count = 0;
        if (w_chunk < wc.n_mine) {
            const uint32_t g0 = wc.first(w_chunk++);
            pf_count = total - g0 < RT_WAVE ? total - g0 : RT_WAVE;
            if (lane < pf_count) {
                uint32_t g = g0 + lane;
                while (g >= s_prefix[seg + 1]) seg++;
                p_i = seg * segcap + (g - s_prefix[seg]);
                p_ox = ra.ox[p_i]; p_oy = ra.oy[p_i]; p_oz = ra.oz[p_i]; p_dx = ra.dx[p_i]; p_dy = ra.dy[p_i]; p_dz = ra.dz[p_i];
            }
        }
    };
    prefetch();
    uint32_t avail = 0, rd = 0;                                       // rays left in the LDS batch, read cursor (wave-uniform)
    LaneTrav L = {};
    L.mode = RT_LANE_DONE; L.hitSlot = -1;
    bool busy = false;                                                // the lane holds a ray (possibly finished, not yet handed to the sink)
    uint32_t cur_i = 0, dummy = 0;
    // Every lane is walking, waiting at a leaf, or idle (finished / never started).  Each round the wave runs ONE of the
    // three phases, the one most lanes are waiting for, so the instructions of a phase are issued for a well-filled wave.
    // Measured on the helmet over the eight-wide records (1440 x 1440, later bounces): 53 % of the rounds walk with 33 lanes on average,
    // 35 % test leaves with 29, 13 % refill with 36; a ray takes 4.75 steps and 2.75 leaf visits.  Two things that did not help: weighting
    // the vote by what a round of each phase costs (a refill round is cheap, so run it for fewer lanes: weights 1:1:4 to 2:3:6, frame
    // 605 -> 606-612 Msamples/s), and two rays per lane, the one that is in the round's phase taking part (rounds -13 %, walking lanes
    // 33 -> 38, but the exchange of the two register sets and three blocks a CU instead of five: 605 -> 470-535).
    // FREE: a lane that holds a leaf still to be tested need not wait for the leaf phase: the order of the visits is free, so in a walk
    // phase it takes the next record within reach from its stack and expands it, all grandchildren going back on the stack (the leaf's
    // primitives may yet shorten the reach: what was pushed is looked at again when it is popped).  The lanes that wait for the majority's
    // phase thus do work that is needed unless the pending leaf turns out to hold the hit.  A lane whose stack top is another leaf, or
    // empty, cannot walk ahead (spec_blocked) until its leaf has been tested.
    bool spec_blocked = false;
#ifdef RT_PHASE_STATS
    unsigned long long ps[6] = {};
#endif
    while (true) {
        const bool walk = busy && L.mode == RT_LANE_WALK, leaf = busy && L.mode == RT_LANE_LEAF, idle = !busy || L.mode == RT_LANE_DONE;
        const uint64_t im = __ballot(idle);
        const uint32_t nW = (uint32_t)__popcll(__ballot(walk)), nL = (uint32_t)__popcll(__ballot(leaf));
        const uint32_t supply = avail + pf_count, nIdle = (uint32_t)__popcll(im), nI = nIdle < supply ? nIdle : supply;
        if (nW == 0 && nL == 0 && nI == 0) break;
        uint32_t info = 0; bool have = false, moved = false;
        float entry = 0.f;
        const bool can_spec = FREE && RT_FREE_SPECULATE && leaf && !spec_blocked;
        const uint32_t nS = FREE && RT_FREE_SPECULATE ? (uint32_t)__popcll(__ballot(can_spec)) : 0u;
        // which phase: without walking ahead, the one most lanes wait for; with it, leaves are tested once RT_FREE_LEAF_MIN lanes hold one or
        // nobody has a record of its own to walk, otherwise walkers and lanes walking ahead against the idle lanes
        const bool leaf_phase = FREE && RT_FREE_SPECULATE ? (nL >= RT_FREE_LEAF_MIN || (nW == 0 && nL > 0 && nL >= nI)) : (!(nW >= nL && nW >= nI) && nL >= nI);
        const bool walk_phase = !leaf_phase && (FREE && RT_FREE_SPECULATE ? (nW + nS > 0 && nW + nS >= nI) : (nW >= nL && nW >= nI));
#ifdef RT_PHASE_STATS
        if (lane == 0) { if (walk_phase) { ps[0]++; ps[1] += nW; } else if (leaf_phase) { ps[2]++; ps[3] += nL; } else { ps[4]++; ps[5] += nI; } }
#endif
        if (walk_phase) {
            if (FREE) {
                uint32_t node = (uint32_t)L.cur; bool step = walk;
                if (can_spec) {
                    uint32_t e; float et;
                    if (stack_next_free(free_reach(L), st, e, et)) {
                        if (e & RT_PAIR_LEAF) { st.push((int)e, et); spec_blocked = true; }
                        else { node = e; step = true; }
                    } else spec_blocked = true;
                }
                if (step) { if (OCT) oct_step_free(recs, L, node, !walk, st, info, have, entry); else quad_step_free(recs, L, node, !walk, st, info, have, entry); }
            }
            else if (MODE == RT_WALK_QUADS) {
                if (__ballot(walk && !L.plain) == 0) { if (walk) lane_walk_quad<true, ANY>(recs, L, st, info, have); }     // the whole wave takes one version
                else if (walk) lane_walk_quad<false, ANY>(recs, L, st, info, have);
            } else {
                if (__ballot(walk && !L.plain) == 0) { if (walk) lane_walk<false, true, 1, ANY>(recs, L, st, dummy, info, have); }
                else if (walk) lane_walk<false, false, 1, ANY>(recs, L, st, dummy, info, have);
            }
            moved = walk;
        } else if (leaf_phase) {
            if (FREE) {
                if (leaf) {
                    LeafOutFree r = OCT ? oct_leaf_free<ANY>(sc, slots, L, st) : lane_leaf_free(sc, slots, L, st);
                    L.found = r.found; L.amb = r.amb; L.tmax = r.tmax; L.hitSlot = r.hitSlot; L.b0 = r.b0; L.b1 = r.b1; L.b2 = r.b2;
                    info = r.info; have = r.have; entry = r.entry;
                    spec_blocked = false;
                }
            } else if (leaf) {
                LeafOut r = lane_leaf<ANY, false, ANY>(sc, slots, L, st, dummy);
                L.found = r.found; L.tmax = r.tmax; L.hitSlot = r.hitSlot; L.b0 = r.b0; L.b1 = r.b1; L.b2 = r.b2;
                info = r.info; have = r.have;
            }
            moved = leaf;
        } else {
            // (the batch switch comes before the result stores so that its wait on the prefetched loads does not also wait on them)
            if (avail == 0) {                                         // LDS batch used up: the prefetched one becomes current (supply > 0 here)
                if (lane < pf_count) {
                    s_ray[lane] = p_ox; s_ray[RT_WAVE + lane] = p_oy; s_ray[2 * RT_WAVE + lane] = p_oz; s_ray[3 * RT_WAVE + lane] = p_dx;
                    s_ray[4 * RT_WAVE + lane] = p_dy; s_ray[5 * RT_WAVE + lane] = p_dz; s_ray[6 * RT_WAVE + lane] = __uint_as_float(p_i);
                }
                avail = pf_count; rd = 0;
                prefetch();
            }
            if (busy && L.mode == RT_LANE_DONE) { sink(cur_i, L); busy = false; }
            const uint32_t rank = prefix_in_mask(im), take = nIdle < avail ? nIdle : avail;
            if (idle && rank < take) {
                uint32_t k = rd + rank;
                F3 po = f3(s_ray[k], s_ray[RT_WAVE + k], s_ray[2 * RT_WAVE + k]), pd = f3(s_ray[3 * RT_WAVE + k], s_ray[4 * RT_WAVE + k], s_ray[5 * RT_WAVE + k]);
                cur_i = __float_as_uint(s_ray[6 * RT_WAVE + k]);
                busy = true; moved = true;
                if (FREE) have = lane_begin_free(sc, L, st, po, pd, tmax0); else have = lane_begin<false>(sc, L, st, po, pd, tmax0, dummy);
                info = OCT ? 0u : sc.root_info;
                if (OCT && have && !oct_ray_ok(L)) { L.amb = true; have = false; }
            }
            rd += take; avail -= take;
        }
        if (moved) { lane_enter(sc, L, info, have); if (FREE) L.entry = entry; }
    }
    if (busy && L.mode == RT_LANE_DONE) sink(cur_i, L);
#ifdef RT_PHASE_STATS
    if (lane == 0) for (int k = 0; k < 6; ++k) atomicAdd(&g_phase[(ANY ? 6 : 0) + k], ps[k]);
#endif
}

template <int MODE, bool FREE = false, bool OCT = false>
__global__ __launch_bounds__(RT_BLOCK, 5) void k_extend_pl(DevScene sc, RenderParams rp, int bounce, PathQueue q, HitBuf hb, uint32_t *cls_list, uint32_t cap,
                                                         uint32_t segcap, float4 *lbuf, BounceCounts *counts, BounceStats *stats, int *spill) {
    __shared__ int s_si[RT_FRAME_RING * RT_BLOCK];
    __shared__ float s_st[RT_FRAME_RING * RT_BLOCK];
    __shared__ float s_rays[(RT_BLOCK / RT_WAVE) * 7 * RT_WAVE];
    __shared__ uint32_t s_prefix[RT_SEG_BLOCKS + 1], s_wave[RT_BLOCK / RT_WAVE], s_cls[RT_NUM_CLASSES];
    const float4 *recs = OCT ? sc.octs : (MODE == RT_WALK_QUADS ? sc.quads : sc.pairs), *slots = sc.slots;
    BounceCounts *bc = counts + bounce;
    if (threadIdx.x < RT_NUM_CLASSES) s_cls[threadIdx.x] = 0;
    const uint32_t total = segment_prefix(bc->queue, s_prefix, s_wave);
    if (blockIdx.x == 0 && threadIdx.x == 0) stats[bounce].n_queue = total;
    // the shading kernels of this bounce append to segment blockIdx.x of the shadow queue and of the next queue
    if (threadIdx.x == 0) { bc->shadow[blockIdx.x] = 0; counts[bounce + 1].queue[blockIdx.x] = 0; }
    size_t nthreads = (size_t)gridDim.x * blockDim.x, gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    PairStackT<RT_FRAME_RING> st; st.lds_i = &s_si[threadIdx.x]; st.lds_t = &s_st[threadIdx.x];
    st.cap = OCT ? sc.oct_stack_cap : (MODE == RT_WALK_QUADS ? sc.quad_stack_cap : RT_MAX_STACK);
    st.spill_i = spill + gtid; st.spill_t = (float *)(spill + (size_t)st.cap * nthreads) + gtid; st.spill_stride = nthreads; st.reset();
    RayArrays ra = {q.ox, q.oy, q.oz, q.dx, q.dy, q.dz};
    auto sink = [&](uint32_t i, const LaneTrav &L) {
        // (order-free walk: -2 marks a ray the second sweep walks again in the reference's order)
        hb.slot[i] = (FREE && L.amb) ? -2 : (L.found ? L.hitSlot : -1); hb.t[i] = L.tmax; hb.b0[i] = L.b0; hb.b1[i] = L.b1; hb.b2[i] = L.b2;
    };
    trace_share<false, MODE, FREE, OCT>(sc, recs, slots, ra, RT_INF, total, s_prefix, segcap, st, &s_rays[(threadIdx.x / RT_WAVE) * 7 * RT_WAVE], sink);
    __threadfence_block();
    __syncthreads();
    uint32_t n_replayed = 0;
    // second sweep, coalesced, every wave over its own chunks: emission / sky / termination (integrator.cpp:242-257) and class binning
    uint32_t seg = 0;
    const uint32_t seg_base = blockIdx.x * segcap;
    const WaveChunks wc(total);
    for (uint32_t k = 0; k < wc.n_mine; ++k) {
        uint32_t g = wc.first(k) + lane_id();
        bool active = g < total;
        int cls = -1;
        uint32_t i = 0;
        if (active) {
            while (g >= s_prefix[seg + 1]) seg++;
            i = seg * segcap + (g - s_prefix[seg]);
            int slot = hb.slot[i];
            if (FREE && slot == -2) {      // near-tie (or a ray that is not plain): BVHAccel::Intersect's own order decides
                LaneTrav R = {};
                trace_one<false, 1, true>(sc, sc.quads, sc.slots, f3(q.ox[i], q.oy[i], q.oz[i]), f3(q.dx[i], q.dy[i], q.dz[i]), RT_INF, st, R);
                slot = R.found ? R.hitSlot : -1;
                hb.slot[i] = slot; hb.t[i] = R.tmax; hb.b0[i] = R.b0; hb.b1[i] = R.b1; hb.b2[i] = R.b2;
                n_replayed++;
            }
            bool found = slot >= 0;
            uint32_t state = q.state[i];
            int depth = state & 0xFFFF; bool spec = (state >> 16) & 1;
            F3 add = f3(0.f, 0.f, 0.f); bool terminal = false;
            if (depth == 0 || spec || (sc.n_lights == 0 && !rp.li)) {      // rp.li: integrator.cpp:185-192, emission does not end the path and there is no sky
                if (found) {
                    int li = sc.slot_light[slot];
                    if (li >= 0) {
                        F3 d = f3(q.dx[i], q.dy[i], q.dz[i]);
                        const DevLight &Lt = sc.lights[li];
                        float4 nf = sc.frames[3 * slot];
                        if (Lt.two_sided || dot3(f3(nf.x, nf.y, nf.z), -d) > 0) add = f3(Lt.Le[0], Lt.Le[1], Lt.Le[2]);   // light.h:73-75
                        if (!black3(add) && !rp.li) terminal = true;
                    }
                } else if (sc.n_lights == 0 && !rp.li) {
                    F3 ud = normalize3(f3(q.dx[i], q.dy[i], q.dz[i]));
                    double t = 0.5 * (ud.y + 1.0);
                    float a = (float)(1.0 - t), b = (float)t;
                    add = f3(1.0f * a + 0.5f * b, 1.0f * a + 0.7f * b, 1.0f * a + 1.0f * b);
                }
            }
            if (!found || depth >= rp.max_depth) terminal = true;
            if (!black3(add)) {
                uint32_t id = q.id[i];
                float4 Lr = lbuf[id];
                Lr.x += q.tr[i] * add.x; Lr.y += q.tg[i] * add.y; Lr.z += q.tb[i] * add.z;
                lbuf[id] = Lr;
            }
            if (!terminal) { cls = sc.materials[sc.slot_material[slot]].kind; if (rp.merge_mirror == 1 && cls == RT_MAT_MIRROR) cls = RT_MAT_MATTE; }
        }
        for (int c = 0; c < RT_NUM_CLASSES; ++c) {
            if (!((rp.class_mask >> c) & 1u)) continue;
            bool mine = (cls == c);
            if (__ballot(mine) == 0) continue;
            uint32_t pos = block_append(mine, &s_cls[c]);
            if (mine) cls_list[(size_t)c * cap + seg_base + pos] = i;
        }
    }
    __syncthreads();
    if (threadIdx.x < RT_NUM_CLASSES) bc->cls[threadIdx.x][blockIdx.x] = s_cls[threadIdx.x];
    if (FREE && n_replayed) atomicAdd(&stats[bounce].n_replay, n_replayed);      // statistics only
}

// OCT: the any-hit walk over the eight-wide records with quantised boxes (rt_oct.h; the order of IntersectP's visits is free anyway, and the
// leaves are tested only behind their exact boxes); rays the record-space slab test cannot take (zero, tiny or huge reciprocal direction
// components) are marked 2 and walked over the quad records in the second sweep.
template <int MODE, bool OCT = false>
__global__ __launch_bounds__(RT_BLOCK, 5) void k_shadow_pl(DevScene sc, int bounce, ShadowQueue sq, int *occluded, uint32_t segcap, float4 *lbuf, BounceCounts *counts,
                                                         BounceStats *stats, int *spill) {
    __shared__ int s_si[RT_FRAME_RING * RT_BLOCK];
    __shared__ float s_st[RT_FRAME_RING * RT_BLOCK];
    __shared__ float s_rays[(RT_BLOCK / RT_WAVE) * 7 * RT_WAVE];
    __shared__ uint32_t s_prefix[RT_SEG_BLOCKS + 1], s_wave[RT_BLOCK / RT_WAVE];
    const float4 *recs = OCT ? sc.octs : (MODE == RT_WALK_QUADS ? sc.quads : sc.pairs), *slots = sc.slots;
    BounceCounts *bc = counts + bounce;
    const uint32_t total = segment_prefix(bc->shadow, s_prefix, s_wave);
    if (blockIdx.x == 0 && threadIdx.x == 0) stats[bounce].n_shadow += total;      // (one launch per light and bounce when every light is sampled)
    size_t nthreads = (size_t)gridDim.x * blockDim.x, gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    PairStackT<RT_FRAME_RING> st; st.lds_i = &s_si[threadIdx.x]; st.lds_t = &s_st[threadIdx.x];
    st.cap = OCT ? sc.oct_stack_cap : (MODE == RT_WALK_QUADS ? sc.quad_stack_cap : RT_MAX_STACK);
    st.spill_i = spill + gtid; st.spill_t = (float *)(spill + (size_t)st.cap * nthreads) + gtid; st.spill_stride = nthreads; st.reset();
    RayArrays ra = {sq.ox, sq.oy, sq.oz, sq.dx, sq.dy, sq.dz};
    // tMax = 1 - ShadowEpsilon (interaction.h SpawnRayTo)
    // (Deferring the leaves here as trace_any_deferred_quads does was measured on the helmet: 0.235 s against 0.237 s for the frame's shadow rays — lanes
    // that refill gain from the early exit of an occluded ray, and the phase vote already keeps them busy.)
    auto sink = [&](uint32_t i, const LaneTrav &L) { occluded[i] = (OCT && L.amb) ? 2 : (L.found ? 1 : 0); };
    trace_share<true, MODE, OCT, OCT>(sc, recs, slots, ra, 1.f - 0.0001f, total, s_prefix, segcap, st, &s_rays[(threadIdx.x / RT_WAVE) * 7 * RT_WAVE], sink);
    __threadfence_block();
    __syncthreads();
    uint32_t seg = 0;
    const WaveChunks wc(total);
    for (uint32_t k = 0; k < wc.n_mine; ++k) {
        uint32_t g = wc.first(k) + lane_id();
        if (g >= total) break;
        while (g >= s_prefix[seg + 1]) seg++;
        uint32_t i = seg * segcap + (g - s_prefix[seg]);
        int occ = occluded[i];
        if (OCT && occ == 2) {
            LaneTrav R = {};
            trace_one<true, 1, true>(sc, sc.quads, sc.slots, f3(sq.ox[i], sq.oy[i], sq.oz[i]), f3(sq.dx[i], sq.dy[i], sq.dz[i]), 1.f - 0.0001f, st, R);
            occ = R.found ? 1 : 0;
        }
        if (!occ) {
            uint32_t id = sq.id[i];
            float4 Lr = lbuf[id];
            Lr.x += sq.cr[i]; Lr.y += sq.cg[i]; Lr.z += sq.cb[i];
            lbuf[id] = Lr;
        }
    }
}

// -------------------------------------------------------------- accumulate --
__global__ __launch_bounds__(RT_BLOCK) void k_accumulate(const float4 *lbuf, int spb, uint32_t npix, float *sum_rgb) {
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        float r = sum_rgb[3 * (size_t)p], g = sum_rgb[3 * (size_t)p + 1], b = sum_rgb[3 * (size_t)p + 2];
        for (int s = 0; s < spb; ++s) { float4 L = lbuf[(size_t)p * spb + s]; r += L.x; g += L.y; b += L.z; }
        sum_rgb[3 * (size_t)p] = r; sum_rgb[3 * (size_t)p + 1] = g; sum_rgb[3 * (size_t)p + 2] = b;
    }
}
// The same sum for passes with many samples per pixel: one wave per pixel reads the pixel's contiguous run of radiance slots
// with coalesced loads (lane l takes samples l, l + 64, ...), then a fixed-order butterfly adds the 64 partial sums, so the
// result does not depend on scheduling.
__global__ __launch_bounds__(RT_BLOCK) void k_accumulate_wave(const float4 *lbuf, int spb, uint32_t npix, float *sum_rgb) {
    const uint32_t lane = threadIdx.x & (RT_WAVE - 1), nwaves = gridDim.x * (RT_BLOCK / RT_WAVE);
    for (uint32_t p = blockIdx.x * (RT_BLOCK / RT_WAVE) + threadIdx.x / RT_WAVE; p < npix; p += nwaves) {
        const float4 *src = lbuf + (size_t)p * spb;
        float r = 0.f, g = 0.f, b = 0.f;
        for (int s = (int)lane; s < spb; s += RT_WAVE) { float4 L = src[s]; r += L.x; g += L.y; b += L.z; }
#pragma unroll
        for (int off = RT_WAVE / 2; off > 0; off >>= 1) { r += __shfl_down(r, off); g += __shfl_down(g, off); b += __shfl_down(b, off); }
        if (lane == 0) { sum_rgb[3 * (size_t)p] += r; sum_rgb[3 * (size_t)p + 1] += g; sum_rgb[3 * (size_t)p + 2] += b; }
    }
}

// ==================================================================== host ==
// Two HIP events that are destroyed on every way out of the function that owns them.
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t create() { hipError_t e = hipEventCreate(&a); return e != hipSuccess ? e : hipEventCreate(&b); }
    ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};
// Per-class kernel times of a pass (RT_RENDER_TIME_KERNELS): an event after every launch, tagged with the class of that launch;
// the time between two consecutive events goes to the later one's class.
enum { RT_TIME_TRACE = 0, RT_TIME_SHADE = 1, RT_TIME_SHADOW = 2, RT_TIME_OTHER = 3 };
struct KernelTimer {
    bool on = false; hipStream_t stream = nullptr;
    std::vector<hipEvent_t> ev; std::vector<int> tag; size_t used = 0;
    double seconds[4] = {0, 0, 0, 0};
    ~KernelTimer() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
    void mark(int cls) {
        if (!on) return;
        if (used == ev.size()) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) { on = false; return; } ev.push_back(e); tag.push_back(0); }
        tag[used] = cls; (void)hipEventRecord(ev[used++], stream);
    }
    void begin_pass() { used = 0; mark(RT_TIME_OTHER); }
    void end_pass() {      // the stream has been synchronised
        for (size_t i = 1; i < used; ++i) { float ms = 0.f; if (hipEventElapsedTime(&ms, ev[i - 1], ev[i]) == hipSuccess) seconds[tag[i]] += (double)ms * 1e-3; }
        used = 0;
    }
};

// Scratch of rt_trace_closest_device_noalloc: [0, 16) the two work counters of the counting kernel, [256, ...) the spill area of
// the per-lane traversal stacks.  Every GPU the library runs on holds its own copy until the process ends.  In host code the
// name below is only a handle for hipGetSymbolAddress (rt_init); the address it yields, kept per context, is what host code
// passes on.
#define RT_NOALLOC_SPILL_BYTES ((size_t)32 << 20)
#define RT_NOALLOC_SCRATCH_BYTES (RT_NOALLOC_SPILL_BYTES + 256)
#define RT_NOALLOC_MAX_RAYS ((size_t)1 << 28)
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;
    int n_cu = 256;
    unsigned char *scratch_ptr = nullptr;     // this device's copy of `scratch`
    hipEvent_t ev_a = nullptr, ev_b = nullptr; // made once, so that the entry point that must not allocate can time its launches
};
static std::string g_init_error;
static std::mutex g_error_mutex;      // render lanes report errors from their own threads
static void set_error(rt_ctx *c, const std::string &s) { std::lock_guard<std::mutex> g(g_error_mutex); if (c) c->error = s; else g_init_error = s; }

struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t b) { if (p) { (void)hipFree(p); p = nullptr; } bytes = b; return b ? hipMalloc(&p, b) : hipSuccess; }
    hipError_t ensure(size_t b) { if (b <= bytes && p) return hipSuccess; return alloc(b); }
};

#define RT_MAX_LANES 4
struct RenderLane {
    hipStream_t stream = nullptr;
    DevBuf d_q[2], d_hits, d_cls, d_shadow, d_lbuf, d_counters, d_sum, d_spill_own;
    int *spill = nullptr;            // traversal-stack spill area of this lane's launches (lane 0: the scene's own)
    size_t ws_cap = 0, sum_pixels = 0;
    bool sum_valid = false;          // d_sum holds the sums of an earlier call (what RT_RENDER_ACCUMULATE adds to)
    std::vector<float> survival;     // fraction of a pass's paths still alive at each bounce, from the last pass that ran every bounce (k_finish)
};

struct rt_scene {
    rt_ctx *ctx = nullptr;
    DevScene dev;
    std::vector<rt_bvh_node> nodes; std::vector<int32_t> order;
    int n_prims = 0;
    bool classes[RT_NUM_CLASSES] = {};
    DevBuf d_nodes, d_slots, d_frames, d_uv, d_spheres, d_slot_mat, d_slot_light, d_mats, d_tex, d_lights, d_envs, d_light_cdf, d_spill, d_pairs, d_quads, d_flat, d_octs;
    int interior_depth = 0;
    std::vector<DevBuf *> d_texels;
    DevBuf d_trace_io, d_work, d_replay;      // d_replay: number of rays the order-free batch trace walked a second time, in the reference's order
    size_t spill_bytes = 0;
    // Rendering runs as up to RT_MAX_LANES independent wavefront pipelines ("lanes") at once, each on its own HIP stream with its own
    // queues, over its own share of the samples (rt_render_device).  Lane 0 runs on the context's stream and owns the result buffer.
    RenderLane lanes[RT_MAX_LANES];
    ~rt_scene() {
        for (DevBuf *b : d_texels) delete b;
        for (int k = 1; k < RT_MAX_LANES; ++k) if (lanes[k].stream) (void)hipStreamDestroy(lanes[k].stream);
    }
};

static rt_ctx *ctx_of(rt_ctx *c) { return c; }
static rt_ctx *ctx_of(rt_scene *s) { return s ? s->ctx : nullptr; }

static int env_int(const char *name, int def) { const char *v = getenv(name); return v && *v ? atoi(v) : def; }
static int tune_leaf_min() { return (env_int("RT_AMD_LEAF_MIN", RT_LEAF_MIN_LANES) & 255) | (env_int("RT_AMD_EXACT_SLABS", 0) ? 256 : 0); }
static int trace_blocks_per_cu() { int v = env_int("RT_AMD_TRACE_BLOCKS_PER_CU", 4); return v < 1 ? 1 : (v > 8 ? 8 : v); }
static int persistent_grid(rt_ctx *c) { return c->n_cu * 8; }   // batch trace kernels (grid-stride)

namespace {
// Host-side copies of the reference's float helpers used while precomputing per-triangle frames
// (triangle.cpp:121-146, reflection.h:139-141).  Built with -ffp-contract=off like the rest of the host code.
struct H3 { float x, y, z; };
inline H3 h3(float x, float y, float z) { H3 r = {x, y, z}; return r; }
inline H3 hsub(H3 a, H3 b) { return h3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline H3 hscale(H3 v, float s) { return h3(s * v.x, s * v.y, s * v.z); }
inline H3 hdivv(H3 v, float f) { float inv = 1.f / f; return h3(v.x * inv, v.y * inv, v.z * inv); }
inline float hlen2(H3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
inline H3 hnorm(H3 v) { return hdivv(v, std::sqrt(hlen2(v))); }
inline H3 hcross(H3 a, H3 b) {
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return h3((float)((ay * bz) - (az * by)), (float)((az * bx) - (ax * bz)), (float)((ax * by) - (ay * bx)));
}
inline void hcoordsys(H3 v1, H3 *v2, H3 *v3) {
    if (std::abs(v1.x) > std::abs(v1.y)) *v2 = hdivv(h3(-v1.z, 0, v1.x), std::sqrt(v1.x * v1.x + v1.z * v1.z));
    else *v2 = hdivv(h3(0, v1.z, -v1.y), std::sqrt(v1.y * v1.y + v1.z * v1.z));
    *v3 = hcross(v1, *v2);
}
// Returns false when Triangle::Intersect would reject every hit on this triangle.
bool triangle_frame(const float *P, const float *UV, H3 *n, H3 *ss, H3 *ts) {
    H3 p0 = h3(P[0], P[1], P[2]), p1 = h3(P[3], P[4], P[5]), p2 = h3(P[6], P[7], P[8]);
    float duv02x = UV[0] - UV[4], duv02y = UV[1] - UV[5], duv12x = UV[2] - UV[4], duv12y = UV[3] - UV[5];
    H3 dp02 = hsub(p0, p2), dp12 = hsub(p1, p2);
    float determinant = duv02x * duv12y - duv02y * duv12x;
    bool degenerateUV = std::abs(determinant) < 1e-8;
    H3 dpdu = h3(0, 0, 0), dpdv = h3(0, 0, 0);
    if (!degenerateUV) {
        float invdet = 1 / determinant;
        H3 a = hscale(dp02, duv12y), b = hscale(dp12, duv02y);
        dpdu = hscale(hsub(a, b), invdet);
        H3 c = hscale(dp02, -duv12x), e = hscale(dp12, duv02x);
        dpdv = hscale(h3(c.x + e.x, c.y + e.y, c.z + e.z), invdet);
    }
    bool ok = true;
    if (degenerateUV || hlen2(hcross(dpdu, dpdv)) == 0) {
        H3 ng = hcross(hsub(p2, p0), hsub(p1, p0));
        if (hlen2(ng) == 0) ok = false;
        else hcoordsys(hnorm(ng), &dpdu, &dpdv);
    }
    *n = hnorm(hcross(dp02, dp12));
    *ss = hnorm(dpdu);
    *ts = hcross(*n, *ss);
    return ok;
}
bool is_affine(const float *m) { return m[12] == 0.f && m[13] == 0.f && m[14] == 0.f && m[15] == 1.f; }
}  // namespace

// Everything rt_scene_create follows through the description: required arrays present, counts sane, every index in range.
// Returns the complaint, or an empty string.
static std::string check_desc(const rt_scene_desc *d) {
    if (d->n_prims <= 0) return "empty scene";
    if (!d->prim_kind || !d->prim_geom || !d->prim_material || !d->prim_light) return "a per-primitive array is NULL";
    if (d->n_tris < 0 || d->n_spheres < 0 || d->n_materials <= 0 || d->n_textures < 0 || d->n_lights < 0) return "negative or zero table size";
    if (d->n_tris > 0 && (!d->tri_p || !d->tri_uv)) return "tri_p / tri_uv is NULL";
    if (d->n_spheres > 0 && !d->spheres) return "spheres is NULL";
    if (!d->materials || (d->n_textures > 0 && !d->textures) || (d->n_lights > 0 && !d->lights)) return "materials / textures / lights is NULL";
    if (d->bvh_nodes && (!d->prim_order || d->n_nodes <= 0)) return "bvh_nodes given without prim_order";
    for (int i = 0; i < d->n_prims; ++i) {
        int g = d->prim_geom[i], k = d->prim_kind[i], m = d->prim_material[i], l = d->prim_light[i];
        if (k != RT_PRIM_TRIANGLE && k != RT_PRIM_SPHERE) return "unknown primitive kind";
        if (g < 0 || g >= (k == RT_PRIM_TRIANGLE ? d->n_tris : d->n_spheres)) return "primitive references out of range (geometry)";
        if (m < 0 || m >= d->n_materials) return "primitive references out of range (material)";
        if (l < -1 || l >= d->n_lights) return "primitive references out of range (light)";
    }
    for (int i = 0; i < d->n_materials; ++i) {
        const rt_material &m = d->materials[i];
        if (m.kind < 0 || m.kind >= RT_NUM_CLASSES) return "unknown material kind";
        int need = m.kind == RT_MAT_MATTE || m.kind == RT_MAT_MIRROR ? 1 : m.kind == RT_MAT_GLASS || m.kind == RT_MAT_PLASTIC ? 3 : (m.kind == RT_MAT_METAL || m.kind == RT_MAT_ROUGHGLASS) ? 4 : 1;
        for (int k = 0; k < 4; ++k)
            if (m.tex[k] >= d->n_textures || m.tex[k] < -1 || (k < need && m.tex[k] < 0)) return "material texture slot out of range";
    }
    if (d->n_env_lights < 0 || (d->n_env_lights > 0 && !d->env_lights)) return "env_lights is NULL";
    for (int i = 0; i < d->n_lights; ++i) {
        const rt_light &L = d->lights[i];
        if (L.kind == RT_LIGHT_INFINITE) {
            if (L.env < 0 || L.env >= d->n_env_lights) return "infinite light references out of range";
            continue;
        }
        if (L.kind != RT_LIGHT_AREA) return "unknown light kind";
        if (L.prim < 0 || L.prim >= d->n_prims || d->prim_kind[L.prim] != RT_PRIM_TRIANGLE) return "area lights must be triangles";
        if (d->prim_light[L.prim] != i) return "a light's primitive does not point back at the light";
    }
    for (int i = 0; i < d->n_env_lights; ++i) {
        const rt_env_light &E = d->env_lights[i];
        if (E.width <= 0 || E.height <= 0 || E.nu <= 0 || E.nv <= 0 || (size_t)E.width * E.height > ((size_t)1 << 28) || (size_t)E.nu * E.nv > ((size_t)1 << 28))
            return "infinite light with an empty or oversized map";
        if (!E.texels || !E.func || !E.cdf || !E.func_int || !E.marginal_cdf) return "infinite light with a NULL table";
    }
    return "";
}

// A tree the kernels can walk: links in range and forward (so every walk terminates), every node reached exactly once from
// the root, every leaf slot covered exactly once, prim_order a permutation, and no more interior levels than a traversal
// stack holds (a walk keeps at most one pending node per interior level; the reference's own stack is nodesToVisit[64],
// bvh.cpp:253).  *interior_levels receives the number of interior levels.
static std::string check_tree(const rt_bvh_node *nodes, int nn, const int32_t *order, int n_prims, int *interior_levels) {
    if (nn <= 0) return "empty tree";
    if (nn > 2 * n_prims - 1) return "more nodes than a binary tree over the primitives can have";
    for (int i = 0; i < nn; ++i) {
        const rt_bvh_node &nd = nodes[i];
        if (nd.nprim > 0) { if (nd.offset < 0 || nd.offset + nd.nprim > n_prims) return "leaf range out of bounds"; }
        else if (nd.offset <= i + 1 || nd.offset >= nn || i + 1 >= nn || nd.axis > 2) return "interior node links out of bounds";
    }
    std::vector<uint8_t> seen_node(nn, 0), seen_slot(n_prims, 0), seen_prim(n_prims, 0);
    std::vector<std::pair<int, int>> todo; todo.push_back({0, 1});
    int deepest = 0;
    while (!todo.empty()) {
        auto [i, level] = todo.back(); todo.pop_back();
        if (seen_node[i]++) return "a node is reachable twice";
        const rt_bvh_node &nd = nodes[i];
        if (nd.nprim > 0) { for (int k = 0; k < nd.nprim; ++k) if (seen_slot[nd.offset + k]++) return "two leaves share a slot"; }
        else { deepest = std::max(deepest, level); todo.push_back({nd.offset, level + 1}); todo.push_back({i + 1, level + 1}); }
    }
    for (int i = 0; i < nn; ++i) if (!seen_node[i]) return "a node is not reachable from the root";
    for (int i = 0; i < n_prims; ++i) {
        if (!seen_slot[i]) return "a leaf slot is covered by no leaf";
        int p = order[i];
        if (p < 0 || p >= n_prims || seen_prim[p]++) return "prim_order is not a permutation";
    }
    if (deepest > RT_MAX_STACK) return "tree deeper than the 64-entry traversal stack";
    if (interior_levels) *interior_levels = deepest;
    return "";
}

extern "C" {

// Host-only: what rt_scene_create checks before it follows any index.
int rt_scene_desc_check(const rt_scene_desc *d, char *msg, size_t msg_cap) {
    std::string why = d ? check_desc(d) : std::string("NULL description");
    if (why.empty() && d->bvh_nodes) why = check_tree(d->bvh_nodes, d->n_nodes, d->prim_order, d->n_prims, nullptr);
    if (msg && msg_cap) snprintf(msg, msg_cap, "%s", why.c_str());
    return why.empty() ? 0 : -1;
}

const char *rt_last_error(rt_ctx *ctx) { return ctx ? ctx->error.c_str() : g_init_error.c_str(); }

int rt_init(int device, rt_ctx **out) {
    if (!out) return -1;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { g_init_error = "no HIP device available (this library has no CPU path)"; return -3; }
    if (device < 0 || device >= n) { g_init_error = "device index out of range"; return -1; }
    if ((e = hipSetDevice(device)) != hipSuccess) { g_init_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return -2; }
    rt_ctx *c = new rt_ctx; c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if ((e = hipStreamCreate(&c->stream)) != hipSuccess) { g_init_error = std::string("hipStreamCreate: ") + hipGetErrorString(e); delete c; return -2; }
    // this device's address of the static scratch array (the device was made current above)
    void *sp = nullptr;
    if ((e = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch))) != hipSuccess || !sp) {
        g_init_error = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e); (void)hipStreamDestroy(c->stream); delete c; return -2;
    }
    c->scratch_ptr = (unsigned char *)sp;
    if ((e = hipEventCreate(&c->ev_a)) != hipSuccess || (e = hipEventCreate(&c->ev_b)) != hipSuccess) {
        g_init_error = std::string("hipEventCreate: ") + hipGetErrorString(e);
        if (c->ev_a) (void)hipEventDestroy(c->ev_a);
        (void)hipStreamDestroy(c->stream); delete c; return -2;
    }
    *out = c;
    return 0;
}
void rt_shutdown(rt_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->ev_a) (void)hipEventDestroy(ctx->ev_a);
    if (ctx->ev_b) (void)hipEventDestroy(ctx->ev_b);
    delete ctx;
}

int rt_scene_create(rt_ctx *this_, const rt_scene_desc *d, rt_scene **out) {
    if (!this_ || !d || !out) return -1;
    rt_ctx *ctx = this_;
    HIP_OK(hipSetDevice(ctx->device));
    // every index of the description is checked here, in original primitive order, before anything follows it
    // (rt_prim_bounds and the tree builder below read through prim_kind / prim_geom)
    { std::string why = check_desc(d); if (!why.empty()) { set_error(ctx, "rt_scene_create: " + why); return -1; } }
    std::unique_ptr<rt_scene> sc(new rt_scene);
    sc->ctx = ctx; sc->n_prims = d->n_prims;
    if (d->bvh_nodes) {
        sc->nodes.assign(d->bvh_nodes, d->bvh_nodes + d->n_nodes); sc->order.assign(d->prim_order, d->prim_order + d->n_prims);
    } else {
        std::vector<float> pb(6 * (size_t)d->n_prims);
        if (rt_prim_bounds(d, pb.data()) != 0) { set_error(ctx, "rt_scene_create: rt_prim_bounds failed"); return -1; }
        sc->nodes.resize(2 * (size_t)d->n_prims); sc->order.resize(d->n_prims);
        int32_t nn = 0;
        if (rt_bvh_build(d->n_prims, pb.data(), sc->nodes.data(), &nn, sc->order.data()) != 0) { set_error(ctx, "rt_bvh_build failed"); return -1; }
        sc->nodes.resize(nn);
    }
    int nn = (int)sc->nodes.size();
    { std::string why = check_tree(sc->nodes.data(), nn, sc->order.data(), d->n_prims, &sc->interior_depth); if (!why.empty()) { set_error(ctx, "rt_scene_create: " + why); return -1; } }
    for (int i = 0; i < d->n_materials; ++i) sc->classes[d->materials[i].kind] = true;
    // slots
    size_t ns = (size_t)d->n_prims;
    std::vector<float> slots(12 * ns), frames(12 * ns, 0.f), uv(6 * ns, 0.f);
    std::vector<int> smat(ns), slight(ns);
    std::vector<float> spheres(32 * (size_t)std::max(1, d->n_spheres), 0.f);
    for (int s = 0; s < d->n_spheres; ++s) {
        const rt_sphere &S = d->spheres[s];
        if (!is_affine(S.o2w) || !is_affine(S.w2o)) { set_error(ctx, "rt_scene_create: sphere transforms must be affine"); return -1; }
        float *r = &spheres[32 * (size_t)s];
        r[0] = S.radius; memcpy(r + 4, S.w2o, 48); memcpy(r + 16, S.o2w, 48);
    }
    for (size_t s = 0; s < ns; ++s) {
        int p = sc->order[s]; int g = d->prim_geom[p];
        float *r = &slots[12 * s]; uint32_t flags = 0; int32_t geom = 0;
        if (d->prim_kind[p] == RT_PRIM_TRIANGLE) {
            const float *P = d->tri_p + 9 * (size_t)g, *UV = d->tri_uv + 6 * (size_t)g;
            memcpy(r, P, 36); memcpy(&uv[6 * s], UV, 24);
            H3 n, ss, ts;
            if (!triangle_frame(P, UV, &n, &ss, &ts)) flags |= RT_SLOT_REJECT_CLOSEST;
            float *f = &frames[12 * s];
            f[0] = n.x; f[1] = n.y; f[2] = n.z; f[4] = ss.x; f[5] = ss.y; f[6] = ss.z; f[8] = ts.x; f[9] = ts.y; f[10] = ts.z;
        } else { flags |= RT_SLOT_SPHERE; geom = g; }
        memcpy(r + 9, &flags, 4); memcpy(r + 10, &p, 4); memcpy(r + 11, &geom, 4);
        smat[s] = d->prim_material[p]; slight[s] = d->prim_light[p];
    }
    std::vector<DevLight> lights((size_t)std::max(1, d->n_lights));
    for (int i = 0; i < d->n_lights; ++i) {
        const rt_light &L = d->lights[i]; DevLight &D = lights[i]; memset(&D, 0, sizeof(D));
        D.kind = L.kind; D.env = L.env;
        if (L.kind == RT_LIGHT_INFINITE) continue;
        if (L.prim < 0 || L.prim >= d->n_prims || d->prim_kind[L.prim] != RT_PRIM_TRIANGLE) { set_error(ctx, "rt_scene_create: area lights must be triangles"); return -1; }
        size_t g = (size_t)d->prim_geom[L.prim];
        const float *P = d->tri_p + 9 * g;
        memcpy(D.p0, P, 12); memcpy(D.p1, P + 3, 12); memcpy(D.p2, P + 6, 12);
        H3 p0 = h3(P[0], P[1], P[2]), p1 = h3(P[3], P[4], P[5]), p2 = h3(P[6], P[7], P[8]);
        H3 c = hcross(hsub(p1, p0), hsub(p2, p0));
        H3 n = hnorm(c);
        D.n[0] = n.x; D.n[1] = n.y; D.n[2] = n.z;
        D.area = (float)(0.5 * std::sqrt(hlen2(c)));
        if (!(D.area > 0.f)) { set_error(ctx, "rt_scene_create: degenerate light triangle"); return -1; }
        if (d->tri_n && d->tri_has_n && d->tri_has_n[g]) { D.has_n = 1; memcpy(D.vn, d->tri_n + 9 * g, 36); }
        else if (L.flip_normal) { D.n[0] = -n.x; D.n[1] = -n.y; D.n[2] = -n.z; }      // triangle.cpp:242-243
        D.Le[0] = L.Le[0]; D.Le[1] = L.Le[1]; D.Le[2] = L.Le[2]; D.two_sided = L.two_sided;
    }
    std::vector<DevTexture> tex((size_t)std::max(1, d->n_textures));
    for (int i = 0; i < d->n_textures; ++i) {
        const rt_texture &T = d->textures[i]; DevTexture &D = tex[i]; memset(&D, 0, sizeof(D));
        D.kind = T.kind; D.width = T.width; D.height = T.height; memcpy(D.value, T.value, 12); D.su = T.su; D.sv = T.sv; D.du = T.du; D.dv = T.dv;
        if (T.kind == RT_TEX_IMAGE) {
            if (T.width <= 0 || T.height <= 0 || !T.texels) { set_error(ctx, "rt_scene_create: image texture without texels"); return -1; }
            size_t bytes = (size_t)T.width * T.height * 3 * sizeof(float);
            DevBuf *b = new DevBuf; sc->d_texels.push_back(b);
            HIP_OK(b->alloc(bytes));
            HIP_OK(hipMemcpy(b->p, T.texels, bytes, hipMemcpyHostToDevice));
            D.texels = (const float *)b->p;
        } else if (T.kind != RT_TEX_CONST) { set_error(ctx, "rt_scene_create: unknown texture kind"); return -1; }
    }
    std::vector<DevEnv> envs((size_t)d->n_env_lights);
    for (int i = 0; i < d->n_env_lights; ++i) {
        const rt_env_light &E = d->env_lights[i]; DevEnv &D = envs[i]; memset(&D, 0, sizeof(D));
        if (!is_affine(E.light_to_world)) { set_error(ctx, "rt_scene_create: light transforms must be affine"); return -1; }
        if (!is_affine(E.world_to_light)) { set_error(ctx, "rt_scene_create: light transforms must be affine"); return -1; }
        memcpy(D.l2w, E.light_to_world, 48); memcpy(D.w2l, E.world_to_light, 48);
        D.width = E.width; D.height = E.height; D.nu = E.nu; D.nv = E.nv; D.marginal_int = E.marginal_int; D.world_radius = E.world_radius;
        auto upf = [&](const float *src, size_t n, const float **dst) -> hipError_t {
            DevBuf *b = new DevBuf; sc->d_texels.push_back(b);
            hipError_t e = b->alloc(n * sizeof(float)); if (e != hipSuccess) return e;
            *dst = (const float *)b->p;
            return hipMemcpy(b->p, src, n * sizeof(float), hipMemcpyHostToDevice);
        };
        HIP_OK(upf(E.texels, 3 * (size_t)E.width * E.height, &D.texels)); HIP_OK(upf(E.func, (size_t)E.nu * E.nv, &D.func));
        HIP_OK(upf(E.cdf, (size_t)E.nv * (E.nu + 1), &D.cdf)); HIP_OK(upf(E.func_int, (size_t)E.nv, &D.func_int));
        HIP_OK(upf(E.marginal_cdf, (size_t)E.nv + 1, &D.marginal_cdf));
    }
    std::vector<DevMaterial> mats((size_t)d->n_materials);
    for (int i = 0; i < d->n_materials; ++i) { mats[i].kind = d->materials[i].kind; memcpy(mats[i].tex, d->materials[i].tex, 16); mats[i].remap = d->materials[i].remap_roughness; }

    auto up = [&](DevBuf &b, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = b.alloc(bytes); if (e != hipSuccess) return e;
        return hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice);
    };
    HIP_OK(up(sc->d_nodes, sc->nodes.data(), sc->nodes.size() * sizeof(rt_bvh_node)));
    HIP_OK(up(sc->d_slots, slots.data(), slots.size() * 4));
    HIP_OK(up(sc->d_frames, frames.data(), frames.size() * 4));
    HIP_OK(up(sc->d_uv, uv.data(), uv.size() * 4));
    HIP_OK(up(sc->d_spheres, spheres.data(), spheres.size() * 4));
    HIP_OK(up(sc->d_slot_mat, smat.data(), smat.size() * 4));
    HIP_OK(up(sc->d_slot_light, slight.data(), slight.size() * 4));
    HIP_OK(up(sc->d_mats, mats.data(), mats.size() * sizeof(DevMaterial)));
    HIP_OK(up(sc->d_tex, tex.data(), tex.size() * sizeof(DevTexture)));
    HIP_OK(up(sc->d_lights, lights.data(), lights.size() * sizeof(DevLight)));
    if (!envs.empty()) HIP_OK(up(sc->d_envs, envs.data(), envs.size() * sizeof(DevEnv)));
    // Distribution1D over func = 1 per light, as UniformLightDistribution builds it (lightdistrib.cpp:12-15, sampling.h:16-27)
    std::vector<float> light_cdf((size_t)d->n_lights + 1, 0.f);
    float light_func_int = 0.f;
    if (d->n_lights > 0) {
        const int n = d->n_lights;
        for (int i = 1; i < n + 1; ++i) light_cdf[i] = light_cdf[i - 1] + 1.f / n;
        light_func_int = light_cdf[n];
        if (light_func_int == 0) { for (int i = 1; i < n + 1; ++i) light_cdf[i] = float(i) / float(n); }
        else for (int i = 1; i < n + 1; ++i) light_cdf[i] /= light_func_int;
    }
    HIP_OK(up(sc->d_light_cdf, light_cdf.data(), light_cdf.size() * sizeof(float)));
    // pair nodes: one 64-byte record per interior node with both children's boxes
    std::vector<int32_t> pairIndex(nn, -1);
    int n_pairs = 0;
    for (int i = 0; i < nn; ++i) if (sc->nodes[i].nprim == 0) pairIndex[i] = n_pairs++;
    auto childInfo = [&](int c) -> uint32_t {
        const rt_bvh_node &nd = sc->nodes[c];
        if (nd.nprim == 0) return (uint32_t)pairIndex[c];
        if (nd.nprim == 1) return RT_PAIR_LEAF | (uint32_t)nd.offset;
        if (nd.nprim <= 128 && nd.offset < (1 << 22)) return RT_PAIR_LEAF_MULTI | RT_PAIR_LEAF_INLINE | ((uint32_t)(nd.nprim - 1) << 22) | (uint32_t)nd.offset;
        return RT_PAIR_LEAF_MULTI | (uint32_t)c;
    };
    if (nn >= (1 << 29)) { set_error(ctx, "rt_scene_create: tree too large"); return -1; }
    std::vector<float> pairs(16 * (size_t)std::max(1, n_pairs), 0.f);
    for (int i = 0; i < nn; ++i) {
        if (sc->nodes[i].nprim != 0) continue;
        float *P = &pairs[16 * (size_t)pairIndex[i]];
        int l = i + 1, r = sc->nodes[i].offset;
        memcpy(P, sc->nodes[l].bmin, 12); memcpy(P + 3, sc->nodes[l].bmax, 12);
        memcpy(P + 6, sc->nodes[r].bmin, 12); memcpy(P + 9, sc->nodes[r].bmax, 12);
        uint32_t il = childInfo(l), ir = childInfo(r); int32_t axis = sc->nodes[i].axis;
        memcpy(P + 12, &il, 4); memcpy(P + 13, &ir, 4); memcpy(P + 14, &axis, 4);
    }
    HIP_OK(up(sc->d_pairs, pairs.data(), pairs.size() * 4));
    // quad nodes: one 128-byte record per even-depth interior node with the boxes of its four grandchildren.  Only for trees
    // whose interior boxes are bit for bit the min/max of their children's (true of every tree rt_bvh_build makes).
    std::vector<float> quads;
    int n_quads = 0, tree_depth = 0;
    bool quads_ok = sc->nodes[0].nprim == 0;
    if (quads_ok) {
        std::vector<int32_t> quadIndex(nn, -1);
        struct Visit { int node, depth; };
        std::vector<Visit> todo; todo.push_back({0, 0});
        std::vector<int> quadNodes;
        while (!todo.empty()) {           // pre-order over even-depth interior nodes, near subtrees contiguous
            Visit v = todo.back(); todo.pop_back();
            tree_depth = std::max(tree_depth, v.depth);
            quadIndex[v.node] = n_quads++; quadNodes.push_back(v.node);
            int kids[2] = {v.node + 1, sc->nodes[v.node].offset};
            int grand[4]; int ng = 0;
            for (int k = 0; k < 2; ++k)
                if (sc->nodes[kids[k]].nprim == 0) { grand[ng++] = kids[k] + 1; grand[ng++] = sc->nodes[kids[k]].offset; tree_depth = std::max(tree_depth, v.depth + 1); }
            for (int g = ng - 1; g >= 0; --g)
                if (sc->nodes[grand[g]].nprim == 0) todo.push_back({grand[g], v.depth + 2});
        }
        auto quadInfo = [&](int c) -> uint32_t {
            const rt_bvh_node &nd = sc->nodes[c];
            if (nd.nprim == 0) return (uint32_t)quadIndex[c];
            if (nd.nprim == 1) return RT_PAIR_LEAF | (uint32_t)nd.offset;
            if (nd.nprim <= 128 && nd.offset < (1 << 22)) return RT_PAIR_LEAF_MULTI | RT_PAIR_LEAF_INLINE | ((uint32_t)(nd.nprim - 1) << 22) | (uint32_t)nd.offset;
            return RT_PAIR_LEAF_MULTI | (uint32_t)c;
        };
        quads.assign(32 * (size_t)n_quads, 0.f);
        for (int qi = 0; qi < n_quads && quads_ok; ++qi) {
            int p = quadNodes[qi];
            float *Q = &quads[32 * (size_t)qi];
            uint32_t meta = (uint32_t)sc->nodes[p].axis & 3u, info[4] = {RT_QUAD_NONE, RT_QUAD_NONE, RT_QUAD_NONE, RT_QUAD_NONE};
            int kids[2] = {p + 1, sc->nodes[p].offset};
            for (int k = 0; k < 2; ++k) {
                const rt_bvh_node &c = sc->nodes[kids[k]];
                float *B = Q + 12 * k;
                if (c.nprim != 0) {
                    memcpy(B, c.bmin, 12); memcpy(B + 3, c.bmax, 12); memcpy(B + 6, c.bmin, 12); memcpy(B + 9, c.bmax, 12);
                    info[2 * k] = quadInfo(kids[k]);
                    meta |= 64u << k;
                } else {
                    const rt_bvh_node &a = sc->nodes[kids[k] + 1], &b = sc->nodes[c.offset];
                    memcpy(B, a.bmin, 12); memcpy(B + 3, a.bmax, 12); memcpy(B + 6, b.bmin, 12); memcpy(B + 9, b.bmax, 12);
                    info[2 * k] = quadInfo(kids[k] + 1); info[2 * k + 1] = quadInfo(c.offset);
                    meta |= ((uint32_t)c.axis & 3u) << (2 + 2 * k);
                    float u[6];
                    for (int j = 0; j < 3; ++j) { u[j] = b.bmin[j] < a.bmin[j] ? b.bmin[j] : a.bmin[j]; u[3 + j] = a.bmax[j] < b.bmax[j] ? b.bmax[j] : a.bmax[j]; }
                    if (memcmp(u, c.bmin, 12) != 0 || memcmp(u + 3, c.bmax, 12) != 0) quads_ok = false;
                }
            }
            memcpy(Q + 24, info, 16); memcpy(Q + 28, &meta, 4);
        }
    }
    // Is every interior box bit for bit the union of its children's (std::min / std::max argument order, geometry.h:833-840)?  True of
    // every tree rt_bvh_build makes; a caller's tree may differ.  The root alone decides whether a walk may skip the root's own slab
    // test; the whole tree decides whether "a leaf is reached exactly when its own box is hit" holds (k_extend_flat).
    auto is_union = [&](int i) {
        const rt_bvh_node &c = sc->nodes[i], &a = sc->nodes[i + 1], &b = sc->nodes[c.offset];
        float u[6];
        for (int j = 0; j < 3; ++j) { u[j] = b.bmin[j] < a.bmin[j] ? b.bmin[j] : a.bmin[j]; u[3 + j] = a.bmax[j] < b.bmax[j] ? b.bmax[j] : a.bmax[j]; }
        return memcmp(u, c.bmin, 12) == 0 && memcmp(u + 3, c.bmax, 12) == 0;
    };
    bool nest_ok = true;
    int n_leaves = 0;
    for (int i = 0; i < nn; ++i) { if (sc->nodes[i].nprim != 0) n_leaves++; else if (!is_union(i)) nest_ok = false; }
    const bool root_implied = sc->nodes[0].nprim == 0 && is_union(0);
    // flat leaf table (k_extend_flat): the leaves' own boxes and info words, depth-first
    std::vector<float> flat;
    if (nest_ok && n_leaves <= RT_FLAT_MAX && !env_int("RT_AMD_NO_FLAT_TABLE", 0)) {
        for (int i = 0; i < nn; ++i) {
            const rt_bvh_node &nd = sc->nodes[i];
            if (nd.nprim == 0) continue;
            float rec[8] = {nd.bmin[0], nd.bmin[1], nd.bmin[2], nd.bmax[0], nd.bmax[1], nd.bmax[2], 0.f, 0.f};
            uint32_t info = childInfo(i);
            memcpy(rec + 6, &info, 4);
            flat.insert(flat.end(), rec, rec + 8);
        }
        HIP_OK(up(sc->d_flat, flat.data(), flat.size() * 4));
    }
    // Eight-wide records with quantised boxes (rt_oct.h) for the order-free batch trace of large trees: one 128-byte record per interior node
    // at a depth that is a multiple of three holds its (at most eight) descendants three levels down — or the leaves met on the way — as
    // 8-bit boxes relative to the node's own box, each containing the descendant's exact box.  Triangle scenes with exactly nested boxes only
    // (a leaf's exact box is rebuilt from its triangles' vertices when the leaf is tested).
    std::vector<uint32_t> octs;
    int n_octs = 0, oct_levels = 0;
    if (nest_ok && sc->nodes[0].nprim == 0 && d->n_spheres == 0 && nn > 4096 && env_int("RT_AMD_OCT", 1)) {
        std::vector<int32_t> octIndex(nn, -1);
        struct Visit { int node, level; };
        std::vector<Visit> todo; todo.push_back({0, 1});
        std::vector<int> octNodes;
        auto entries_of = [&](int r, int *ent) {            // the up-to-eight descendants of interior node r, three levels down
            int cur[8], ncur = 0, nxt[8], nnxt;
            cur[ncur++] = r + 1; cur[ncur++] = sc->nodes[r].offset;
            for (int lvl = 0; lvl < 2; ++lvl) {
                nnxt = 0;
                for (int k = 0; k < ncur; ++k) {
                    const rt_bvh_node &c = sc->nodes[cur[k]];
                    if (c.nprim != 0) nxt[nnxt++] = cur[k];
                    else { nxt[nnxt++] = cur[k] + 1; nxt[nnxt++] = c.offset; }
                }
                ncur = nnxt; memcpy(cur, nxt, sizeof(int) * ncur);
            }
            memcpy(ent, cur, sizeof(int) * ncur);
            return ncur;
        };
        while (!todo.empty()) {
            Visit v = todo.back(); todo.pop_back();
            oct_levels = std::max(oct_levels, v.level);
            octIndex[v.node] = n_octs++; octNodes.push_back(v.node);
            int ent[8]; int ne = entries_of(v.node, ent);
            for (int k = ne - 1; k >= 0; --k) if (sc->nodes[ent[k]].nprim == 0) todo.push_back({ent[k], v.level + 1});
        }
        octs.assign(32 * (size_t)n_octs, 0u);
        bool oct_ok = true;
        for (int oi = 0; oi < n_octs && oct_ok; ++oi) {
            const int r = octNodes[oi];
            const rt_bvh_node &R = sc->nodes[r];
            uint32_t *W = &octs[32 * (size_t)oi];
            int ent[8]; const int ne = entries_of(r, ent);
            int ex[3];
            for (int a = 0; a < 3; ++a) {
                // smallest power-of-two step with which 255 steps from the node's lower bound reach its upper bound
                double extent = (double)R.bmax[a] - (double)R.bmin[a];
                int e = -100;
                while (e < 100 && 255.0 * std::ldexp(1.0, e) < extent) ++e;
                ex[a] = e;
                memcpy(&W[a], &R.bmin[a], 4);
            }
            for (int k = 0; k < ne; ++k) {
                const rt_bvh_node &c = sc->nodes[ent[k]];
                for (int a = 0; a < 3; ++a) {
                    // quantised planes, checked in exact (double) arithmetic: origin + q * step <= lo, >= hi
                    const double o = R.bmin[a], step = std::ldexp(1.0, ex[a]);
                    int qlo = (int)std::floor(((double)c.bmin[a] - o) / step), qhi = (int)std::ceil(((double)c.bmax[a] - o) / step);
                    qlo = std::max(0, std::min(255, qlo)); qhi = std::max(0, std::min(255, qhi));
                    while (qlo > 0 && o + qlo * step > (double)c.bmin[a]) --qlo;
                    while (qhi < 255 && o + qhi * step < (double)c.bmax[a]) ++qhi;
                    if (o + qlo * step > (double)c.bmin[a] || o + qhi * step < (double)c.bmax[a]) oct_ok = false;
                    uint8_t *B = (uint8_t *)&W[4 + 4 * a];
                    B[2 * k] = (uint8_t)qlo; B[2 * k + 1] = (uint8_t)qhi;
                }
                W[16 + k] = c.nprim == 0 ? (uint32_t)octIndex[ent[k]] : childInfo(ent[k]);
            }
            W[3] = (uint32_t)(ex[0] + 128) | ((uint32_t)(ex[1] + 128) << 8) | ((uint32_t)(ex[2] + 128) << 16) | ((uint32_t)ne << 24);
        }
        if (!oct_ok) { octs.clear(); n_octs = 0; }
        else HIP_OK(up(sc->d_octs, octs.data(), octs.size() * 4));
    }
    if (env_int("RT_AMD_NO_QUADS", 0)) quads_ok = false;
    int quad_cap = 3 * (tree_depth / 2 + 1) + 2;      // at most three pushes per record on the path from the root
    const int oct_cap = n_octs ? 7 * (oct_levels + 1) + 2 : 0;      // ... and seven with eight-wide records
    if (quads_ok) HIP_OK(up(sc->d_quads, quads.data(), quads.size() * 4));
    size_t spill_ints = (size_t)persistent_grid(ctx) * RT_BLOCK * (size_t)std::max(std::max(RT_MAX_STACK, quads_ok ? quad_cap : 0), oct_cap) * 2;
    HIP_OK(sc->d_spill.alloc(spill_ints * 4));
    sc->spill_bytes = spill_ints * 4;
    DevScene &D = sc->dev;
    D.nodes = (const float4 *)sc->d_nodes.p; D.slots = (const float4 *)sc->d_slots.p; D.frames = (const float4 *)sc->d_frames.p;
    D.slot_uv = (const float *)sc->d_uv.p; D.spheres = (const float *)sc->d_spheres.p; D.slot_material = (const int *)sc->d_slot_mat.p;
    D.slot_light = (const int *)sc->d_slot_light.p; D.materials = (const DevMaterial *)sc->d_mats.p; D.textures = (const DevTexture *)sc->d_tex.p;
    D.lights = (const DevLight *)sc->d_lights.p; D.n_nodes = nn; D.n_slots = d->n_prims; D.n_lights = d->n_lights; D.pad = 0; D.n_materials = d->n_materials; D.n_textures = d->n_textures;
    D.pairs = (const float4 *)sc->d_pairs.p; D.n_pairs = n_pairs; D.envs = envs.empty() ? nullptr : (const DevEnv *)sc->d_envs.p;
    D.n_envs = (int)envs.size(); D.light_cdf = (const float *)sc->d_light_cdf.p; D.light_func_int = light_func_int;
    memcpy(D.root_box, sc->nodes[0].bmin, 12); memcpy(D.root_box + 3, sc->nodes[0].bmax, 12);
    D.root_info = childInfo(0);
    D.quads = quads_ok ? (const float4 *)sc->d_quads.p : nullptr; D.n_quads = quads_ok ? n_quads : 0; D.quad_stack_cap = quad_cap;
    D.octs = n_octs ? (const float4 *)sc->d_octs.p : nullptr; D.n_octs = n_octs; D.oct_stack_cap = oct_cap;
    D.root_implied = root_implied ? 1 : 0;
    D.flat = flat.empty() ? nullptr : (const float4 *)sc->d_flat.p; D.n_flat = (int)(flat.size() / 8);
    *out = sc.release();
    return 0;
}

void rt_scene_destroy(rt_scene *scene) {
    if (!scene) return;
    (void)hipSetDevice(scene->ctx->device);
    (void)hipStreamSynchronize(scene->ctx->stream);
    for (int k = 1; k < RT_MAX_LANES; ++k) if (scene->lanes[k].stream) (void)hipStreamSynchronize(scene->lanes[k].stream);
    delete scene;
}

int rt_scene_get_bvh(rt_scene *this_, rt_bvh_node *nodes_out, int32_t *n_nodes_out, int32_t *order_out) {
    if (!this_ || !n_nodes_out) return -1;
    *n_nodes_out = (int32_t)this_->nodes.size();
    if (nodes_out) memcpy(nodes_out, this_->nodes.data(), this_->nodes.size() * sizeof(rt_bvh_node));
    if (order_out) memcpy(order_out, this_->order.data(), this_->order.size() * sizeof(int32_t));
    return 0;
}

// Closest hits of the batch trace through the order-free walk (with the ordered replay of what it flags)?  By default where the scene
// has eight-wide records (large triangle scenes, rt_oct.h): on the 10 M-triangle soup three tree levels per 128-byte line instead of two
// bring the launch from 28.4 ms to 21.7 ms (105 of 16.7 M rays replayed).  Over the quad records the nearest-first order alone visits no
// fewer records than the reference's order by split axis and ray sign (29.8 ms), so scenes without eight-wide records keep the ordered
// walk unless RT_AMD_FREE_WALK=1 asks for the order-free one (kept as a cross-check of the primitives it shares with the frame kernels).
// RT_AMD_FREE_WALK=0: the reference's order throughout; RT_AMD_OCT_WALK=0: order-free over quad records.
static bool free_walk_enabled(rt_scene *s) { return s->dev.quads != nullptr && env_int("RT_AMD_FREE_WALK", s->dev.octs ? 1 : 0) != 0 && !env_int("RT_AMD_NODE_WALK", 0); }

// Batch trace launch: quad records when the scene has them, pair records otherwise and for the counting variant (whose
// counts are the reference's per-node and per-primitive test counts).
extern "C++" {
template <bool ANY, bool COUNT>
static void launch_trace(rt_scene *this_, int grid, size_t n, const float *rays, int aos, int *prim, float *t, float *b1, float *b2, uint8_t *any_out,
                         unsigned long long *work, int *spill = nullptr, bool allow_node_walk = true, uint32_t *replay = nullptr) {
    rt_ctx *ctx = this_->ctx;
    if (!spill) spill = (int *)this_->d_spill.p;      // the scene's own spill area, sized for the largest grid (rt_scene_create)
    if (allow_node_walk && env_int("RT_AMD_NODE_WALK", 0))      // knob: the one-node-per-step walk the frame kernels use
        hipLaunchKernelGGL((k_trace_batch<ANY, COUNT>), dim3(grid), dim3(RT_BLOCK), 0, ctx->stream, this_->dev, n, rays, aos, prim, t, b1, b2, any_out,
                           spill, work);
    else if (!ANY && !COUNT && this_->dev.quads && replay) {
        // closest hits without the reference's visit order; the rays whose answer that order could change are walked again in it inside the launch
        (void)hipMemsetAsync(replay, 0, 16, ctx->stream);      // [0] rays walked a second time, [2..3] the waves' ray counter
        if (this_->dev.octs && env_int("RT_AMD_OCT_WALK", 1))
            hipLaunchKernelGGL((k_trace_pairs<false, false, true, true, true>), dim3(grid), dim3(RT_BLOCK), 0, ctx->stream, this_->dev, n, rays, aos, prim, t, b1, b2, any_out,
                               spill, work, tune_leaf_min(), replay);
        else
        hipLaunchKernelGGL((k_trace_pairs<false, false, true, true>), dim3(grid), dim3(RT_BLOCK), 0, ctx->stream, this_->dev, n, rays, aos, prim, t, b1, b2, any_out,
                           spill, work, tune_leaf_min(), replay);
    } else if (!COUNT && this_->dev.quads)
        hipLaunchKernelGGL((k_trace_pairs<ANY, false, true>), dim3(grid), dim3(RT_BLOCK), 0, ctx->stream, this_->dev, n, rays, aos, prim, t, b1, b2, any_out,
                           spill, work, tune_leaf_min());
    else
        hipLaunchKernelGGL((k_trace_pairs<ANY, COUNT, false>), dim3(grid), dim3(RT_BLOCK), 0, ctx->stream, this_->dev, n, rays, aos, prim, t, b1, b2, any_out,
                           spill, work, tune_leaf_min());
}
}

static int trace_host(rt_scene *this_, bool any, size_t n, const float *o, const float *d, const float *tmax, int32_t *out_prim, float *out_t, float *out_b,
                      uint8_t *out_hit) {
    rt_ctx *ctx = this_->ctx;
    if (n == 0) return 0;
    HIP_OK(hipSetDevice(ctx->device));
    std::vector<float> rays(7 * n);
    for (size_t i = 0; i < n; ++i) { memcpy(&rays[7 * i], o + 3 * i, 12); memcpy(&rays[7 * i + 3], d + 3 * i, 12); rays[7 * i + 6] = tmax[i]; }
    size_t in_bytes = 7 * n * 4, out_bytes = any ? n : 16 * n;
    HIP_OK(this_->d_trace_io.ensure(in_bytes + out_bytes + 256));
    char *base = (char *)this_->d_trace_io.p;
    HIP_OK(hipMemcpyAsync(base, rays.data(), in_bytes, hipMemcpyHostToDevice, ctx->stream));
    char *outp = base + ((in_bytes + 255) & ~(size_t)255);
    int grid = (int)std::min<size_t>((n + RT_BLOCK - 1) / RT_BLOCK, (size_t)ctx->n_cu * trace_blocks_per_cu());
    uint32_t *replay = nullptr;
    if (!any && free_walk_enabled(this_)) { HIP_OK(this_->d_replay.ensure(16)); replay = (uint32_t *)this_->d_replay.p; }
    if (any)
        launch_trace<true, false>(this_, grid, n, (const float *)base, 1, nullptr, nullptr, nullptr, nullptr, (uint8_t *)outp, nullptr);
    else
        launch_trace<false, false>(this_, grid, n, (const float *)base, 1, (int *)outp, (float *)outp + n, (float *)outp + 2 * n, (float *)outp + 3 * n, nullptr,
                                   nullptr, nullptr, true, replay);
    HIP_OK(hipGetLastError());
    if (any) HIP_OK(hipMemcpyAsync(out_hit, outp, n, hipMemcpyDeviceToHost, ctx->stream));
    else {
        HIP_OK(hipMemcpyAsync(out_prim, outp, 4 * n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_OK(hipMemcpyAsync(out_t, outp + 4 * n, 4 * n, hipMemcpyDeviceToHost, ctx->stream));
    }
    std::vector<float> b12;
    if (!any && out_b) { b12.resize(2 * n); HIP_OK(hipMemcpyAsync(b12.data(), outp + 8 * n, 8 * n, hipMemcpyDeviceToHost, ctx->stream)); }
    HIP_OK(hipStreamSynchronize(ctx->stream));
    if (!any && out_b) for (size_t i = 0; i < n; ++i) { out_b[2 * i] = b12[i]; out_b[2 * i + 1] = b12[n + i]; }
    return 0;
}

int rt_trace_closest(rt_scene *scene, size_t n, const float *o, const float *d, const float *tmax, int32_t *out_prim, float *out_t, float *out_b) {
    if (!scene || (n && (!o || !d || !tmax || !out_prim || !out_t))) return -1;
    return trace_host(scene, false, n, o, d, tmax, out_prim, out_t, out_b, nullptr);
}
int rt_trace_any(rt_scene *scene, size_t n, const float *o, const float *d, const float *tmax, uint8_t *out_hit) {
    if (!scene || (n && (!o || !d || !tmax || !out_hit))) return -1;
    return trace_host(scene, true, n, o, d, tmax, nullptr, nullptr, nullptr, out_hit);
}

int rt_trace_closest_device(rt_scene *this_, size_t n, const void *rays, void *hits, int repeats, int count_work, rt_trace_stats *stats) {
    if (!this_ || !rays || !hits || n == 0) return -1;
    rt_ctx *ctx = this_->ctx;
    HIP_OK(hipSetDevice(ctx->device));
    if (repeats < 1) repeats = 1;
    int grid = (int)std::min<size_t>((n + RT_BLOCK - 1) / RT_BLOCK, (size_t)ctx->n_cu * trace_blocks_per_cu());
    int *prim = (int *)hits; float *t = (float *)hits + n, *b1 = (float *)hits + 2 * n, *b2 = (float *)hits + 3 * n;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (count_work) {
        HIP_OK(this_->d_work.ensure(16));
        HIP_OK(hipMemsetAsync(this_->d_work.p, 0, 16, ctx->stream));
        launch_trace<false, true>(this_, grid, n, (const float *)rays, 0, prim, t, b1, b2, nullptr, (unsigned long long *)this_->d_work.p);
        HIP_OK(hipGetLastError());
        unsigned long long w[2];
        HIP_OK(hipMemcpyAsync(w, this_->d_work.p, 16, hipMemcpyDeviceToHost, ctx->stream));
        HIP_OK(hipStreamSynchronize(ctx->stream));
        if (stats) { stats->nodes_visited = w[0]; stats->prims_tested = w[1]; }
    }
    uint32_t *replay = nullptr;
    if (free_walk_enabled(this_)) { HIP_OK(this_->d_replay.ensure(16)); replay = (uint32_t *)this_->d_replay.p; }
    EventPair ev;
    HIP_OK(ev.create());
    HIP_OK(hipEventRecord(ev.a, ctx->stream));
    for (int r = 0; r < repeats; ++r)
        launch_trace<false, false>(this_, grid, n, (const float *)rays, 0, prim, t, b1, b2, nullptr, nullptr, nullptr, true, replay);
    HIP_OK(hipEventRecord(ev.b, ctx->stream));
    HIP_OK(hipEventSynchronize(ev.b));
    HIP_OK(hipGetLastError());
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, ev.a, ev.b));
    if (stats) stats->seconds = (double)ms * 1e-3 / repeats;
    if (stats && replay) { uint32_t nr = 0; HIP_OK(hipMemcpy(&nr, replay, 4, hipMemcpyDeviceToHost)); stats->rays_replayed = nr; }
    return 0;
}

// rt_trace_closest_device for callers that cannot allocate: the spill area of the traversal stacks and the work counters live
// in this device's copy of the static `scratch` array instead of buffers the library allocates.  The grid is cut down until its
// lanes' stacks fit the spill area (32 MiB): 256 blocks while a lane needs at most 64 entries, 162 for the deepest tree rt_scene_create accepts.
int rt_trace_closest_device_noalloc(rt_scene *this_, size_t n, const void *rays, void *hits, int count_work, rt_trace_stats *stats) {
    if (!this_ || !rays || !hits || n == 0) return -1;
    rt_ctx *ctx = this_->ctx;
    if (n > RT_NOALLOC_MAX_RAYS) { set_error(ctx, "rt_trace_closest_device_noalloc: more than 2^28 rays in one call"); return -1; }
    if (count_work && !stats) { set_error(ctx, "rt_trace_closest_device_noalloc: count_work needs a stats record to return the counts in"); return -1; }
    if (!ctx->scratch_ptr) { set_error(ctx, "rt_trace_closest_device_noalloc: the context has no scratch address"); return -1; }
    HIP_OK(hipSetDevice(ctx->device));
    // stack entries per lane the kernels may spill: the counting kernel walks pair records (RT_MAX_STACK), the other one quad records
    const int cap = std::max(RT_MAX_STACK, this_->dev.quads ? this_->dev.quad_stack_cap : 0);
    const size_t per_block = (size_t)RT_BLOCK * (size_t)cap * 2 * sizeof(int);
    const size_t fit = RT_NOALLOC_SPILL_BYTES / per_block;
    if (fit < 1) { set_error(ctx, "rt_trace_closest_device_noalloc: the tree's traversal stack does not fit the static scratch"); return -1; }
    int grid = (int)std::min<size_t>(std::min<size_t>((n + RT_BLOCK - 1) / RT_BLOCK, (size_t)ctx->n_cu * trace_blocks_per_cu()), fit);
    int *prim = (int *)hits; float *t = (float *)hits + n, *b1 = (float *)hits + 2 * n, *b2 = (float *)hits + 3 * n;
    unsigned long long *work = (unsigned long long *)ctx->scratch_ptr;
    int *spill = (int *)(ctx->scratch_ptr + 256);
    if (stats) memset(stats, 0, sizeof(*stats));
    if (count_work) {
        HIP_OK(hipMemsetAsync(work, 0, 16, ctx->stream));        // the array keeps the last call's counts
        launch_trace<false, true>(this_, grid, n, (const float *)rays, 0, prim, t, b1, b2, nullptr, work, spill, false);
        HIP_OK(hipGetLastError());
        unsigned long long w[2];
        HIP_OK(hipMemcpyAsync(w, work, 16, hipMemcpyDeviceToHost, ctx->stream));
        HIP_OK(hipStreamSynchronize(ctx->stream));
        stats->nodes_visited = w[0]; stats->prims_tested = w[1];
    }
    if (stats) HIP_OK(hipEventRecord(ctx->ev_a, ctx->stream));
    launch_trace<false, false>(this_, grid, n, (const float *)rays, 0, prim, t, b1, b2, nullptr, nullptr, spill, false);
    HIP_OK(hipGetLastError());
    if (stats) {
        HIP_OK(hipEventRecord(ctx->ev_b, ctx->stream));
        HIP_OK(hipEventSynchronize(ctx->ev_b));
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, ctx->ev_a, ctx->ev_b));
        stats->seconds = (double)ms * 1e-3;
    } else HIP_OK(hipStreamSynchronize(ctx->stream));
    return 0;
}

static void bind_queue(PathQueue &q, void *base, size_t cap) {
    float *f = (float *)base;
    q.ox = f; q.oy = f + cap; q.oz = f + 2 * cap; q.dx = f + 3 * cap; q.dy = f + 4 * cap; q.dz = f + 5 * cap;
    q.tr = f + 6 * cap; q.tg = f + 7 * cap; q.tb = f + 8 * cap; q.id = (uint32_t *)(f + 9 * cap); q.state = (uint32_t *)(f + 10 * cap);
    q.es = f + 11 * cap;
}

extern "C++" {
// Dynamic LDS of k_shade<.., TABLES = 1>: the scene's shading tables; small_tables_lds_bytes: TABLES = 2.
static size_t shade_lds_bytes(const DevScene &d) {
    return (size_t)d.n_slots * (48 + 48 + 24 + 4) + (size_t)d.n_textures * sizeof(DevTexture) + (size_t)d.n_lights * sizeof(DevLight) +
           (size_t)d.n_materials * sizeof(DevMaterial);
}
static size_t small_tables_lds_bytes(const DevScene &d) {
    return (size_t)d.n_textures * sizeof(DevTexture) + (size_t)d.n_lights * sizeof(DevLight) + (size_t)d.n_materials * sizeof(DevMaterial);
}
static bool small_tables_in_lds(const DevScene &d) {
    return d.n_materials <= RT_LDS_TAB_MATERIALS && d.n_textures <= RT_LDS_TAB_TEXTURES && d.n_lights <= RT_LDS_TAB_LIGHTS && !env_int("RT_AMD_SHADE_GLOBAL_TABLES", 0);
}
static bool shade_tables_in_lds(const DevScene &d) {
    return d.n_slots <= RT_LDS_TAB_SLOTS && d.n_materials <= RT_LDS_TAB_MATERIALS && d.n_textures <= RT_LDS_TAB_TEXTURES && d.n_lights <= RT_LDS_TAB_LIGHTS &&
           !env_int("RT_AMD_SHADE_GLOBAL_TABLES", 0);
}
template <int MAT>
static void launch_shade(rt_scene *s, RenderLane &ln, const RenderParams &rp, int bounce, const PathQueue &q, const PathQueue &qn, const HitBuf &hb, uint32_t cap,
                         uint32_t segcap, const ShadowQueue &sq, int nseg, bool fuse_shadow, float4 *lbuf, BounceStats *stats) {
    if (rp.li || rp.nee_mode) {   // the estimators next to Li_re: one general variant (tables in global memory, either kind of light)
        hipLaunchKernelGGL((k_shade<MAT, 0, false, true, true>), dim3(nseg), dim3(RT_BLOCK), 0, ln.stream, s->dev, rp, bounce, q, qn, hb, (const uint32_t *)ln.d_cls.p,
                           cap, segcap, sq, (BounceCounts *)ln.d_counters.p, lbuf, stats);
        return;
    }
    if (s->dev.envs) {        // scenes with an infinite area light: tables in global memory, shadow rays through the shadow queue
        hipLaunchKernelGGL((k_shade<MAT, 0, false, true>), dim3(nseg), dim3(RT_BLOCK), 0, ln.stream, s->dev, rp, bounce, q, qn, hb, (const uint32_t *)ln.d_cls.p,
                           cap, segcap, sq, (BounceCounts *)ln.d_counters.p, lbuf, stats);
        return;
    }
    const bool lds_tables = shade_tables_in_lds(s->dev);
    const size_t dyn = lds_tables ? shade_lds_bytes(s->dev) : 0;
    if (lds_tables && fuse_shadow)
        hipLaunchKernelGGL((k_shade<MAT, 1, true>), dim3(nseg), dim3(RT_BLOCK), dyn, ln.stream, s->dev, rp, bounce, q, qn, hb, (const uint32_t *)ln.d_cls.p,
                           cap, segcap, sq, (BounceCounts *)ln.d_counters.p, lbuf, stats);
    else if (lds_tables)
        hipLaunchKernelGGL((k_shade<MAT, 1, false>), dim3(nseg), dim3(RT_BLOCK), dyn, ln.stream, s->dev, rp, bounce, q, qn, hb, (const uint32_t *)ln.d_cls.p,
                           cap, segcap, sq, (BounceCounts *)ln.d_counters.p, lbuf, stats);
    else if (small_tables_in_lds(s->dev))
        hipLaunchKernelGGL((k_shade<MAT, 2, false>), dim3(nseg), dim3(RT_BLOCK), small_tables_lds_bytes(s->dev), ln.stream, s->dev, rp, bounce, q, qn, hb,
                           (const uint32_t *)ln.d_cls.p, cap, segcap, sq, (BounceCounts *)ln.d_counters.p, lbuf, stats);
    else
        hipLaunchKernelGGL((k_shade<MAT, 0, false>), dim3(nseg), dim3(RT_BLOCK), 0, ln.stream, s->dev, rp, bounce, q, qn, hb, (const uint32_t *)ln.d_cls.p,
                           cap, segcap, sq, (BounceCounts *)ln.d_counters.p, lbuf, stats);
}
}  // extern "C++"

}  // extern "C"
__global__ __launch_bounds__(RT_BLOCK) void k_add_frames(float *__restrict__ sum, const float *__restrict__ other, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) sum[i] += other[i];
}
extern "C" {
struct LanePlan { int spb = 1, nseg = RT_BLOCK; size_t segcap = 0; };
// Pass size, queue segments and workspace of one lane (allocates; called before the render is timed).
static int lane_prepare(rt_scene *this_, RenderLane &ln, const rt_camera *cam, const rt_render_opts *o, size_t pool, bool pool_is_callers, LanePlan &plan) {
    rt_ctx *ctx = this_->ctx;
    HIP_OK(hipSetDevice(ctx->device));
    int W = cam->width, H = cam->height;
    size_t npix = (size_t)W * H;
    int S = o->sample_end - o->sample_begin;
    int spb = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(S, 1), pool / npix));
    // queue segments = grid of every frame kernel: five blocks per CU where every kernel of the frame allows it
    const int ncu = std::max(1, ctx->n_cu);
    const bool small_tree = this_->dev.n_nodes <= 4096;
    // (the metal and plastic shading kernels need more than the 96 registers that five waves per SIMD leave each lane)
    const bool five_fit = !this_->classes[RT_MAT_METAL] && !this_->classes[RT_MAT_PLASTIC] && !this_->classes[RT_MAT_ROUGHGLASS];
    int nseg = env_int("RT_AMD_FRAME_BLOCKS_PER_CU", (small_tree || five_fit) ? 5 : 4) * ncu;
    nseg = std::max(RT_BLOCK, std::min(RT_SEG_BLOCKS, nseg / RT_BLOCK * RT_BLOCK));
    size_t n_pass = 0, segcap = 0;
    while (true) {
        n_pass = npix * (size_t)spb;
        if (n_pass >= (1ull << 31)) { set_error(ctx, "rt_render: pass too large"); return -1; }
        // segment capacity: a block's share of the pass, plus room for the rounding of every class kernel's share
        // (a block that takes chunks of RT_BLOCK round robin handles at most its even share rounded up to a chunk, once per class kernel)
        size_t share = (n_pass + nseg - 1) / nseg;
        share = (share + RT_BLOCK - 1) & ~(size_t)(RT_BLOCK - 1);
        segcap = share + (size_t)RT_BLOCK * (RT_NUM_CLASSES + 1);
        size_t cap = segcap * nseg;
        if (cap >= (1ull << 31)) { set_error(ctx, "rt_render: pass too large"); return -1; }
        if (cap <= ln.ws_cap) break;
        hipError_t e = ln.d_q[0].alloc(12 * cap * 4);
        if (e == hipSuccess) e = ln.d_q[1].alloc(12 * cap * 4);
        if (e == hipSuccess) e = ln.d_hits.alloc(5 * cap * 4);
        if (e == hipSuccess) e = ln.d_cls.alloc((size_t)(RT_NUM_CLASSES + 1) * cap * 4);      // one list per class + k_extend_flat's replay list
        if (e == hipSuccess) e = ln.d_shadow.alloc(10 * cap * 4);
        if (e == hipSuccess) e = ln.d_lbuf.alloc(cap * 16);
        if (e == hipSuccess) { ln.ws_cap = cap; break; }
        // someone else took the memory between the query and the allocation (another context on this GPU): give everything back and
        // try with passes of half the size; an explicit pool_paths that does not fit is the caller's error
        (void)ln.d_q[0].alloc(0); (void)ln.d_q[1].alloc(0); (void)ln.d_hits.alloc(0); (void)ln.d_cls.alloc(0);
        (void)ln.d_shadow.alloc(0); (void)ln.d_lbuf.alloc(0);
        ln.ws_cap = 0;
        (void)hipGetLastError();
        if (e != hipErrorOutOfMemory || pool_is_callers || spb == 1) { set_error(ctx, std::string("rt_render: workspace allocation: ") + hipGetErrorString(e)); return -1; }
        spb = (spb + 1) / 2;
    }
    size_t n_bounce = (size_t)o->max_depth + 2;
    HIP_OK(ln.d_counters.ensure(n_bounce * (sizeof(BounceCounts) + sizeof(BounceStats))));
    if (ln.sum_pixels != npix) { HIP_OK(ln.d_sum.alloc(npix * 3 * 4)); ln.sum_pixels = npix; ln.sum_valid = false; }
    plan.spb = spb; plan.nseg = nseg; plan.segcap = segcap;
    return 0;
}

// One lane's share of a render call: samples [o->sample_begin, o->sample_end) of every pixel into ln.d_sum, with the plan lane_prepare made.
// Runs on the calling thread; every launch goes to ln.stream.
static int render_lane(rt_scene *this_, RenderLane &ln, const rt_camera *cam, const rt_render_opts *o, const LanePlan &plan, rt_render_stats *stats) {
    rt_ctx *ctx = this_->ctx;
    HIP_OK(hipSetDevice(ctx->device));
    int W = cam->width, H = cam->height;
    size_t npix = (size_t)W * H;
    const int spb = plan.spb, nseg = plan.nseg;
    const size_t segcap = plan.segcap;
    const bool small_tree = this_->dev.n_nodes <= 4096;
    size_t n_bounce = (size_t)o->max_depth + 2;
    size_t counts_bytes = n_bounce * sizeof(BounceCounts), stats_bytes = n_bounce * sizeof(BounceStats);
    bool fresh = !ln.sum_valid || !(o->flags & RT_RENDER_ACCUMULATE);
    ln.sum_valid = true;
    if (fresh) HIP_OK(hipMemsetAsync(ln.d_sum.p, 0, npix * 3 * 4, ln.stream));

    size_t wcap = ln.ws_cap;
    PathQueue q[2]; bind_queue(q[0], ln.d_q[0].p, wcap); bind_queue(q[1], ln.d_q[1].p, wcap);
    HitBuf hb; { float *f = (float *)ln.d_hits.p; hb.slot = (int *)f; hb.t = f + wcap; hb.b0 = f + 2 * wcap; hb.b1 = f + 3 * wcap; hb.b2 = f + 4 * wcap; }
    ShadowQueue sq; { float *f = (float *)ln.d_shadow.p; sq.id = (uint32_t *)f; sq.ox = f + wcap; sq.oy = f + 2 * wcap; sq.oz = f + 3 * wcap; sq.dx = f + 4 * wcap;
                      sq.dy = f + 5 * wcap; sq.dz = f + 6 * wcap; sq.cr = f + 7 * wcap; sq.cg = f + 8 * wcap; sq.cb = f + 9 * wcap; }
    BounceCounts *counts = (BounceCounts *)ln.d_counters.p;
    BounceStats *bstats = (BounceStats *)((char *)ln.d_counters.p + counts_bytes);
    float4 *lbuf = (float4 *)ln.d_lbuf.p;
    int grid = persistent_grid(ctx);
    if (nseg * RT_BLOCK > grid * RT_BLOCK) { set_error(ctx, "rt_render: spill area smaller than the wavefront grid"); return -1; }

    RenderParams rp; memset(&rp, 0, sizeof(rp));
    memcpy(rp.r2c, cam->raster_to_camera, 64); memcpy(rp.c2w, cam->camera_to_world, 64);
    rp.width = W; rp.height = H; rp.strata_x = o->strata_x; rp.strata_y = o->strata_y; rp.max_depth = o->max_depth; rp.k0 = o->seed_lo; rp.k1 = o->seed_hi;
    rp.d_width = make_fastdiv((uint32_t)W); rp.d_strata_x = make_fastdiv((uint32_t)o->strata_x);
    if ((int64_t)o->strata_x * o->strata_y >= (1ll << 31)) { set_error(ctx, "rt_render: too many strata"); return -1; }
    rp.d_nstrata = make_fastdiv((uint32_t)(o->strata_x * o->strata_y));
    // the estimators next to Li_re run through the general shading kernels: no fused shadow rays, no k_finish
    rp.li = (o->flags & RT_RENDER_LI) ? 1 : 0; rp.rr_threshold = o->rr_threshold;
    for (int c = 0; c < RT_NUM_CLASSES; ++c) if (this_->classes[c]) rp.class_mask |= 1u << c;
    const bool all_lights = (o->flags & RT_RENDER_ALL_LIGHTS) != 0 && this_->dev.n_lights > 0;
    const bool general = rp.li || all_lights;

    rt_render_stats st; memset(&st, 0, sizeof(st));
    EventPair ev;
    HIP_OK(ev.create());
    HIP_OK(hipEventRecord(ev.a, ln.stream));
    KernelTimer tm; tm.on = (o->flags & RT_RENDER_TIME_KERNELS) != 0; tm.stream = ln.stream;
    std::vector<BounceStats> hs(n_bounce);
    const uint32_t ucap = (uint32_t)wcap, useg = (uint32_t)segcap;
    // scenes small enough are traversed out of an LDS copy (no spheres: their records stay in global memory)
    const bool lds_scene = this_->dev.n_nodes <= RT_LDS_SCENE_NODES && this_->dev.n_slots <= RT_LDS_SCENE_SLOTS && this_->interior_depth <= RT_FRAME_RING;
    const bool frame_old = env_int("RT_AMD_FRAME_NODE_WALK", 0) != 0;       // knob: the one-ray-per-lane-per-round kernels
    const int walk_mode = this_->dev.quads ? RT_WALK_QUADS : RT_WALK_PAIRS;
    // Which walk for which bounce.  Round-based kernels (one ray per lane per round, no scheduling overhead) win while the
    // rays of a wave take about the same number of steps: always on small trees, and for camera rays on large ones;
    // persistent lanes win on deep trees once the rays are incoherent.  Measured, Msamples/s: Cornell 1174 round-based
    // vs 1002 persistent; spheres 3445 vs 3169; helmet 273 round-based, 328 persistent, 346 with only bounce 0 round-based.
    const int round_bounces = env_int("RT_AMD_FRAME_ROUND_BOUNCES", small_tree ? 1 << 20 : 1);
    const int shadow_round_bounces = env_int("RT_AMD_SHADOW_ROUND_BOUNCES", round_bounces);
    const int round_pairs = (this_->interior_depth <= 8 && env_int("RT_AMD_ROUND_PAIRS", 1)) ? 1 : 0;
    const int round_quads = (this_->dev.quads && env_int("RT_AMD_ROUND_PAIRS", 1)) ? 1 : 0;
    // Round-robin chunks or contiguous shares for the round-based kernels (block_chunks).  Measured at 800x800, 256 spp, Msamples/s,
    // (extend, shade, shadow) round robin: helmet 353 (0,0,0), 421 (1,0,0), 403 (0,1,0), 411 (1,1,1), 422 (1,0,1); Cornell 1426, 1448,
    // 1425, 1430, 1439; spheres 3194, 3221, 3212, 3193, 3275.  Traversal cost varies across the image, so the traversal kernels gain
    // from dealing chunks out; the shading kernels read through index lists and do better on contiguous shares.
    const int rr_extend = env_int("RT_AMD_RR_EXTEND", 1) ? 2 : 0;
    const int rr_shadow = env_int("RT_AMD_RR_SHADOW", 1) ? 2 : 0;
    rp.rr_shade = env_int("RT_AMD_RR_SHADE", 0);
    rp.pack_shadow = env_int("RT_AMD_PACK_SHADOW", 1);
    rp.merge_mirror = (this_->classes[RT_MAT_MATTE] && this_->classes[RT_MAT_MIRROR]) ? std::max(0, std::min(2, env_int("RT_AMD_MERGE_MIRROR", 2))) : 0;
    // incoherent bounces of LDS scenes: lanes carry long rays over and refill (k_extend_carry); camera rays stay round-based
    const int carry_from = env_int("RT_AMD_CARRY_FROM", 1);
    // incoherent bounces of LDS scenes with few leaves: no walk, every leaf box tested in lockstep (k_extend_flat)
    const bool use_flat = lds_scene && round_pairs && !frame_old && this_->dev.flat != nullptr && env_int("RT_AMD_FLAT", 1) != 0;
    const int flat_from = env_int("RT_AMD_FLAT_FROM", 1);
    // later bounces on large trees: the persistent-lane walk without the reference's visit order (near-ties walked again in that order).
    // On where the scene has the eight-wide records (RT_AMD_FREE_PL=0 turns it off, =1 forces it over the quad records too).  Measured on the
    // helmet, 1440 x 1440 @ 64 spp: over the eight-wide records 597 Msamples/s against 569 in the reference's order over the quad records
    // (MetalRoughSpheres 2820 against 2725; 7 rays in 100 000 replayed, images identical).  Over the quad records nearest-first order alone
    // gives 575 against 574 — the split-axis order is as good a guess at "near" as the entry distances — and with lanes that hold an untested
    // leaf walking ahead (RT_FREE_SPECULATE, leaves tested once 8 / 16 / 32 / 48 lanes hold one): 496 / 524 / 540 / 512: the records fetched
    // ahead are wasted whenever the pending leaf holds the hit, and they cost L1 room and LDS stack traffic.
    const bool free_pl = env_int("RT_AMD_FREE_PL", this_->dev.octs ? 1 : 0) != 0;
    // shadow rays traced inside the shading kernels: scenes whose tree and tables both sit in LDS
    // (only while five blocks of the fused kernel still share a CU's 160 KB of LDS: 25.6 KB static + the scene's tables <= 32 KB)
    const bool all_in_lds = lds_scene && round_pairs && !frame_old && shade_tables_in_lds(this_->dev) && shade_lds_bytes(this_->dev) <= 6400 && !this_->dev.envs;
    // knob: shadow rays of LDS scenes through the shadow queue to k_shadow_carry (lanes refill as their any-hit walks end) instead of
    // inside the shading kernel, where a wave runs its 64 shadow rays to completion.  Measured on the Cornell frame: shading alone
    // 128 ms + k_shadow_carry 93 ms (round-based k_shadow: 106 ms) against 210 ms for the fused kernel, so fusing stays the default
    const int shadow_carry_from = (lds_scene && round_pairs && !frame_old) ? env_int("RT_AMD_SHADOW_CARRY_FROM", 1 << 30) : 1 << 30;
    const bool fuse_shadow = all_in_lds && !general && shadow_carry_from > o->max_depth && env_int("RT_AMD_FUSE_SHADOW", 1) != 0;
    // the late, near-empty bounces of LDS scenes in one launch (k_finish)
    const bool use_finish = (all_in_lds || (!lds_scene && round_quads && !frame_old)) && !this_->dev.envs && !general && env_int("RT_AMD_FINISH", 1) != 0;
    const int finish_chunks = env_int("RT_AMD_FINISH_CHUNKS", 2);
    for (int s0 = o->sample_begin; s0 < o->sample_end; s0 += spb) {
        int cur = std::min(spb, o->sample_end - s0);
        rp.spb = cur; rp.d_spb = make_fastdiv((uint32_t)cur); rp.sample0 = s0; rp.n_paths = (uint32_t)(npix * (size_t)cur);
        HIP_OK(hipMemsetAsync(bstats, 0, stats_bytes, ln.stream));
        tm.begin_pass();
        hipLaunchKernelGGL(k_raygen, dim3(nseg), dim3(RT_BLOCK), 0, ln.stream, rp, q[0], useg, lbuf, counts);
        tm.mark(RT_TIME_OTHER);
        int cq = 0;
        // k_finish takes over at the first bounce expected to hold no more than two chunks of 64 paths per wave of the grid
        int finish_at = 1 << 30;
        if (use_finish)
            for (size_t b = (size_t)env_int("RT_AMD_FINISH_MIN_BOUNCE", 4); b < ln.survival.size(); ++b)
                if ((double)ln.survival[b] * rp.n_paths <= (double)finish_chunks * RT_WAVE * (RT_BLOCK / RT_WAVE) * nseg) { finish_at = (int)b; break; }
        for (int b = 0; b <= o->max_depth; ++b) {
            if (b >= finish_at) {
                if (lds_scene)
                    hipLaunchKernelGGL(k_finish<true>, dim3(nseg), dim3(RT_BLOCK), shade_lds_bytes(this_->dev), ln.stream, this_->dev, rp, b, q[cq], useg, lbuf,
                                       counts, bstats, ln.spill);
                else
                    hipLaunchKernelGGL(k_finish<false>, dim3(nseg), dim3(RT_BLOCK), 0, ln.stream, this_->dev, rp, b, q[cq], useg, lbuf, counts, bstats,
                                       ln.spill);
                tm.mark(RT_TIME_OTHER);
                break;
            }
            if (use_flat && b >= flat_from)
                hipLaunchKernelGGL(k_extend_flat, dim3(nseg), dim3(RT_BLOCK), 0, ln.stream, this_->dev, rp, b, q[cq], hb, (uint32_t *)ln.d_cls.p,
                                   ucap, useg, lbuf, counts, bstats, this_->dev.flat, this_->dev.n_flat);
            else if (frame_old || b < round_bounces) {
                if (lds_scene && round_pairs && b >= carry_from)
                    hipLaunchKernelGGL(k_extend_carry, dim3(nseg), dim3(RT_BLOCK), 0, ln.stream, this_->dev, rp, b, q[cq], hb, (uint32_t *)ln.d_cls.p,
                                       ucap, useg, lbuf, counts, bstats);
                else if (lds_scene)
                    hipLaunchKernelGGL(k_extend<true>, dim3(nseg), dim3(RT_BLOCK), 0, ln.stream, this_->dev, rp, b, q[cq], hb, (uint32_t *)ln.d_cls.p,
                                       ucap, useg, lbuf, counts, bstats, ln.spill, round_pairs | rr_extend);
                else
                    hipLaunchKernelGGL(k_extend<false>, dim3(nseg), dim3(RT_BLOCK), 0, ln.stream, this_->dev, rp, b, q[cq], hb, (uint32_t *)ln.d_cls.p,
                                       ucap, useg, lbuf, counts, bstats, ln.spill, round_quads | rr_extend);
            } else if (walk_mode == RT_WALK_QUADS && free_pl && this_->dev.octs && env_int("RT_AMD_OCT_PL", 1))
                hipLaunchKernelGGL((k_extend_pl<RT_WALK_QUADS, true, true>), dim3(nseg), dim3(RT_BLOCK), 0, ln.stream, this_->dev, rp, b, q[cq], hb,
                                   (uint32_t *)ln.d_cls.p, ucap, useg, lbuf, counts, bstats, ln.spill);
            else if (walk_mode == RT_WALK_QUADS && free_pl)
                hipLaunchKernelGGL((k_extend_pl<RT_WALK_QUADS, true>), dim3(nseg), dim3(RT_BLOCK), 0, ln.stream, this_->dev, rp, b, q[cq], hb,
                                   (uint32_t *)ln.d_cls.p, ucap, useg, lbuf, counts, bstats, ln.spill);
            else if (walk_mode == RT_WALK_QUADS)
                hipLaunchKernelGGL(k_extend_pl<RT_WALK_QUADS>, dim3(nseg), dim3(RT_BLOCK), 0, ln.stream, this_->dev, rp, b, q[cq], hb,
                                   (uint32_t *)ln.d_cls.p, ucap, useg, lbuf, counts, bstats, ln.spill);
            else
                hipLaunchKernelGGL(k_extend_pl<RT_WALK_PAIRS>, dim3(nseg), dim3(RT_BLOCK), 0, ln.stream, this_->dev, rp, b, q[cq], hb,
                                   (uint32_t *)ln.d_cls.p, ucap, useg, lbuf, counts, bstats, ln.spill);
            tm.mark(RT_TIME_TRACE);
            if (rp.li && this_->dev.envs) {
                hipLaunchKernelGGL(k_env_escape, dim3(nseg), dim3(RT_BLOCK), 0, ln.stream, this_->dev, b, q[cq], hb, useg, lbuf, counts);
                tm.mark(RT_TIME_OTHER);
            }
            // UniformSampleAllLights: one shading launch (direct light of that light only) and one shadow launch per light, then the BSDF sampling
            const int n_steps = (all_lights && b < o->max_depth) ? this_->dev.n_lights + 1 : 1;
            for (int step = 0; step < n_steps; ++step) {
            if (all_lights) {
                rp.nee_mode = step < this_->dev.n_lights ? 1 : 2; rp.nee_light = step;
                if (step > 0) HIP_OK(hipMemsetAsync(counts[b].shadow, 0, sizeof(counts[b].shadow), ln.stream));
            }
            if (b < o->max_depth) {
                if (this_->classes[RT_MAT_MATTE]) launch_shade<RT_MAT_MATTE>(this_, ln, rp, b, q[cq], q[cq ^ 1], hb, ucap, useg, sq, nseg, fuse_shadow, lbuf, bstats);
                if (this_->classes[RT_MAT_MIRROR] && !rp.merge_mirror) launch_shade<RT_MAT_MIRROR>(this_, ln, rp, b, q[cq], q[cq ^ 1], hb, ucap, useg, sq, nseg, fuse_shadow, lbuf, bstats);
                if (this_->classes[RT_MAT_GLASS]) launch_shade<RT_MAT_GLASS>(this_, ln, rp, b, q[cq], q[cq ^ 1], hb, ucap, useg, sq, nseg, fuse_shadow, lbuf, bstats);
                if (this_->classes[RT_MAT_METAL]) launch_shade<RT_MAT_METAL>(this_, ln, rp, b, q[cq], q[cq ^ 1], hb, ucap, useg, sq, nseg, fuse_shadow, lbuf, bstats);
                if (this_->classes[RT_MAT_PLASTIC]) launch_shade<RT_MAT_PLASTIC>(this_, ln, rp, b, q[cq], q[cq ^ 1], hb, ucap, useg, sq, nseg, fuse_shadow, lbuf, bstats);
                if (this_->classes[RT_MAT_METALROUGH]) launch_shade<RT_MAT_METALROUGH>(this_, ln, rp, b, q[cq], q[cq ^ 1], hb, ucap, useg, sq, nseg, fuse_shadow, lbuf, bstats);
                if (this_->classes[RT_MAT_ROUGHGLASS]) launch_shade<RT_MAT_ROUGHGLASS>(this_, ln, rp, b, q[cq], q[cq ^ 1], hb, ucap, useg, sq, nseg, fuse_shadow, lbuf, bstats);
                tm.mark(RT_TIME_SHADE);
                if (this_->dev.n_lights > 0 && !fuse_shadow && rp.nee_mode != 2) {
                    if (frame_old || b < shadow_round_bounces || all_lights) {      // (k_shadow_pl borrows the hit buffer, which the next light's shading still reads)
                        if (b >= shadow_carry_from)
                            hipLaunchKernelGGL(k_shadow_carry, dim3(nseg), dim3(RT_BLOCK), 0, ln.stream, this_->dev, b, sq, useg, lbuf, counts, bstats);
                        else if (lds_scene)
                            hipLaunchKernelGGL(k_shadow<true>, dim3(nseg), dim3(RT_BLOCK), 0, ln.stream, this_->dev, b, sq, useg, lbuf, counts, bstats,
                                               ln.spill, round_pairs | rr_shadow);
                        else
                            hipLaunchKernelGGL(k_shadow<false>, dim3(nseg), dim3(RT_BLOCK), 0, ln.stream, this_->dev, b, sq, useg, lbuf, counts, bstats,
                                               ln.spill, round_quads | rr_shadow);
                    } else if (walk_mode == RT_WALK_QUADS && this_->dev.octs && env_int("RT_AMD_OCT_SHADOW", 1))
                        hipLaunchKernelGGL((k_shadow_pl<RT_WALK_QUADS, true>), dim3(nseg), dim3(RT_BLOCK), 0, ln.stream, this_->dev, b, sq, hb.slot, useg, lbuf,
                                           counts, bstats, ln.spill);
                    else if (walk_mode == RT_WALK_QUADS)       // the hit buffer's slot array is free once the shading kernels are done
                        hipLaunchKernelGGL(k_shadow_pl<RT_WALK_QUADS>, dim3(nseg), dim3(RT_BLOCK), 0, ln.stream, this_->dev, b, sq, hb.slot, useg, lbuf,
                                           counts, bstats, ln.spill);
                    else
                        hipLaunchKernelGGL(k_shadow_pl<RT_WALK_PAIRS>, dim3(nseg), dim3(RT_BLOCK), 0, ln.stream, this_->dev, b, sq, hb.slot, useg, lbuf,
                                           counts, bstats, ln.spill);
                    tm.mark(RT_TIME_SHADOW);
                }
            }
            }
            cq ^= 1;
        }
        if (cur >= 32) {
            int ag = (int)std::min<size_t>((npix + RT_BLOCK / RT_WAVE - 1) / (RT_BLOCK / RT_WAVE), (size_t)grid * 8);
            hipLaunchKernelGGL(k_accumulate_wave, dim3(ag), dim3(RT_BLOCK), 0, ln.stream, (const float4 *)lbuf, cur, (uint32_t)npix, (float *)ln.d_sum.p);
        } else {
            int ag = (int)std::min<size_t>((npix + RT_BLOCK - 1) / RT_BLOCK, (size_t)grid * 4);
            hipLaunchKernelGGL(k_accumulate, dim3(ag), dim3(RT_BLOCK), 0, ln.stream, (const float4 *)lbuf, cur, (uint32_t)npix, (float *)ln.d_sum.p);
        }
        tm.mark(RT_TIME_OTHER);
        HIP_OK(hipGetLastError());
        HIP_OK(hipMemcpyAsync(hs.data(), bstats, stats_bytes, hipMemcpyDeviceToHost, ln.stream));
        HIP_OK(hipStreamSynchronize(ln.stream));
        tm.end_pass();
        st.samples += rp.n_paths;
        for (size_t b = 0; b < n_bounce; ++b) { st.rays_closest += hs[b].n_queue; st.rays_shadow += hs[b].n_shadow; st.rays_replayed += hs[b].n_replay; if (hs[b].n_queue) st.iterations++; }
        if (use_finish && finish_at == 1 << 30 && rp.n_paths >= 4096) {       // a pass through every bounce: note how fast the paths die out
            ln.survival.assign((size_t)o->max_depth + 1, 0.f);
            for (int b = 0; b <= o->max_depth; ++b) ln.survival[b] = (float)((double)hs[b].n_queue / rp.n_paths);
        }
    }
    HIP_OK(hipEventRecord(ev.b, ln.stream));
    HIP_OK(hipEventSynchronize(ev.b));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, ev.a, ev.b));
    st.seconds_total = (double)ms * 1e-3;
    st.seconds_trace = tm.seconds[RT_TIME_TRACE]; st.seconds_shade = tm.seconds[RT_TIME_SHADE]; st.seconds_shadow = tm.seconds[RT_TIME_SHADOW];
    st.seconds_other = tm.seconds[RT_TIME_OTHER];
    if (stats) *stats = st;
    return 0;
}

int rt_render_device(rt_scene *this_, const rt_camera *cam, const rt_render_opts *o, void **out_device_sum, rt_render_stats *stats) {
    if (!this_ || !cam || !o || !out_device_sum) return -1;
    rt_ctx *ctx = this_->ctx;
    HIP_OK(hipSetDevice(ctx->device));
    int W = cam->width, H = cam->height;
    if (W <= 0 || H <= 0 || o->strata_x <= 0 || o->strata_y <= 0 || o->sample_begin < 0 || o->sample_end < o->sample_begin || o->max_depth < 0 || o->max_depth > 65534) {
        set_error(ctx, "rt_render: invalid camera or options"); return -1;
    }
    size_t npix = (size_t)W * H;
    if (npix > (1u << 30)) { set_error(ctx, "rt_render: image too large"); return -1; }
    const int S = o->sample_end - o->sample_begin;
    // Lanes.  Two independent pipelines over half of the samples each finish a frame sooner than one pipeline over all of them: the
    // kernels of one lane fill the other's tails (every bounce ends with a few blocks still running) and the memory waits of its
    // shading kernels.  Measured on the 800 x 800, 1024 spp Cornell frame: 1608 Msamples/s with one lane, 1869 with two, 1891 with three,
    // 1790 with four; unequal shares (56 / 62 / 70 % to the first of two lanes, so that their tails would not coincide) are slower than equal
    // ones.  Small renders stay on one lane, and so does a call that asks for per-class kernel times (the events that
    // bracket every launch would serialise the lanes).
    // (a third lane pays once a call brings half a billion paths: 1881 -> 1896 on that frame, helmet 1440 x 1440 @ 1024 spp 611 -> 629; on the
    // 128-spp share that one of eight ranks renders two lanes take 48.0 ms and three 49.2)
    const size_t paths_total = npix * (size_t)std::max(S, 0);
    int n_lanes = std::max(1, std::min(RT_MAX_LANES, env_int("RT_AMD_RENDER_LANES", paths_total >= ((size_t)512 << 20) ? 3 : 2)));
    if ((o->flags & RT_RENDER_TIME_KERNELS) || paths_total < ((size_t)16 << 20)) n_lanes = 1;
    n_lanes = std::max(1, std::min(n_lanes, S));
    // Paths per pass.  Every pass ends in a tail of near-empty bounces whose cost is fixed, so passes should be few and large:
    // 200 bytes of queue state per path (two queues of 48, hits 20, class lists 28, shadow queue 40, radiance slot 16); up to half of the
    // free HBM goes to it (at most 2^30 paths), shared out over the lanes, so that the 655 M paths of an 800 x 800, 1024 spp frame are ONE pass
    // per lane on a 288 GB card: measured 1572 Msamples/s with three passes, 1598 with two, 1622 with one (one lane).
    size_t pool = (size_t)o->pool_paths;
    if (o->pool_paths <= 0) {
        size_t free_b = 0, total_b = 0;
        HIP_OK(hipMemGetInfo(&free_b, &total_b));
        for (int k = 0; k < RT_MAX_LANES; ++k) free_b += this_->lanes[k].ws_cap * 200;       // the workspaces of an earlier call are reused
        pool = std::min<size_t>((size_t)1 << 30, std::max<size_t>((size_t)1 << 20, free_b / 2 / 200));
    }
    pool = std::max<size_t>(1, pool / n_lanes);
    for (int k = n_lanes; k < RT_MAX_LANES; ++k) {         // lanes this call does not use give their queues back
        RenderLane &ln = this_->lanes[k];
        if (ln.ws_cap) { (void)ln.d_q[0].alloc(0); (void)ln.d_q[1].alloc(0); (void)ln.d_hits.alloc(0); (void)ln.d_cls.alloc(0); (void)ln.d_shadow.alloc(0); (void)ln.d_lbuf.alloc(0); ln.ws_cap = 0; }
    }
    this_->lanes[0].stream = ctx->stream; this_->lanes[0].spill = (int *)this_->d_spill.p;
    for (int k = 1; k < n_lanes; ++k) {
        RenderLane &ln = this_->lanes[k];
        if (!ln.stream) HIP_OK(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
        HIP_OK(ln.d_spill_own.ensure(this_->spill_bytes));
        ln.spill = (int *)ln.d_spill_own.p;
    }
    std::vector<rt_render_opts> lo((size_t)n_lanes, *o);
    std::vector<rt_render_stats> ls((size_t)n_lanes);
    std::vector<int> rc((size_t)n_lanes, 0);
    for (int k = 0; k < n_lanes; ++k) {
        int32_t b = 0, e = 0;
        rt_split_samples(S, k, n_lanes, &b, &e);
        lo[k].sample_begin = o->sample_begin + b; lo[k].sample_end = o->sample_begin + e;
        if (k) lo[k].flags &= ~RT_RENDER_ACCUMULATE;       // only lane 0's buffer carries an earlier call's sums
        memset(&ls[k], 0, sizeof(ls[k]));
    }
    std::vector<LanePlan> plan((size_t)n_lanes);
    for (int k = 0; k < n_lanes; ++k) {
        int r = lane_prepare(this_, this_->lanes[k], cam, &lo[k], pool, o->pool_paths > 0, plan[k]);
        if (r != 0) return r;
    }
    EventPair ev;
    HIP_OK(ev.create());
    HIP_OK(hipEventRecord(ev.a, ctx->stream));
    {
        std::vector<std::thread> th;
        for (int k = 1; k < n_lanes; ++k)
            th.emplace_back([&, k]() { rc[k] = render_lane(this_, this_->lanes[k], cam, &lo[k], plan[k], &ls[k]); });
        rc[0] = render_lane(this_, this_->lanes[0], cam, &lo[0], plan[0], &ls[0]);
        for (auto &t : th) t.join();
    }
    for (int k = 0; k < n_lanes; ++k) if (rc[k] != 0) return rc[k];
    // every lane has synchronised its stream; the other lanes' sums are added to lane 0's in lane order, so the result does not depend on timing
    for (int k = 1; k < n_lanes; ++k)
        hipLaunchKernelGGL(k_add_frames, dim3(std::max(1, ctx->n_cu) * 4), dim3(RT_BLOCK), 0, ctx->stream, (float *)this_->lanes[0].d_sum.p,
                           (const float *)this_->lanes[k].d_sum.p, npix * 3);
    HIP_OK(hipGetLastError());
    HIP_OK(hipEventRecord(ev.b, ctx->stream));
    HIP_OK(hipEventSynchronize(ev.b));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, ev.a, ev.b));
    rt_render_stats st = ls[0];
    for (int k = 1; k < n_lanes; ++k) {
        st.samples += ls[k].samples; st.rays_closest += ls[k].rays_closest; st.rays_shadow += ls[k].rays_shadow; st.iterations += ls[k].iterations; st.rays_replayed += ls[k].rays_replayed;
    }
    st.seconds_total = (double)ms * 1e-3;
    if (stats) *stats = st;
    *out_device_sum = this_->lanes[0].d_sum.p;
    return 0;
}

// ---------------------------------------------------------------- several GPUs --
}  // extern "C"


struct rt_multi {
    std::vector<rt_ctx *> ctx;
    std::vector<rt_scene *> scene;
    // on device[0], one set per other device: a frame-sized staging buffer, the stream its peer copy runs on, the event that says it has landed
    std::vector<std::unique_ptr<DevBuf>> staging;
    std::vector<hipStream_t> copy_stream;
    std::vector<hipEvent_t> copied;
    std::string error;
};
static std::string g_multi_error;

extern "C" {

int rt_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int rt_split_samples(int32_t total, int32_t rank, int32_t world, int32_t *begin, int32_t *end) {
    if (total < 0 || world < 1 || rank < 0 || rank >= world || !begin || !end) return -1;
    const int32_t base = total / world, rem = total % world;
    *begin = rank * base + std::min(rank, rem);
    *end = *begin + base + (rank < rem ? 1 : 0);
    return 0;
}

const char *rt_multi_last_error(rt_multi *m) { return m ? m->error.c_str() : g_multi_error.c_str(); }

void rt_multi_destroy(rt_multi *m) {
    if (!m) return;
    for (size_t i = 0; i < m->scene.size(); ++i) rt_scene_destroy(m->scene[i]);
    if (!m->ctx.empty()) {
        (void)hipSetDevice(m->ctx[0]->device);
        m->staging.clear();
        for (hipStream_t st : m->copy_stream) if (st) (void)hipStreamDestroy(st);
        for (hipEvent_t ev : m->copied) if (ev) (void)hipEventDestroy(ev);
    }
    for (size_t i = 0; i < m->ctx.size(); ++i) rt_shutdown(m->ctx[i]);
    delete m;
}

int rt_multi_create(const int32_t *devices, int32_t n_devices, const rt_scene_desc *desc, rt_multi **out) {
    if (!devices || n_devices < 1 || !desc || !out) { g_multi_error = "rt_multi_create: bad arguments"; return -1; }
    std::unique_ptr<rt_multi> m(new rt_multi);
    for (int i = 0; i < n_devices; ++i) {
        rt_ctx *c = nullptr;
        if (rt_init(devices[i], &c) != 0) { g_multi_error = std::string("rt_multi_create: ") + rt_last_error(nullptr); rt_multi_destroy(m.release()); return -2; }
        m->ctx.push_back(c);
        rt_scene *sc = nullptr;
        if (rt_scene_create(c, desc, &sc) != 0) { g_multi_error = std::string("rt_multi_create: ") + rt_last_error(c); rt_multi_destroy(m.release()); return -1; }
        m->scene.push_back(sc);
    }
    // the merge copies straight between devices where the hardware allows it (xGMI); where it does not, hipMemcpyPeer stages through the host
    for (int i = 1; i < n_devices; ++i) {
        if (devices[i] == devices[0]) continue;
        int can = 0;
        (void)hipSetDevice(devices[0]);
        if (hipDeviceCanAccessPeer(&can, devices[0], devices[i]) == hipSuccess && can) { hipError_t e = hipDeviceEnablePeerAccess(devices[i], 0); if (e != hipSuccess) (void)hipGetLastError(); }
    }
    *out = m.release();
    return 0;
}

int rt_multi_render(rt_multi *m, const rt_camera *cam, const rt_render_opts *opts, float *out_sum_rgb, rt_render_stats *stats_per_device,
                    rt_render_stats *stats_total) {
    if (!m || !cam || !opts || !out_sum_rgb) return -1;
    const int n = (int)m->ctx.size();
    const int32_t total = opts->sample_end - opts->sample_begin;
    if (total < 0) { m->error = "rt_multi_render: empty or reversed sample range"; return -1; }
    if (opts->flags & RT_RENDER_ACCUMULATE) { m->error = "rt_multi_render: RT_RENDER_ACCUMULATE is not supported (every call merges fresh per-device sums)"; return -1; }
    std::vector<int> rc(n, 0);
    std::vector<void *> dsum(n, nullptr);
    std::vector<rt_render_stats> st(n);
    std::vector<std::thread> pool;
    for (int i = 0; i < n; ++i)
        pool.emplace_back([&, i]() {
            rt_render_opts o = *opts;
            int32_t b = 0, e = 0;
            rt_split_samples(total, i, n, &b, &e);
            o.sample_begin = opts->sample_begin + b; o.sample_end = opts->sample_begin + e;
            memset(&st[i], 0, sizeof(st[i]));
            rc[i] = rt_render_device(m->scene[i], cam, &o, &dsum[i], &st[i]);
        });
    for (auto &t : pool) t.join();
    for (int i = 0; i < n; ++i) if (rc[i] != 0) { m->error = std::string("rt_multi_render: device ") + std::to_string(i) + ": " + rt_last_error(m->ctx[i]); return rc[i]; }
    // one merge per frame: the other devices' sums are added onto device[0]'s, in device order (the result does not depend on timing)
    const size_t nfl = (size_t)cam->width * cam->height * 3, bytes = nfl * sizeof(float);
    rt_ctx *root = m->ctx[0];
    auto fail = [&](const char *what, hipError_t e) { m->error = std::string("rt_multi_render: ") + what + ": " + hipGetErrorString(e); return -2; };
    hipError_t e;
    if ((e = hipSetDevice(root->device)) != hipSuccess) return fail("hipSetDevice", e);
    // every other device's frame travels on a stream of its own into a staging buffer of its own (the copies run side by side, one xGMI
    // link each); the adds then follow on the root's stream in device order, each waiting only for its own copy
    while ((int)m->staging.size() < n - 1) {
        m->staging.emplace_back(new DevBuf); m->copy_stream.push_back(nullptr); m->copied.push_back(nullptr);
        if ((e = hipStreamCreateWithFlags(&m->copy_stream.back(), hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
        if ((e = hipEventCreateWithFlags(&m->copied.back(), hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
    }
    for (int i = 1; i < n; ++i) {
        if ((e = m->staging[i - 1]->ensure(bytes)) != hipSuccess) return fail("staging buffer", e);
        if ((e = hipMemcpyPeerAsync(m->staging[i - 1]->p, root->device, dsum[i], m->ctx[i]->device, bytes, m->copy_stream[i - 1])) != hipSuccess) return fail("hipMemcpyPeerAsync", e);
        if ((e = hipEventRecord(m->copied[i - 1], m->copy_stream[i - 1])) != hipSuccess) return fail("hipEventRecord", e);
    }
    for (int i = 1; i < n; ++i) {
        if ((e = hipStreamWaitEvent(root->stream, m->copied[i - 1], 0)) != hipSuccess) return fail("hipStreamWaitEvent", e);
        hipLaunchKernelGGL(k_add_frames, dim3(std::min<size_t>((nfl + RT_BLOCK - 1) / RT_BLOCK, (size_t)root->n_cu * 8)), dim3(RT_BLOCK), 0, root->stream,
                           (float *)dsum[0], (const float *)m->staging[i - 1]->p, nfl);
    }
    if ((e = hipGetLastError()) != hipSuccess) return fail("k_add_frames", e);
    if ((e = hipMemcpyAsync(out_sum_rgb, dsum[0], bytes, hipMemcpyDeviceToHost, root->stream)) != hipSuccess) return fail("hipMemcpyAsync", e);
    if ((e = hipStreamSynchronize(root->stream)) != hipSuccess) return fail("hipStreamSynchronize", e);
    if (stats_per_device) memcpy(stats_per_device, st.data(), sizeof(rt_render_stats) * n);
    if (stats_total) {
        rt_render_stats t; memset(&t, 0, sizeof(t));
        for (int i = 0; i < n; ++i) {
            t.samples += st[i].samples; t.rays_closest += st[i].rays_closest; t.rays_shadow += st[i].rays_shadow; t.iterations += st[i].iterations; t.rays_replayed += st[i].rays_replayed;
            t.seconds_total = std::max(t.seconds_total, st[i].seconds_total);      // the devices ran side by side
            t.seconds_trace += st[i].seconds_trace; t.seconds_shade += st[i].seconds_shade; t.seconds_shadow += st[i].seconds_shadow; t.seconds_other += st[i].seconds_other;
        }
        *stats_total = t;
    }
    return 0;
}

int rt_render(rt_scene *this_, const rt_camera *cam, const rt_render_opts *opts, float *out_sum_rgb, rt_render_stats *stats) {
    if (!this_ || !out_sum_rgb) return -1;
    void *dsum = nullptr;
    int rc = rt_render_device(this_, cam, opts, &dsum, stats);
    if (rc != 0) return rc;
    rt_ctx *ctx = this_->ctx;
    HIP_OK(hipMemcpy(out_sum_rgb, dsum, (size_t)cam->width * cam->height * 3 * 4, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"

/*
 * rt_amd.h — C ABI of the MI355X wavefront path-tracing backend.
 *
 * The reference renderer (sedcftyv/RayTracing) has no FFI or plugin API; its two
 * seams are C++ virtual interfaces:
 *   - Integrator::Render(const Scene&, int w, int h)      include/core/integrator.h:17-21
 *   - Primitive::Intersect / IntersectP (Aggregate)       include/core/primitive.h:10-19,39-44
 * Every entry point below replaces one of those seams (or a helper that the
 * reference calls behind them) and cites it.  The host-side C++ classes under
 * raytracing_amd/host/ keep the reference's class names and constructor
 * signatures and call into this ABI from PathIntegrator::Render.
 *
 * Conventions: plain pointers and sizes, caller-owned host memory unless a
 * function says "device"; the library copies what it is given; 0 = success,
 * negative = error (rt_last_error() returns the text); no exceptions cross the
 * boundary; one host thread drives a context at a time; calls are synchronous
 * unless the name ends in _async.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- data -- */

/* One flattened BVH node; byte-for-byte the reference's LinearBVHNode
 * (include/core/bvh.cpp:43-47): depth-first order, first child = index+1. */
typedef struct rt_bvh_node {
    float    bmin[3];
    float    bmax[3];
    int32_t  offset;   /* leaf: first slot in prim_order; interior: index of second child */
    uint16_t nprim;    /* 0 = interior */
    uint8_t  axis;     /* interior: split axis 0/1/2 */
    uint8_t  pad;
} rt_bvh_node;

enum { RT_PRIM_TRIANGLE = 0, RT_PRIM_SPHERE = 1 };

/* A sphere as the reference stores it (include/core/sphere.h, sphere.cpp:29-73):
 * radius plus both 4x4 row-major matrices of its Transform. */
typedef struct rt_sphere {
    float radius;
    float o2w[16];
    float w2o[16];
} rt_sphere;

enum {
    RT_TEX_CONST = 0,  /* ConstantTexture<Spectrum> or <Float> (value[0] used for Float) */
    RT_TEX_IMAGE = 1   /* ImageTexture<RGBSpectrum,Spectrum>: level-0 float RGB, Repeat, bilinear */
};

/* include/core/texture.h:18-25 (constant), include/core/imagemap.h:36-51 +
 * include/core/mipmap.h:190-200 (image, level 0 after the pow-2 resample). */
typedef struct rt_texture {
    int32_t      kind;
    int32_t      width, height;   /* image only */
    float        value[3];        /* constant only */
    const float *texels;          /* image only: height*width*3 floats, row-major, row 0 first */
    float        su, sv, du, dv;  /* UVMapping2D (include/core/texture.cpp:5-13); image only */
} rt_texture;

enum {
    RT_MAT_MATTE      = 0,  /* tex[0]=Kd                                   material.cpp:5-10  */
    RT_MAT_MIRROR     = 1,  /* tex[0]=Kr                                   material.cpp:12-20 */
    RT_MAT_GLASS      = 2,  /* tex[0]=Kr tex[1]=Kt tex[2]=index            material.cpp:22-30 */
    RT_MAT_METAL      = 3,  /* tex[0]=eta tex[1]=k tex[2]=urough tex[3]=vrough, flag remap
                               (urough/vrough already resolved against `roughness`) material.cpp:51-67 */
    RT_MAT_PLASTIC    = 4,  /* tex[0]=Kd tex[1]=Ks tex[2]=roughness, flag remap material.cpp:32-49 */
    RT_MAT_METALROUGH = 5,  /* tex[0]=base_color tex[1]=metalroughness or -1    material.cpp:69-82 */
    RT_MAT_ROUGHGLASS = 6   /* tex[0]=Kr tex[1]=Kt tex[2]=index tex[3]=roughness, flag remap.  No material of the reference builds
                               it: this is the carrier of its MicrofacetTransmission lobe (reflection.cpp:311-355), paired, as in
                               pbrt's glass, with MicrofacetReflection under FresnelDielectric(1, index) (reflection.cpp:241-266) */
};

typedef struct rt_material {
    int32_t kind;
    int32_t tex[4];          /* indices into rt_scene_desc.textures, -1 = absent */
    int32_t remap_roughness; /* METAL / PLASTIC */
    int32_t pad;
} rt_material;

enum { RT_LIGHT_AREA = 0, RT_LIGHT_INFINITE = 1 };

/* InfiniteAreaLight (include/core/infinite.h:14-40, infinite.cpp:4-106): everything Sample_Li reads, as the constructor leaves it.
 * texels: level 0 of Lmap (after the resample to powers of two, mipmap.h:70-121), row-major RGB floats.  The Distribution2D
 * (include/core/sampling.h:15-91, sampling.cpp:113-123) over the nu x nv = 2 width x 2 height image of luminance x sin(theta):
 * row v of func / cdf is pConditionalV[v] (nu values / nu + 1 values), func_int[v] its funcInt (= the marginal's func),
 * marginal_cdf / marginal_int the marginal's cdf (nv + 1 values) and funcInt.  world_radius: Scene::WorldBound().BoundingSphere
 * (infinite.h:20-22, scene.cpp:6-13). */
typedef struct rt_env_light {
    float        light_to_world[16], world_to_light[16];
    int32_t      width, height;
    const float *texels;
    int32_t      nu, nv;
    const float *func;
    const float *cdf;
    const float *func_int;
    const float *marginal_cdf;
    float        marginal_int;
    float        world_center[3];
    float        world_radius;
} rt_env_light;

/* One entry of Scene::lights (include/core/scene.h:18), in the scene script's order: the index is what
 * UniformSampleOneLight picks (integrator.cpp:63).  RT_LIGHT_AREA: a DiffuseAreaLight on one triangle
 * (include/core/light.cpp:28-54, light.h:68-89).  RT_LIGHT_INFINITE: env names an rt_env_light, prim is -1. */
typedef struct rt_light {
    int32_t prim;        /* index into the ORIGINAL primitive list */
    float   Le[3];
    int32_t two_sided;
    float   area;        /* DiffuseAreaLight::area (light.cpp:35), which only Power() reads.  Not used by the library: sampling
                            divides by Triangle::Area() computed from the vertices (triangle.cpp:222-227,246), as the reference does */
    int32_t flip_normal; /* the shape's reverseOrientation ^ transformSwapsHandedness: Triangle::Sample turns the sampled
                            normal round when the mesh has no vertex normals (triangle.cpp:237-243); Intersect never does */
    int32_t kind;        /* RT_LIGHT_* */
    int32_t env;         /* RT_LIGHT_INFINITE: index into rt_scene_desc.env_lights */
} rt_light;

/* The whole scene, flattened.  Primitive arrays are in the ORIGINAL order the
 * scene script pushed them (RayTracing.cpp: prims.push_back); prim_order maps a
 * BVH leaf slot to that original index (the reference's orderedPrims,
 * bvh.cpp:97-103).  bvh_nodes may be NULL: rt_scene_create then builds the tree
 * with rt_bvh_build. */
typedef struct rt_scene_desc {
    int32_t            n_prims;
    const uint8_t     *prim_kind;      /* RT_PRIM_*                                  */
    const int32_t     *prim_geom;      /* index into tri_* or spheres                */
    const int32_t     *prim_material;  /* index into materials                       */
    const int32_t     *prim_light;     /* index into lights, -1 = not an emitter     */

    int32_t            n_tris;
    const float       *tri_p;          /* n_tris*9: world-space p0,p1,p2 (triangle.cpp:19-20) */
    const float       *tri_uv;         /* n_tris*6: uv0,uv1,uv2 ((0,0),(1,0),(1,1) if the mesh has none, triangle.h:46-57) */
    const float       *tri_n;          /* n_tris*9 world-space vertex normals, or NULL */
    const uint8_t     *tri_has_n;      /* n_tris, or NULL */

    int32_t            n_spheres;
    const rt_sphere   *spheres;

    int32_t            n_materials;
    const rt_material *materials;
    int32_t            n_textures;
    const rt_texture  *textures;
    int32_t            n_lights;
    const rt_light    *lights;

    int32_t            n_nodes;
    const rt_bvh_node *bvh_nodes;      /* optional */
    const int32_t     *prim_order;     /* n_prims, required iff bvh_nodes != NULL */

    int32_t             n_env_lights;
    const rt_env_light *env_lights;
} rt_scene_desc;

/* PerspectiveCamera state that GenerateRay reads (include/core/camera.cpp:52-58,
 * camera.h:24-45): both row-major 4x4 matrices. */
typedef struct rt_camera {
    float   raster_to_camera[16];
    float   camera_to_world[16];
    int32_t width, height;
} rt_camera;

typedef struct rt_render_opts {
    int32_t  strata_x, strata_y;   /* StratifiedSampler(x, y, jitter=true, 1): spp = x*y (stratified.h:11-15) */
    int32_t  sample_begin, sample_end; /* this call renders samples [begin,end) of every pixel */
    int32_t  max_depth;            /* PathIntegrator maxDepth (RayTracing.cpp:431 passes 50) */
    uint32_t seed_lo, seed_hi;     /* Philox key */
    int32_t  pool_paths;           /* path slots kept in flight (200 bytes of device memory each); 0 = library default: half of the free
                                      device memory, at most 2^30, halved again if the allocation fails */
    int32_t  flags;                /* RT_RENDER_* */
    float    rr_threshold;         /* PathIntegrator's rrThreshold (integrator.h:72, the reference's default is 1); read only under
                                      RT_RENDER_LI; 0 switches Li's Russian roulette off (every path then runs to max_depth) */
} rt_render_opts;

enum {
    RT_RENDER_ACCUMULATE   = 1, /* add onto the existing contents of the sum buffer instead of clearing it */
    RT_RENDER_TIME_KERNELS = 2, /* bracket every launch with HIP events and fill seconds_trace / _shade / _shadow / _other
                                   (costs a few percent: the events serialise the launches) */
    RT_RENDER_LI           = 4, /* estimate with PathIntegrator::Li (integrator.cpp:175-229) instead of Li_re (:236-281), which is what
                                   the reference's Render calls: emitted light is added wherever a camera or specular ray lands, without ending
                                   the path; rays that leave the scene see the infinite lights; the light is picked through the uniform
                                   Distribution1D; Russian roulette tests the whole path's throughput times etaScale against rr_threshold */
    RT_RENDER_ALL_LIGHTS   = 8  /* direct lighting by UniformSampleAllLights (integrator.cpp:26-51, one sample per light) instead of
                                   UniformSampleOneLight; costs one shading and one shadow launch per light and bounce */
};

typedef struct rt_render_stats {
    uint64_t samples;          /* camera paths started */
    uint64_t rays_closest;     /* closest-hit traversals */
    uint64_t rays_shadow;      /* any-hit traversals */
    uint64_t iterations;       /* bounce levels that had launches of their own, summed over passes (the near-empty late
                                  bounces of a pass run inside one launch and count as one level) */
    double   seconds_total;    /* device time, HIP events on the library's stream */
    /* The four below are filled only under RT_RENDER_TIME_KERNELS, else 0: */
    double   seconds_trace;    /* closest-hit kernels (k_extend*)                                          */
    double   seconds_shade;    /* shading kernels (k_shade*; on scenes that fit LDS they trace the shadow ray too) */
    double   seconds_shadow;   /* any-hit kernels (k_shadow*)                                              */
    double   seconds_other;    /* ray generation, late-bounce kernel (k_finish), per-pixel accumulation    */
    uint64_t rays_replayed;    /* closest-hit rays whose order-free test was not conclusive (two candidates nearly tied, a third leaf
                                  in reach, a direction with a zero component) and that were walked again in the order of
                                  BVHAccel::Intersect (bvh.cpp:247-282); counted in rays_closest once */
} rt_render_stats;

typedef struct rt_trace_stats {
    uint64_t nodes_visited;    /* slab tests performed (reference traversal order) */
    uint64_t prims_tested;     /* primitive tests performed */
    double   seconds;          /* average kernel time over `repeats`, HIP events */
    uint64_t rays_replayed;    /* rays of the last launch whose order-free walk was not conclusive and that were walked again in the
                                  order of BVHAccel::Intersect (rt_trace_closest_device only) */
} rt_trace_stats;

typedef struct rt_ctx   rt_ctx;
typedef struct rt_scene rt_scene;

/* ------------------------------------------------------------ functions -- */

/* Library version string. */
const char *rt_version(void);

/* Create a context bound to HIP device `device` (one process per GPU).
 * Fails (negative) when no HIP device is present: there is no CPU fallback. */
int rt_init(int device, rt_ctx **out);
void rt_shutdown(rt_ctx *ctx);
const char *rt_last_error(rt_ctx *ctx);   /* ctx may be NULL for rt_init failures */

/* Host-side BVH builder with the reference's decisions (BVHAccel::BVHAccel /
 * recursiveBuild / flattenBVHTree, include/core/bvh.cpp:63-245; SplitMethod::Middle,
 * maxPrimsInNode = 1).  prim_bounds = n*6 floats (min xyz, max xyz) in original
 * order.  nodes_out must hold 2*n-1 nodes, order_out n ints.  Needs no GPU. */
int rt_bvh_build(int32_t n_prims, const float *prim_bounds,
                 rt_bvh_node *nodes_out, int32_t *n_nodes_out, int32_t *order_out);

/* The same builder with the other two constructor arguments of BVHAccel (include/core/bvh.h:17-20):
 * maxPrimsInNode (clamped to 255, bvh.cpp:65; only the SAH rule reads it, bvh.cpp:195) and the split method, numbered
 * as the reference's enum.  HLBVH has no builder of its own in the reference and takes the SAH case (bvh.cpp:143-144). */
enum { RT_SPLIT_SAH = 0, RT_SPLIT_HLBVH = 1, RT_SPLIT_MIDDLE = 2, RT_SPLIT_EQUAL_COUNTS = 3 };
int rt_bvh_build_ex(int32_t n_prims, const float *prim_bounds, int32_t max_prims_in_node, int32_t split_method,
                    rt_bvh_node *nodes_out, int32_t *n_nodes_out, int32_t *order_out);

/* World bounds of every primitive of a desc, n_prims*6 floats
 * (Triangle::WorldBound triangle.cpp:45-50, Shape::WorldBound shape.cpp:16 for spheres). */
int rt_prim_bounds(const rt_scene_desc *desc, float *bounds_out);

/* The checks rt_scene_create runs before it follows any index of a description: required arrays present, every
 * primitive / material / texture / light index in range, and (when bvh_nodes is given) a tree the kernels can walk — links
 * in range and forward, every node and leaf slot reached exactly once, prim_order a permutation, at most 64 interior levels
 * (the reference's own traversal stack, include/core/bvh.cpp:253).  0 = consistent; otherwise negative and msg (may be NULL)
 * receives the reason.  Needs no GPU.  The reference has no counterpart: its object graph cannot be inconsistent. */
int rt_scene_desc_check(const rt_scene_desc *desc, char *msg, size_t msg_cap);

/* Flatten + upload a scene (replaces the pointer graph behind Scene/BVHAccel,
 * include/core/scene.h:9-24, bvh.h:15-37).  Fails (negative) on a description rt_scene_desc_check rejects. */
int rt_scene_create(rt_ctx *ctx, const rt_scene_desc *desc, rt_scene **out);
void rt_scene_destroy(rt_scene *scene);
/* Copy the tree the scene uses back to the caller (for parity checks). */
int rt_scene_get_bvh(rt_scene *scene, rt_bvh_node *nodes_out, int32_t *n_nodes_out, int32_t *order_out);

/* Closest hit for a batch of rays — Scene::Intersect -> BVHAccel::Intersect
 * (include/core/scene.cpp:15-18, bvh.cpp:247-282).  o, d: n*3 floats; tmax: n floats.
 * out_prim: original primitive index or -1; out_t: hit distance (tmax on miss);
 * out_b: n*2 barycentrics b1,b2 for triangles / (u,v) for spheres (may be NULL).
 * Host pointers. */
int rt_trace_closest(rt_scene *scene, size_t n, const float *o, const float *d, const float *tmax,
                     int32_t *out_prim, float *out_t, float *out_b);

/* Any hit — Scene::IntersectP -> BVHAccel::IntersectP (scene.cpp:20-23, bvh.cpp:284-320). */
int rt_trace_any(rt_scene *scene, size_t n, const float *o, const float *d, const float *tmax,
                 uint8_t *out_hit);

/* Device-resident variants for measurement: pointers are HIP device pointers on
 * the context's device, rays laid out SoA (ox[n] oy[n] oz[n] dx[n] dy[n] dz[n] tmax[n]
 * contiguous in `rays`), results SoA (prim[n] as int32, then t[n], b1[n], b2[n]
 * as float, contiguous in `hits`).  The kernel is launched `repeats` times on
 * the library stream and timed with HIP events; stats may be NULL. */
int rt_trace_closest_device(rt_scene *scene, size_t n, const void *rays, void *hits,
                            int repeats, int count_work, rt_trace_stats *stats);

/* rt_trace_closest_device (count_work / stats as there, one launch, same rays / hits layout, synchronous) for callers that
 * cannot allocate device memory: tight serving loops, harnesses that forbid hipMalloc, set-up phases of code that later runs under
 * HIP graph capture.  (The call itself launches on the library's own stream and waits for the result, so it cannot be captured
 * into a graph as it stands.)  It has no counterpart in the reference.  The call allocates nothing: the traversal stacks' spill area
 * and the work counters, which the other entry points keep in buffers the library allocates, live in a static array of 32 MiB +
 * 256 bytes in device memory that every GPU the library runs on holds until the process ends (its address is taken once per
 * context, in rt_init).  The results are those of rt_trace_closest_device, bit for bit (the walk is always the one in the reference's
 * order over the quad records, which needs nothing beyond the stacks); the grid is cut down to the
 * blocks whose lanes' stacks fit the array: 32 MiB / (256 lanes x 8 bytes x max(64, 3 * (interior levels / 2 + 1) + 2) entries),
 * i.e. 256 blocks for trees of up to 39 interior levels and 162 at the 64 levels rt_scene_create accepts.  The two HIP events that
 * time the launch belong to the context and are shared by its scenes.  Limits, refused with an error: n > 2^28 rays; count_work
 * without stats.  All calls on one GPU share that GPU's array, so calls that overlap on one GPU (two contexts on the same device,
 * from two threads) race; the array keeps the last call's contents, and the call clears the part that must start at zero (the
 * counters). */
int rt_trace_closest_device_noalloc(rt_scene *scene, size_t n, const void *rays, void *hits, int count_work, rt_trace_stats *stats);

/* Render samples [sample_begin, sample_end) of every pixel and return the
 * per-pixel SUM of radiance (not the mean) — the loop of
 * SamplerIntegrator::Render / render_pixel (include/core/integrator.cpp:317-412)
 * around PathIntegrator::Li_re (integrator.cpp:236-281).
 * out_sum_rgb: width*height*3 floats, row 0 = raster y 0, host memory. */
int rt_render(rt_scene *scene, const rt_camera *cam, const rt_render_opts *opts,
              float *out_sum_rgb, rt_render_stats *stats);

/* Same, but the sum stays in a device buffer owned by the scene
 * (width*height*3 floats); *out_device_sum receives its device pointer.  Used by
 * the multi-GPU driver, which reduces the buffers with RCCL. */
int rt_render_device(rt_scene *scene, const rt_camera *cam, const rt_render_opts *opts,
                     void **out_device_sum, rt_render_stats *stats);

/* ---- every GPU of the node from one process ----
 * The reference renders a frame with one call, Integrator::Render (include/core/integrator.h:17-21), on a pool of threads
 * that share the pixels (integrator.cpp:378-405).  Here the samples of every pixel are shared out instead: device k of n
 * renders samples rt_split_samples(k, n) of every pixel on its own copy of the scene, driven by its own host thread, and the
 * per-device sum buffers are merged onto devices[0] once per frame (peer copies over xGMI + an add, in device order).  A device
 * may be listed more than once (two contexts on one GPU). */
typedef struct rt_multi rt_multi;
int rt_device_count(void);                                  /* HIP devices visible to the process (0 without a GPU) */
/* The share of [0, total) that rank `rank` of `world` renders: contiguous, exhaustive, sizes differ by at most one. Host only. */
int rt_split_samples(int32_t total, int32_t rank, int32_t world, int32_t *begin, int32_t *end);
int rt_multi_create(const int32_t *devices, int32_t n_devices, const rt_scene_desc *desc, rt_multi **out);
void rt_multi_destroy(rt_multi *m);
const char *rt_multi_last_error(rt_multi *m);               /* m may be NULL for rt_multi_create failures */
/* rt_render over all devices of m: out_sum_rgb as in rt_render; stats_per_device (n_devices records) and stats_total
 * (counts summed, seconds_total = slowest device) may be NULL.  RT_RENDER_ACCUMULATE is refused with an error: every call
 * merges fresh per-device sums (add frames on the host, or use rt_render_device per device, to accumulate). */
int rt_multi_render(rt_multi *m, const rt_camera *cam, const rt_render_opts *opts, float *out_sum_rgb,
                    rt_render_stats *stats_per_device, rt_render_stats *stats_total);

/* mean -> NaN->0 -> sqrt -> 256*clamp(.,0,0.999) -> u8 — write_color
 * (include/core/integrator.cpp:289-308).  Host, needs no GPU. */
int rt_tonemap_u8(const float *sum_rgb, int32_t width, int32_t height, int32_t spp, uint8_t *out_rgb);

/* The other output format of the reference: the P3 (text) PPM its first-generation entry file writes — header
 * (RayTracing/main.cpp:106) and one line per pixel through that file's write_color (RayTracing/main.cpp:28-48, also
 * include/color.h), whose arithmetic differs from integrator.cpp's in where the division by spp happens.  Rows as stored
 * (row 0 first).  Host, needs no GPU.  0 = ok, -2 = the file cannot be written. */
int rt_write_ppm(const char *path, const float *sum_rgb, int32_t width, int32_t height, int32_t spp);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */

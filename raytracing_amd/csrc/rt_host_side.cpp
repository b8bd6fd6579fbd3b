// rt_host_side.cpp — the entry points of rt_amd.h that run on the host and need no GPU:
// rt_bvh_build, rt_prim_bounds, rt_tonemap_u8, rt_version.
#include "rt_amd.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

namespace {

// One primitive during the build: centroid for the split decisions, id to find its box again.
struct Item { float c[3]; int32_t id; };

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; ++k) { lo[k] = std::numeric_limits<float>::max(); hi[k] = std::numeric_limits<float>::lowest(); }
    }
    void grow(const float *p) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    void grow(const Box &b) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
};

// Emits the linear tree directly in depth-first pre-order (a node's first child is the
// next node), which is the layout flattenBVHTree produces (bvh.cpp:229-245).  The split
// rule is recursiveBuild's SplitMethod::Middle with its fall-through to EqualCounts
// (bvh.cpp:105-141); node boxes are unions of the children's boxes, which equals the
// reference's loop over the range because min/max never round.  Leaf slots are handed out
// in depth-first order.
// Split rules of BVHAccel::recursiveBuild (bvh.cpp:86-227).  The enum values are the reference's (bvh.h:17); HLBVH has no
// builder of its own there and falls into the SAH case of the switch (bvh.cpp:143-144).
struct TreeEmitter {
    std::vector<Item> items;
    const float *bounds;   // n*6
    rt_bvh_node *nodes;
    int32_t *order;
    int32_t nNodes = 0, nSlots = 0;
    int maxPrims = 1, method = RT_SPLIT_MIDDLE;

    static float area(const Box &b) {    // Bounds3::SurfaceArea, geometry.h:772-775
        float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
        return 2 * (dx * dy + dx * dz + dy * dz);
    }
    // bucket of a centroid along dim: 12 * Bounds3::Offset (geometry.h:794-800), truncated, last bucket closed (bvh.cpp:159-162)
    static int bucket(const Box &cb, const float *c, int dim) {
        float o = c[dim] - cb.lo[dim];
        if (cb.hi[dim] > cb.lo[dim]) o /= cb.hi[dim] - cb.lo[dim];
        int b = (int)(12 * o);
        return b == 12 ? 11 : b;
    }

    Box emit(int start, int end) {
        int32_t me = nNodes++;
        std::memset(&nodes[me], 0, sizeof(rt_bvh_node));
        int count = end - start;
        Box cb; cb.reset();
        for (int i = start; i < end; ++i) cb.grow(items[i].c);
        float dx = cb.hi[0] - cb.lo[0], dy = cb.hi[1] - cb.lo[1], dz = cb.hi[2] - cb.lo[2];
        int dim = (dx > dy && dx > dz) ? 0 : (dy > dz ? 1 : 2);   // Bounds3::MaximumExtent, geometry.h:780-788
        Box all; all.reset();                                      // bounds of the range (bvh.cpp:92-94)
        for (int i = start; i < end; ++i) { const float *b = bounds + 6 * (size_t)items[i].id; all.grow(b); all.grow(b + 3); }
        bool leaf = count == 1 || cb.hi[dim] == cb.lo[dim];
        int mid = (start + end) / 2;
        Item *first = items.data() + start, *last = items.data() + end;
        auto equalCounts = [&]() {
            mid = (start + end) / 2;
            std::nth_element(first, items.data() + mid, last, [dim](const Item &a, const Item &b) { return a.c[dim] < b.c[dim]; });
        };
        if (!leaf) {
            if (method == RT_SPLIT_MIDDLE) {
                float pmid = (cb.lo[dim] + cb.hi[dim]) / 2;
                Item *m = std::partition(first, last, [dim, pmid](const Item &it) { return it.c[dim] < pmid; });
                mid = (int)(m - items.data());
                if (mid == start || mid == end) equalCounts();     // the reference's case fall-through (bvh.cpp:131-133)
            } else if (method == RT_SPLIT_EQUAL_COUNTS) equalCounts();
            else if (count <= 2) equalCounts();                    // SAH, bvh.cpp:145-154
            else {
                int cnt[12] = {0}; Box bb[12];
                for (int k = 0; k < 12; ++k) bb[k].reset();
                for (int i = start; i < end; ++i) {
                    int b = bucket(cb, items[i].c, dim);
                    const float *pb = bounds + 6 * (size_t)items[i].id;
                    cnt[b]++; bb[b].grow(pb); bb[b].grow(pb + 3);
                }
                float cost[11];
                for (int i = 0; i < 11; ++i) {
                    Box b0, b1; b0.reset(); b1.reset();
                    int c0 = 0, c1 = 0;
                    for (int j = 0; j <= i; ++j) { b0.grow(bb[j]); c0 += cnt[j]; }
                    for (int j = i + 1; j < 12; ++j) { b1.grow(bb[j]); c1 += cnt[j]; }
                    cost[i] = 1 + (c0 * area(b0) + c1 * area(b1)) / area(all);
                }
                float minCost = cost[0]; int minBucket = 0;
                for (int i = 1; i < 11; ++i) if (cost[i] < minCost) { minCost = cost[i]; minBucket = i; }
                float leafCost = (float)count;
                if (count > maxPrims || minCost < leafCost) {
                    const Box cbc = cb;
                    Item *m = std::partition(first, last, [&cbc, dim, minBucket](const Item &it) { return bucket(cbc, it.c, dim) <= minBucket; });
                    mid = (int)(m - items.data());
                } else leaf = true;
            }
        }
        Box box;
        if (leaf) {
            box = all;
            nodes[me].offset = nSlots; nodes[me].nprim = (uint16_t)count;
            for (int i = start; i < end; ++i) order[nSlots++] = items[i].id;
        } else {
            box = emit(start, mid);
            int32_t second = nNodes;
            Box right = emit(mid, end);
            box.grow(right);                                       // InitInterior: Union of the children (bvh.cpp:30-34)
            nodes[me].offset = second; nodes[me].nprim = 0; nodes[me].axis = (uint8_t)dim;
        }
        for (int k = 0; k < 3; ++k) { nodes[me].bmin[k] = box.lo[k]; nodes[me].bmax[k] = box.hi[k]; }
        return box;
    }
};

inline void xf_point(const float *m, float x, float y, float z, float *out) {   // Transform::operator()(Point3f), transform.h:135-146
    float xp = m[0] * x + m[1] * y + m[2] * z + m[3];
    float yp = m[4] * x + m[5] * y + m[6] * z + m[7];
    float zp = m[8] * x + m[9] * y + m[10] * z + m[11];
    float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
    if (wp == 1) { out[0] = xp; out[1] = yp; out[2] = zp; }
    else { float inv = 1.f / wp; out[0] = inv * xp; out[1] = inv * yp; out[2] = inv * zp; }
}

}  // namespace

extern "C" {

const char *rt_version(void) { return "raytracing_amd 0.1 (gfx950)"; }

int rt_bvh_build_ex(int32_t n, const float *prim_bounds, int32_t max_prims_in_node, int32_t split_method, rt_bvh_node *nodes_out, int32_t *n_nodes_out,
                    int32_t *order_out) {
    if (n < 0 || (n > 0 && (!prim_bounds || !nodes_out || !order_out)) || !n_nodes_out) return -1;
    if (split_method < RT_SPLIT_SAH || split_method > RT_SPLIT_EQUAL_COUNTS || max_prims_in_node < 1) return -1;
    if (n == 0) { *n_nodes_out = 0; return 0; }
    TreeEmitter te;
    te.maxPrims = std::min(255, (int)max_prims_in_node);      // bvh.cpp:65
    te.method = split_method == RT_SPLIT_HLBVH ? RT_SPLIT_SAH : split_method;
    te.items.resize(n);
    for (int32_t i = 0; i < n; ++i) {
        const float *b = prim_bounds + 6 * (size_t)i;
        for (int k = 0; k < 3; ++k) te.items[i].c[k] = .5f * b[k] + .5f * b[3 + k];   // bvh.cpp:14
        te.items[i].id = i;
    }
    te.bounds = prim_bounds; te.nodes = nodes_out; te.order = order_out;
    te.emit(0, n);
    *n_nodes_out = te.nNodes;
    return 0;
}

int rt_bvh_build(int32_t n, const float *prim_bounds, rt_bvh_node *nodes_out, int32_t *n_nodes_out, int32_t *order_out) {
    return rt_bvh_build_ex(n, prim_bounds, 1, RT_SPLIT_MIDDLE, nodes_out, n_nodes_out, order_out);
}

int rt_prim_bounds(const rt_scene_desc *d, float *out) {
    if (!d || !out || d->n_prims < 0 || (d->n_prims > 0 && (!d->prim_kind || !d->prim_geom))) return -1;
    for (int32_t i = 0; i < d->n_prims; ++i) {       // nothing is read through an index that has not been checked
        const int g = d->prim_geom[i];
        if (d->prim_kind[i] == RT_PRIM_TRIANGLE ? (!d->tri_p || g < 0 || g >= d->n_tris)
                                                : (d->prim_kind[i] != RT_PRIM_SPHERE || !d->spheres || g < 0 || g >= d->n_spheres)) return -1;
    }
    for (int32_t i = 0; i < d->n_prims; ++i) {
        Box b; b.reset();
        if (d->prim_kind[i] == RT_PRIM_TRIANGLE) {
            const float *p = d->tri_p + 9 * (size_t)d->prim_geom[i];
            b.grow(p); b.grow(p + 3); b.grow(p + 6);
        } else {
            // Shape::WorldBound = ObjectToWorld(ObjectBound): the 8 transformed corners (transform.cpp:181-192)
            const rt_sphere &s = d->spheres[d->prim_geom[i]];
            float r = s.radius;
            for (int corner = 0; corner < 8; ++corner) {
                float q[3];
                xf_point(s.o2w, (corner & 1) ? r : -r, (corner & 2) ? r : -r, (corner & 4) ? r : -r, q);
                b.grow(q);
            }
        }
        for (int k = 0; k < 3; ++k) { out[6 * (size_t)i + k] = b.lo[k]; out[6 * (size_t)i + 3 + k] = b.hi[k]; }
    }
    return 0;
}

int rt_tonemap_u8(const float *sum_rgb, int32_t width, int32_t height, int32_t spp, uint8_t *out_rgb) {
    if (!sum_rgb || !out_rgb || width <= 0 || height <= 0 || spp <= 0) return -1;
    size_t n = (size_t)width * height * 3;
    float fspp = (float)spp;
    for (size_t i = 0; i < n; ++i) {
        float v = sum_rgb[i] / fspp;                 // colObj / samplesPerPixel, integrator.cpp:337
        if (v != v) v = 0.f;                         // NaN -> 0, integrator.cpp:294-296
        v = (float)std::sqrt((double)v);             // gamma 2 through a double sqrt, integrator.cpp:299-302
        float c = v < 0.f ? 0.f : (v > 0.999f ? 0.999f : v);
        out_rgb[i] = (uint8_t)(int)(256 * c);
    }
    return 0;
}

// The P3 text output of the reference's first-generation entry file: header "P3\nW H\n255\n" (main.cpp:106), then one
// "r g b\n" line per pixel through its write_color (main.cpp:28-48): NaN -> 0, scale = 1.0 / spp as a double,
// sqrt(scale * c) in double stored back into the float channel, 256 * clamp(c, 0, 0.999) with the clamp in double, truncated.
int rt_write_ppm(const char *path, const float *sum_rgb, int32_t width, int32_t height, int32_t spp) {
    if (!path || !sum_rgb || width <= 0 || height <= 0 || spp <= 0) return -1;
    FILE *f = fopen(path, "wb");
    if (!f) return -2;
    fprintf(f, "P3\n%d %d\n255\n", width, height);
    const double scale = 1.0 / spp;
    size_t n = (size_t)width * height;
    for (size_t p = 0; p < n; ++p) {
        int v[3];
        for (int c = 0; c < 3; ++c) {
            float r = sum_rgb[3 * p + c];
            if (r != r) r = 0.0;
            r = (float)std::sqrt(scale * r);
            double cl = r < 0.0 ? 0.0 : (r > 0.999 ? 0.999 : (double)r);
            if (cl != cl) cl = 0.0;      // the square root of a negative sum: the reference's cast of NaN to int is undefined; 0 here
            v[c] = static_cast<int>(256 * cl);
        }
        fprintf(f, "%d %d %d\n", v[0], v[1], v[2]);
    }
    return fclose(f) == 0 ? 0 : -2;
}

}  // extern "C"

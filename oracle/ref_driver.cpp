// ref_driver.cpp — drives the REFERENCE renderer's own classes (compiled from /root/reference by
// oracle/Makefile `ref`) on scenes dumped by this repository, and writes raw result arrays that
// tests/golden/make_golden.py turns into the committed golden fixtures.
//
// TEST INFRASTRUCTURE ONLY; exists only where /root/reference is mounted.  Everything geometric and
// radiometric below is computed by the reference's code: BVHAccel (bvh.cpp), Triangle / Sphere,
// the materials / BSDFs, StratifiedSampler, PerspectiveCamera, PathIntegrator::Li_re.  This file only
// constructs the objects from a flat scene dump and loops over rays / pixels.
//
// The pixel loop is the reference's render_pixel (integrator.cpp:317-339) restated without its
// thread pool: one process per band of rows, because `perfectspecular` (integrator.cpp:235) is a
// process-wide global that concurrent threads would race on.  Radiance is accumulated in float
// instead of being quantised to 8 bits.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

// The flattened tree and the reordered primitive list are private members of BVHAccel; the driver
// reads them to export the reference's tree.  bvh.cpp is compiled separately; LinearBVHNode is
// re-declared here with the layout of bvh.cpp:43-47 (checked against sizeof below).
#define private public
#include "core/bvh.h"
#include "core/sampling.h"
#include "core/mipmap.h"
#include "core/infinite.h"
#undef private
#include "core/camera.h"
#include "core/imagemap.h"
#include "core/integrator.h"
#include "core/interaction.h"
#include "core/light.h"
#include "core/material.h"
#include "core/microfacet.h"
#include "core/reflection.h"
#include "core/scene.h"
#include "core/sphere.h"
#include "core/stratified.h"
#include "core/texture.h"
#include "core/triangle.h"

#include "rt_amd.h"

extern bool perfectspecular;   // integrator.cpp:235

struct LinearBVHNode {
    Bounds3f bounds;
    union { int primitivesOffset; int secondChildOffset; };
    uint16_t nPrimitives; uint8_t axis; uint8_t pad[1];
};
static_assert(sizeof(LinearBVHNode) == 32, "LinearBVHNode layout");

// The reference defines MicrofacetTransmission (reflection.cpp:311-355) but none of its materials adds it to a BSDF.  This
// material — the driver's, built from the reference's own BxDF classes the way pbrt's glass assembles them — does, so that the
// lobe can be exercised through the reference's BSDF / integrator code.
class RoughGlassTestMaterial : public Material {
public:
    RoughGlassTestMaterial(const std::shared_ptr<Texture<Spectrum>> &Kr, const std::shared_ptr<Texture<Spectrum>> &Kt, const std::shared_ptr<Texture<Float>> &index,
                           const std::shared_ptr<Texture<Float>> &rough, bool remap) : Kr(Kr), Kt(Kt), index(index), rough(rough), remap(remap) {}
    void ComputeScatteringFunctions(SurfaceInteraction *si, TransportMode mode, bool allowMultipleLobes) const override {
        Float eta = index->Evaluate(*si), r = rough->Evaluate(*si);
        Spectrum R = Kr->Evaluate(*si).Clamp(), T = Kt->Evaluate(*si).Clamp();
        si->bsdf = new BSDF(*si, eta);
        if (R.IsBlack() && T.IsBlack()) return;
        if (remap) r = TrowbridgeReitzDistribution::RoughnessToAlpha(r);
        if (!R.IsBlack()) si->bsdf->Add(new MicrofacetReflection(R, new TrowbridgeReitzDistribution(r, r), new FresnelDielectric(1.f, eta)));
        if (!T.IsBlack()) si->bsdf->Add(new MicrofacetTransmission(T, new TrowbridgeReitzDistribution(r, r), 1.f, eta, mode));
    }
private:
    std::shared_ptr<Texture<Spectrum>> Kr, Kt; std::shared_ptr<Texture<Float>> index, rough; bool remap;
};

// The reference defines UniformSampleAllLights (integrator.cpp:26-51) but no integrator of it calls the function.  These two
// estimators are the driver's: the control flow of PathIntegrator::Li_re / Li around the reference's own UniformSampleAllLights,
// BSDF, light and scene code, so that the function can be exercised inside whole paths.
struct AllLightsEstimator {
    int maxDepth; Float rrThreshold; std::vector<int> one;
    AllLightsEstimator(int maxDepth, Float rr, const Scene &scene) : maxDepth(maxDepth), rrThreshold(rr), one(scene.lights.size(), 1) {}
    // recursive form, emission ends the path (the shape of Li_re, integrator.cpp:236-281)
    Spectrum Recursive(const Ray &ray, const Scene &scene, Sampler &sampler, int depth, bool afterSpecular) const {
        Spectrum L(0.f);
        SurfaceInteraction isect;
        bool hit = scene.Intersect(ray, &isect);
        if (depth == 0 || afterSpecular || scene.lights.size() == 0) {
            if (hit) { L += isect.Le(-ray.d); if (!L.IsBlack()) return L; }
            else if (scene.lights.size() == 0) {
                Vector3f u = Normalize(ray.d);
                auto t = 0.5 * (u[1] + 1.0);
                L += (1.0 - t) * Spectrum(1.0, 1.0, 1.0) + t * Spectrum(0.5, 0.7, 1.0);
            }
        }
        if (!hit || depth >= maxDepth) return L;
        isect.ComputeScatteringFunctions(ray, true);
        if (isect.bsdf->NumComponents(BxDFType(BSDF_ALL & ~BSDF_SPECULAR)) > 0) L += UniformSampleAllLights(isect, scene, sampler, one, false);
        Vector3f wo = -ray.d, wi; Float pdf; BxDFType flags;
        Spectrum f = isect.bsdf->Sample_f(wo, &wi, sampler.Get2D(), &pdf, BSDF_ALL, &flags);
        if (f.IsBlack() || pdf == 0.f) return L;
        Spectrum beta = Spectrum(1.f) * (f * AbsDot(wi, isect.shading.n) / pdf);
        Ray next = isect.SpawnRay(wi);
        if (depth > 3) {
            Float q = std::max((Float).05, 1 - beta.MaxComponentValue());
            if (sampler.Get1D() < q) return L;
            beta /= 1 - q;
        }
        return L + beta * Recursive(next, scene, sampler, depth + 1, (flags & BSDF_SPECULAR) != 0);
    }
    // iterative form with path throughput (the shape of Li, integrator.cpp:175-229)
    Spectrum Iterative(const Ray &r, const Scene &scene, Sampler &sampler) const {
        Spectrum L(0.f), beta(1.f);
        Ray ray(r);
        bool afterSpecular = false;
        Float etaScale = 1;
        for (int bounces = 0;; ++bounces) {
            SurfaceInteraction isect;
            bool hit = scene.Intersect(ray, &isect);
            if (bounces == 0 || afterSpecular) {
                if (hit) L += beta * isect.Le(-ray.d);
                else for (const auto &light : scene.infiniteLights) L += beta * light->Le(ray);
            }
            if (!hit || bounces >= maxDepth) break;
            isect.ComputeScatteringFunctions(ray, true);
            if (isect.bsdf->NumComponents(BxDFType(BSDF_ALL & ~BSDF_SPECULAR)) > 0) L += beta * UniformSampleAllLights(isect, scene, sampler, one, false);
            Vector3f wo = -ray.d, wi; Float pdf; BxDFType flags;
            Spectrum f = isect.bsdf->Sample_f(wo, &wi, sampler.Get2D(), &pdf, BSDF_ALL, &flags);
            if (f.IsBlack() || pdf == 0.f) break;
            beta *= f * AbsDot(wi, isect.shading.n) / pdf;
            afterSpecular = (flags & BSDF_SPECULAR) != 0;
            if ((flags & BSDF_SPECULAR) && (flags & BSDF_TRANSMISSION)) {
                Float eta = isect.bsdf->eta;
                etaScale *= (Dot(wo, isect.n) > 0) ? (eta * eta) : 1 / (eta * eta);
            }
            ray = isect.SpawnRay(wi);
            Spectrum rrBeta = beta * etaScale;
            if (rrBeta.MaxComponentValue() < rrThreshold && bounces > 3) {
                Float q = std::max((Float).05, 1 - rrBeta.MaxComponentValue());
                if (sampler.Get1D() < q) break;
                beta /= 1 - q;
            }
        }
        return L;
    }
};

// ------------------------------------------------------------- scene dump --
struct Reader {
    std::vector<uint8_t> buf; size_t p = 0;
    explicit Reader(const std::string &path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
        buf.assign((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    }
    template <typename T> T get() { T v; memcpy(&v, &buf[p], sizeof(T)); p += sizeof(T); return v; }
    template <typename T> std::vector<T> arr(size_t n) { std::vector<T> v(n); if (n) memcpy(v.data(), &buf[p], n * sizeof(T)); p += n * sizeof(T); return v; }
    std::string str() { int32_t n = get<int32_t>(); std::string s((const char *)&buf[p], n); p += n; return s; }
};

struct RefScene {
    std::vector<std::shared_ptr<Primitive>> prims;          // original order
    std::map<const Primitive *, int> primIndex;
    std::vector<std::shared_ptr<Light>> lights;
    std::shared_ptr<BVHAccel> bvh;
    std::shared_ptr<Scene> scene;
    std::vector<std::unique_ptr<Transform>> xf;
};

static Transform *Keep(RefScene &rs, const Transform &t) { rs.xf.emplace_back(new Transform(t)); return rs.xf.back().get(); }

static void LoadScene(const std::string &path, RefScene &rs) {
    Reader r(path);
    if (r.get<uint32_t>() != 0x43535452u) { fprintf(stderr, "bad scene dump\n"); exit(2); }
    int n_prims = r.get<int32_t>(), n_tris = r.get<int32_t>(), n_spheres = r.get<int32_t>(), n_mat = r.get<int32_t>(), n_tex = r.get<int32_t>(), n_lights = r.get<int32_t>();
    auto kind = r.arr<uint8_t>(n_prims); auto geom = r.arr<int32_t>(n_prims); auto pmat = r.arr<int32_t>(n_prims); auto plight = r.arr<int32_t>(n_prims);
    auto tri_p = r.arr<float>(9 * (size_t)n_tris); auto tri_uv = r.arr<float>(6 * (size_t)n_tris);
    int has_n = r.get<int32_t>();
    std::vector<float> tri_n; std::vector<uint8_t> tri_has_n;
    if (has_n) { tri_n = r.arr<float>(9 * (size_t)n_tris); tri_has_n = r.arr<uint8_t>(n_tris); }
    auto spheres = r.arr<float>(33 * (size_t)n_spheres);
    auto mats = r.arr<int32_t>(7 * (size_t)n_mat);
    std::vector<std::shared_ptr<Texture<Spectrum>>> texS(n_tex); std::vector<std::shared_ptr<Texture<Float>>> texF(n_tex);
    for (int i = 0; i < n_tex; ++i) {
        int k = r.get<int32_t>(); r.get<int32_t>(); r.get<int32_t>();
        float v[3]; for (int c = 0; c < 3; ++c) v[c] = r.get<float>();
        float su = r.get<float>(), sv = r.get<float>(), du = r.get<float>(), dv = r.get<float>();
        std::string png = r.str();
        if (k == RT_TEX_CONST) { texS[i] = std::make_shared<ConstantTexture<Spectrum>>(Spectrum(v[0], v[1], v[2])); texF[i] = std::make_shared<ConstantTexture<Float>>(v[0]); }
        else texS[i] = std::make_shared<ImageTexture<RGBSpectrum, Spectrum>>(std::make_unique<UVMapping2D>(su, sv, du, dv), png, false, 8.f, ImageWrap::Repeat, 1.f, false);
    }
    std::vector<std::shared_ptr<Material>> materials(n_mat);
    for (int i = 0; i < n_mat; ++i) {
        const int32_t *m = &mats[7 * (size_t)i]; const int32_t *t = m + 1; bool remap = m[5] != 0;
        switch (m[0]) {
        case RT_MAT_MATTE: materials[i] = std::make_shared<MatteMaterial>(texS[t[0]], std::make_shared<ConstantTexture<Float>>(0.f)); break;
        case RT_MAT_MIRROR: materials[i] = std::make_shared<MirrorMaterial>(texS[t[0]]); break;
        case RT_MAT_GLASS: materials[i] = std::make_shared<GlassMaterial>(texS[t[0]], texS[t[1]], texF[t[2]]); break;
        case RT_MAT_METAL: materials[i] = std::make_shared<MetalMaterial>(texS[t[0]], texS[t[1]], texF[t[2]], texF[t[2]], texF[t[3]], remap); break;
        case RT_MAT_PLASTIC: materials[i] = std::make_shared<PlasticMaterial>(texS[t[0]], texS[t[1]], texF[t[2]], remap); break;
        case RT_MAT_ROUGHGLASS: materials[i] = std::make_shared<RoughGlassTestMaterial>(texS[t[0]], texS[t[1]], texF[t[2]], texF[t[3]], remap); break;
        case RT_MAT_METALROUGH: materials[i] = std::make_shared<MetalRoughnessMaterial>(texS[t[0]], t[1] >= 0 ? texS[t[1]] : nullptr); break;
        default: fprintf(stderr, "bad material\n"); exit(2);
        }
    }
    struct LightRec { int32_t prim; float Le[3]; int32_t two; float area; int32_t flip; int32_t kind; int32_t env; };
    auto lrec = r.arr<LightRec>(n_lights);
    rs.lights.resize(n_lights);
    // infinite area lights: transform, scale and the .hdr file; everything else is the reference constructor's doing (infinite.cpp:4-71)
    int n_env = r.get<int32_t>();
    struct EnvRec { float m[16], mi[16], L[3]; std::string path; };
    std::vector<EnvRec> envs(n_env);
    for (EnvRec &e : envs) { for (float &v : e.m) v = r.get<float>(); for (float &v : e.mi) v = r.get<float>(); for (float &v : e.L) v = r.get<float>(); e.path = r.str(); }
    for (int i = 0; i < n_lights; ++i)
        if (lrec[i].kind == RT_LIGHT_INFINITE) {
            const EnvRec &e = envs[lrec[i].env];
            Float m[4][4], mi[4][4]; memcpy(m, e.m, 64); memcpy(mi, e.mi, 64);
            rs.lights[i] = std::make_shared<InfiniteAreaLight>(Transform(Matrix4x4(m), Matrix4x4(mi)), Spectrum(e.L[0], e.L[1], e.L[2]), 1, e.path);
        }
    Transform *I = Keep(rs, Transform());
    for (int i = 0; i < n_prims; ++i) {
        std::shared_ptr<Shape> shape;
        if (kind[i] == RT_PRIM_TRIANGLE) {
            size_t g = geom[i]; int idx[3] = {0, 1, 2};
            Point3f P[3]; Point2f UV[3]; Normal3f N[3];
            for (int k = 0; k < 3; ++k) { P[k] = Point3f(tri_p[9 * g + 3 * k], tri_p[9 * g + 3 * k + 1], tri_p[9 * g + 3 * k + 2]); UV[k] = Point2f(tri_uv[6 * g + 2 * k], tri_uv[6 * g + 2 * k + 1]); }
            bool hn = has_n && tri_has_n[g];
            if (hn) for (int k = 0; k < 3; ++k) N[k] = Normal3f(tri_n[9 * g + 3 * k], tri_n[9 * g + 3 * k + 1], tri_n[9 * g + 3 * k + 2]);
            // the dump's transforms are the identity, so reverseOrientation alone carries the light's normal flip (triangle.cpp:242-243)
            bool rev = plight[i] >= 0 && lrec[plight[i]].flip != 0;
            shape = CreateTriangleMesh(I, I, rev, 1, idx, 3, P, nullptr, hn ? N : nullptr, UV)[0];
        } else {
            const float *s = &spheres[33 * (size_t)geom[i]];
            Float m[4][4], mi[4][4]; memcpy(m, s + 1, 64); memcpy(mi, s + 17, 64);
            Transform *o2w = Keep(rs, Transform(Matrix4x4(m), Matrix4x4(mi))), *w2o = Keep(rs, Transform(Matrix4x4(mi), Matrix4x4(m)));
            shape = std::make_shared<Sphere>(o2w, w2o, false, s[0]);
        }
        std::shared_ptr<AreaLight> area;
        if (plight[i] >= 0) {
            const LightRec &L = lrec[plight[i]];
            area = std::make_shared<DiffuseAreaLight>(*I, Spectrum(L.Le[0], L.Le[1], L.Le[2]), 1, shape, L.two != 0);
            rs.lights[plight[i]] = area;
        }
        auto gp = std::make_shared<GeometricPrimitive>(shape, materials[pmat[i]], area);
        rs.primIndex[gp.get()] = i;
        rs.prims.push_back(gp);
    }
    // REF_BVH_MAXPRIMS / REF_BVH_SPLIT select the other constructor arguments (bvh.h:17-20); unset = the reference's defaults
    const char *mp = getenv("REF_BVH_MAXPRIMS"), *sm = getenv("REF_BVH_SPLIT");
    rs.bvh = std::make_shared<BVHAccel>(rs.prims, mp ? atoi(mp) : 1, sm ? (BVHAccel::SplitMethod)atoi(sm) : BVHAccel::SplitMethod::Middle);
    rs.scene = std::make_shared<Scene>(rs.bvh, rs.lights);
}

static void WriteAll(const std::string &path, const void *p, size_t bytes) {
    FILE *f = fopen(path.c_str(), "wb"); if (!f) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(2); }
    fwrite(p, 1, bytes, f); fclose(f);
}

struct CamSpec { float eye[3], look[3], up[3], fov, nearPlane; int32_t W, H; };
static std::shared_ptr<Camera> MakeCamera(const CamSpec &c) {
    Transform lookat = LookAt(Point3f(c.eye[0], c.eye[1], c.eye[2]), Point3f(c.look[0], c.look[1], c.look[2]), Vector3f(c.up[0], c.up[1], c.up[2]));
    Transform c2w = Inverse(lookat);
    return std::shared_ptr<Camera>(CreatePerspectiveCamera(c2w, c.nearPlane, c.fov, c.W, c.H));
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: ref_driver <bvh|trace|render|camera|transform|probe> ...\n"); return 1; }
    std::string cmd = argv[1];
    if (cmd == "bvh") {          // bvh scene.bin out.bin : [n_nodes][nodes 32B each][order i32 * n_prims]
        RefScene rs; LoadScene(argv[2], rs);
        // count nodes by walking the flattened tree
        const LinearBVHNode *nodes = (const LinearBVHNode *)rs.bvh->nodes;
        int n = (int)rs.prims.size(), total = 2 * n - 1;   // one primitive per leaf unless centroids coincide; recount below
        {   // exact count: depth-first walk
            std::vector<int> stack{0}; int maxIdx = 0;
            while (!stack.empty()) { int i = stack.back(); stack.pop_back(); maxIdx = std::max(maxIdx, i);
                if (nodes[i].nPrimitives == 0) { stack.push_back(i + 1); stack.push_back(nodes[i].secondChildOffset); } }
            total = maxIdx + 1;
        }
        std::vector<int32_t> order(n);
        for (int i = 0; i < n; ++i) order[i] = rs.primIndex[rs.bvh->primitives[i].get()];
        std::string out; out.resize(4 + 32 * (size_t)total + 4 * (size_t)n);
        memcpy(&out[0], &total, 4); memcpy(&out[4], nodes, 32 * (size_t)total); memcpy(&out[4 + 32 * (size_t)total], order.data(), 4 * (size_t)n);
        WriteAll(argv[3], out.data(), out.size());
        return 0;
    }
    if (cmd == "env") {          // env scene.bin out.bin : tables of the scene's first InfiniteAreaLight as its constructor and Preprocess leave them
        RefScene rs; LoadScene(argv[2], rs);
        const InfiniteAreaLight *env = nullptr;
        for (auto &l : rs.lights) if (!env) env = dynamic_cast<const InfiniteAreaLight *>(l.get());
        if (!env) { fprintf(stderr, "no infinite light\n"); return 2; }
        int w = env->Lmap->Width(), h = env->Lmap->Height();
        const Distribution2D &d2 = *env->distribution;
        int nv = d2.pMarginal->Count(), nu = d2.pConditionalV[0]->Count();
        std::vector<float> out;
        auto put = [&](float v) { out.push_back(v); };
        put((float)w); put((float)h); put((float)nu); put((float)nv);
        for (int t = 0; t < h; ++t) for (int s2 = 0; s2 < w; ++s2) { const RGBSpectrum &c = env->Lmap->Texel(0, s2, t); put(c[0]); put(c[1]); put(c[2]); }
        for (int v = 0; v < nv; ++v) for (float f : d2.pConditionalV[v]->func) put(f);
        for (int v = 0; v < nv; ++v) for (float f : d2.pConditionalV[v]->cdf) put(f);
        for (int v = 0; v < nv; ++v) put(d2.pConditionalV[v]->funcInt);
        for (float f : d2.pMarginal->cdf) put(f);
        put(d2.pMarginal->funcInt);
        put(env->worldCenter.x); put(env->worldCenter.y); put(env->worldCenter.z); put(env->worldRadius);
        // a few Sample_Li known answers: (ref point, u) -> wi, pdf, Li
        RNG rng(17);
        for (int i = 0; i < 256; ++i) {
            Point2f u(rng.UniformFloat(), rng.UniformFloat());
            Interaction ref(Point3f(rng.UniformFloat() * 4, rng.UniformFloat() * 4, rng.UniformFloat() * 4), 0.f);
            Vector3f wi; Float pdf = 0; VisibilityTester vis;
            Spectrum Li = env->Sample_Li(ref, u, &wi, &pdf, &vis);
            put(u[0]); put(u[1]); put(ref.p.x); put(ref.p.y); put(ref.p.z); put(wi.x); put(wi.y); put(wi.z); put(pdf); put(Li[0]); put(Li[1]); put(Li[2]);
            put(vis.P1().p.x); put(vis.P1().p.y); put(vis.P1().p.z);
        }
        WriteAll(argv[3], out.data(), out.size() * 4);
        return 0;
    }
    if (cmd == "trace") {        // trace scene.bin rays.bin(n x 7 f32) out.bin : per ray [prim i32][t f32][u f32][v f32][any u32]
        RefScene rs; LoadScene(argv[2], rs);
        Reader rr(argv[3]); size_t n = rr.buf.size() / 28; auto rays = rr.arr<float>(7 * n);
        struct Out { int32_t prim; float t, u, v; uint32_t any; };
        std::vector<Out> out(n);
        for (size_t i = 0; i < n; ++i) {
            const float *q = &rays[7 * i];
            Ray ray(Point3f(q[0], q[1], q[2]), Vector3f(q[3], q[4], q[5]), q[6]);
            SurfaceInteraction isect;
            bool hit = rs.scene->Intersect(ray, &isect);
            out[i].prim = hit ? rs.primIndex[isect.primitive] : -1; out[i].t = ray.tMax; out[i].u = hit ? isect.uv[0] : 0; out[i].v = hit ? isect.uv[1] : 0;
            Ray ray2(Point3f(q[0], q[1], q[2]), Vector3f(q[3], q[4], q[5]), q[6]);
            out[i].any = rs.scene->IntersectP(ray2) ? 1 : 0;
            isect.bsdf = nullptr;
        }
        WriteAll(argv[4], out.data(), out.size() * sizeof(Out));
        return 0;
    }
    if (cmd == "camera") {       // camera cam.bin film.bin(n x 2 f32) out.bin : per sample o[3] d[3]
        Reader cr(argv[2]); CamSpec c = cr.get<CamSpec>();
        auto cam = MakeCamera(c);
        Reader fr(argv[3]); size_t n = fr.buf.size() / 8; auto film = fr.arr<float>(2 * n);
        std::vector<float> out(6 * n);
        for (size_t i = 0; i < n; ++i) {
            CameraSample cs; cs.pFilm = Point2f(film[2 * i], film[2 * i + 1]); cs.time = 0; cs.pLens = Point2f(0, 0);
            Ray r; cam->GenerateRay(cs, &r);
            out[6 * i] = r.o.x; out[6 * i + 1] = r.o.y; out[6 * i + 2] = r.o.z; out[6 * i + 3] = r.d.x; out[6 * i + 4] = r.d.y; out[6 * i + 5] = r.d.z;
        }
        WriteAll(argv[4], out.data(), out.size() * 4);
        return 0;
    }
    if (cmd == "transform") {    // transform ops.bin([n_ops i32][kinds i32*n][args f32*3n][n_pts i32][pts f32*3n]) out.bin : pts, m[16], minv[16]
        Reader r(argv[2]); int n_ops = r.get<int32_t>(); auto kinds = r.arr<int32_t>(n_ops); auto args = r.arr<float>(3 * (size_t)n_ops);
        int n_pts = r.get<int32_t>(); auto pts = r.arr<float>(3 * (size_t)n_pts);
        Transform t;
        for (int i = 0; i < n_ops; ++i) {
            const float *a = &args[3 * (size_t)i]; Transform o;
            switch (kinds[i]) { case 0: o = Translate(Vector3f(a[0], a[1], a[2])); break; case 1: o = RotateX(a[0]); break; case 2: o = RotateY(a[0]); break;
                                case 3: o = RotateZ(a[0]); break; case 4: o = Scale(a[0], a[1], a[2]); break; }
            t = (i == 0) ? o : t * o;
        }
        std::vector<float> out(3 * (size_t)n_pts + 32);
        for (int i = 0; i < n_pts; ++i) { Point3f p = t(Point3f(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2])); out[3 * i] = p.x; out[3 * i + 1] = p.y; out[3 * i + 2] = p.z; }
        memcpy(&out[3 * (size_t)n_pts], t.GetMatrix().m, 64); memcpy(&out[3 * (size_t)n_pts + 16], t.GetInverseMatrix().m, 64);
        WriteAll(argv[3], out.data(), out.size() * 4);
        return 0;
    }
    if (cmd == "probe") {        // probe scene.bin probes.bin(n x [o3 d3 wi3 u0 u1] f32) out.bin : per probe 26 floats
        RefScene rs; LoadScene(argv[2], rs);
        Reader pr(argv[3]); size_t n = pr.buf.size() / 44; auto q = pr.arr<float>(11 * n);
        std::vector<float> out(26 * n, 0.f);
        for (size_t i = 0; i < n; ++i) {
            const float *a = &q[11 * i]; float *o = &out[26 * i];
            Ray ray(Point3f(a[0], a[1], a[2]), Vector3f(a[3], a[4], a[5]));
            SurfaceInteraction isect;
            if (!rs.scene->Intersect(ray, &isect)) { o[0] = -1; continue; }
            o[0] = (float)rs.primIndex[isect.primitive];
            o[1] = isect.p.x; o[2] = isect.p.y; o[3] = isect.p.z; o[4] = isect.n.x; o[5] = isect.n.y; o[6] = isect.n.z;
            o[7] = isect.wo.x; o[8] = isect.wo.y; o[9] = isect.wo.z; o[10] = isect.uv[0]; o[11] = isect.uv[1];
            o[12] = isect.shading.dpdu.x; o[13] = isect.shading.dpdu.y; o[14] = isect.shading.dpdu.z;
            isect.ComputeScatteringFunctions(ray, true);
            Spectrum f = isect.bsdf->f(isect.wo, Vector3f(a[6], a[7], a[8]), BxDFType(BSDF_ALL & ~BSDF_SPECULAR));
            Vector3f wi; Float pdf = 0; BxDFType flags = BxDFType(0);
            Spectrum fs = isect.bsdf->Sample_f(-ray.d, &wi, Point2f(a[9], a[10]), &pdf, BSDF_ALL, &flags);
            o[15] = f[0]; o[16] = f[1]; o[17] = f[2]; o[18] = wi.x; o[19] = wi.y; o[20] = wi.z; o[21] = fs[0]; o[22] = fs[1]; o[23] = fs[2]; o[24] = pdf; o[25] = (float)(int)flags;
        }
        WriteAll(argv[4], out.data(), out.size() * 4);
        return 0;
    }
    if (cmd == "render") {       // render scene.bin cam.bin sx sy max_depth seed_offset nproc out.f32 [estimator [rrThreshold]]  (sum of Li per pixel, W*H*3)
                                 // estimator: 0 = PathIntegrator::Li_re (what the reference's Render calls), 1 = PathIntegrator::Li,
                                 // 2 / 3 = the driver's AllLightsEstimator in its recursive / iterative form
        RefScene rs; LoadScene(argv[2], rs);
        Reader cr(argv[3]); CamSpec c = cr.get<CamSpec>();
        int sx = atoi(argv[4]), sy = atoi(argv[5]), maxDepth = atoi(argv[6]); long long seedOff = atoll(argv[7]); int nproc = std::max(1, atoi(argv[8]));
        auto cam = MakeCamera(c);
        std::shared_ptr<Sampler> ss = std::make_shared<StratifiedSampler>(sx, sy, true, 1);
        Bounds2i pixelBounds;
        const int estimator = argc > 10 ? atoi(argv[10]) : 0;
        const Float rrThreshold = argc > 11 ? (Float)atof(argv[11]) : 1.f;
        PathIntegrator integ(maxDepth, cam, ss, pixelBounds, rrThreshold);
        integ.Preprocess(*rs.scene, *ss);
        AllLightsEstimator all(maxDepth, rrThreshold, *rs.scene);
        int W = c.W, H = c.H;
        size_t bytes = (size_t)W * H * 3 * sizeof(float);
        float *img = (float *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
        std::vector<pid_t> kids;
        for (int k = 0; k < nproc; ++k) {
            pid_t pid = nproc > 1 ? fork() : 0;
            if (pid == 0) {
                for (int j = k; j < H; j += nproc)
                    for (int i = 0; i < W; ++i) {
                        std::unique_ptr<Sampler> ps = ss->Clone((int)(i + j * W + seedOff));
                        Point2i pixel(i, j);
                        ps->StartPixel(pixel);
                        Spectrum col(0.0);
                        do {
                            Ray r; CameraSample cs = ps->GetCameraSample(pixel);
                            cam->GenerateRay(cs, &r);
                            perfectspecular = false;
                            if (estimator == 0) col += integ.Li_re(r, *rs.scene, *ps, 0);
                            else if (estimator == 1) col += integ.Li(r, *rs.scene, *ps, 0);
                            else if (estimator == 2) col += all.Recursive(r, *rs.scene, *ps, 0, false);
                            else col += all.Iterative(r, *rs.scene, *ps);
                        } while (ps->StartNextSample());
                        float *px = img + 3 * ((size_t)j * W + i);
                        px[0] = col[0]; px[1] = col[1]; px[2] = col[2];
                    }
                if (nproc > 1) _exit(0);
            } else kids.push_back(pid);
        }
        for (pid_t p : kids) { int st; waitpid(p, &st, 0); }
        WriteAll(argv[9], img, bytes);
        return 0;
    }
    fprintf(stderr, "unknown command %s\n", cmd.c_str());
    return 1;
}

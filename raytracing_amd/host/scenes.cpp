// scenes.cpp — the scenes of the reference's entry file, written against the host surface
// of rt_host.h, plus a C API so tests and benchmarks can obtain flattened scenes.
//
// Scene contents follow the reference's scene functions (RayTracing/RayTracing.cpp):
//   "cornell"   Cornellbox()      :46-192
//   "sphere1"   sphere1()         :344-408
//   "helmet"    DamagedHelmet()   :218-316   ("helmet_grey": the same with a room of albedo 0.8, a test scene)
//   "mrspheres" MetalRoughSphere() :194-216 (the model's geometry buffer is absent from the reference; see BuildMRSpheres)
//   "spheres2"  sphere1() plus a GlassMaterial sphere and an earth-textured sphere
//               (the Lambertian / metal / dielectric + earthmap configuration)
//   "cornell_metal", "spheres3"  test scenes for branches the shipped scenes never take (see BuildCornellMetal, BuildSpheres)
//   "soup"      N random triangles in a 100^3 box (BVH traversal stress; no reference counterpart)
#include "rt_host.h"

#include <stdexcept>

namespace {

struct BuiltScene {
    std::shared_ptr<Scene> scene;        // object graph (empty for "soup")
    std::unique_ptr<FlatScene> flat;
    Transform lookat;
    Float fov = 40.0f, nearPlane = 0.01f;
    std::vector<std::unique_ptr<Transform>> xforms;   // shapes keep raw Transform pointers
    std::vector<std::unique_ptr<ModelLoad>> loaders;
    const Transform *Keep(const Transform &t) { xforms.emplace_back(new Transform(t)); return xforms.back().get(); }
};

std::string g_error;

std::shared_ptr<Texture<Spectrum>> Const(const Spectrum &s) { return make_shared<ConstantTexture<Spectrum>>(s); }
std::shared_ptr<Texture<Float>> ConstF(Float v) { return make_shared<ConstantTexture<Float>>(v); }
std::shared_ptr<Material> Matte(const Spectrum &s) { return make_shared<MatteMaterial>(Const(s), ConstF(0.0f)); }

// The reference writes its boxes as 12 explicit triangles.  Corner code per vertex:
// x,z in {0,1} times `side`; y in {0, T (yTop), S (ySide)}.  Faces in the reference's order:
// y=0, y=yTop, z=0, x=0, x=side, z=side (RayTracing.cpp:99-109,136-149,163-176).
enum { Y0, YT, YS };
const unsigned char kBoxPattern[12][3][3] = {
    {{0, Y0, 1}, {1, Y0, 1}, {0, Y0, 0}}, {{1, Y0, 1}, {1, Y0, 0}, {0, Y0, 0}},
    {{0, YT, 1}, {0, YT, 0}, {1, YT, 1}}, {{1, YT, 1}, {0, YT, 0}, {1, YT, 0}},
    {{0, Y0, 0}, {1, Y0, 0}, {1, YS, 0}}, {{0, Y0, 0}, {1, YS, 0}, {0, YS, 0}},
    {{0, Y0, 0}, {0, YS, 1}, {0, Y0, 1}}, {{0, Y0, 0}, {0, YS, 0}, {0, YS, 1}},
    {{1, Y0, 0}, {1, YS, 1}, {1, Y0, 1}}, {{1, Y0, 0}, {1, YS, 0}, {1, YS, 1}},
    {{0, Y0, 1}, {1, Y0, 1}, {1, YS, 1}}, {{0, Y0, 1}, {1, YS, 1}, {0, YS, 1}},
};
std::vector<std::shared_ptr<Shape>> BoxTriangles(BuiltScene &bs, const Transform &o2w, int nTris, Float side, Float yTop, Float ySide) {
    std::vector<Point3f> P; std::vector<int> idx;
    for (int t = 0; t < nTris; ++t)
        for (int k = 0; k < 3; ++k) {
            const unsigned char *c = kBoxPattern[t][k];
            Float y = c[1] == Y0 ? 0.f : (c[1] == YT ? yTop : ySide);
            P.push_back(Point3f(c[0] ? side : 0.f, y, c[2] ? side : 0.f));
            idx.push_back((int)idx.size());
        }
    const Transform *po = bs.Keep(o2w), *pi = bs.Keep(Inverse(o2w));
    return CreateTriangleMesh(po, pi, false, nTris, idx.data(), (int)P.size(), P.data(), nullptr, nullptr, nullptr, nullptr);
}
// The 1x1 ceiling light: two triangles at height y (RayTracing.cpp:73-90, 247-264).
void AddCeilingLight(BuiltScene &bs, std::vector<std::shared_ptr<Primitive>> &prims, std::vector<std::shared_ptr<Light>> &lights,
                     const std::shared_ptr<Material> &mat, Float emit) {
    float length_Floor = 5;
    const float y = length_Floor - 0.01;
    float len = 0.5;
    Point3f P[6] = {Point3f(-len, y, len), Point3f(-len, y, -len), Point3f(len, y, len), Point3f(len, y, len), Point3f(-len, y, -len), Point3f(len, y, -len)};
    int idx[6] = {0, 1, 2, 3, 4, 5};
    Transform o2w = Translate(Vector3f(2.5f, 0.0f, 2.5f)) * Transform();
    const Transform *po = bs.Keep(o2w), *pi = bs.Keep(Inverse(o2w));
    auto tris = CreateTriangleMesh(po, pi, false, 2, idx, 6, P, nullptr, nullptr, nullptr, nullptr);
    for (int i = 0; i < 2; ++i) {
        std::shared_ptr<AreaLight> area = make_shared<DiffuseAreaLight>(*po, Spectrum(emit), 1, tris[i], false);
        lights.push_back(area);
        prims.push_back(make_shared<GeometricPrimitive>(tris[i], mat, area));
    }
}

// plastic: the short cube gets a PlasticMaterial (not in the reference's scene; exercises the two-lobe BSDF of material.cpp:32-49)
// env: an InfiniteAreaLight (include/core/infinite.h) joins the two ceiling triangles in scene.lights; the room's front is open, so
// its light reaches the inside (no shipped scene constructs one)
// roughGlass: the short cube is a rough dielectric (RoughGlassMaterial: the reference's MicrofacetTransmission lobe, which none of its
// own materials uses, next to a dielectric-Fresnel MicrofacetReflection)
void BuildCornell(BuiltScene &bs, bool plastic = false, const std::string &envMap = "", bool roughGlass = false) {
    std::vector<std::shared_ptr<Primitive>> prims; std::vector<std::shared_ptr<Light>> lights;
    AddCeilingLight(bs, prims, lights, Matte(Spectrum(1.0f, 1.0f, 1.0f)), 20.f);
    std::shared_ptr<Material> mirror = make_shared<MirrorMaterial>(Const(Spectrum(1.f)));
    std::shared_ptr<Material> mgreen = Matte(Spectrum(.12f, .45f, .15f)), mred = Matte(Spectrum(.65f, .05f, .05f)), mwhite = Matte(Spectrum(.73f, .73f, .73f));
    Float L = 5;
    auto room = BoxTriangles(bs, Transform(), 10, L, L, L);
    for (int i = 0; i < 10; ++i) prims.push_back(make_shared<GeometricPrimitive>(room[i], i < 6 ? mwhite : (i < 8 ? mred : mgreen), nullptr));
    Float h = 1.486f;
    auto cube = BoxTriangles(bs, Translate(Vector3f(0.7f, 0.f, 2.8f)) * RotateY(18) * Transform(), 12, h, h, h);
    std::shared_ptr<Material> mcube = mwhite;
    if (plastic) mcube = make_shared<PlasticMaterial>(Const(Spectrum(.25f, .35f, .65f)), Const(Spectrum(.4f, .4f, .4f)), ConstF(0.15f), true);
    if (roughGlass) mcube = make_shared<RoughGlassMaterial>(Const(Spectrum(.9f, .9f, .9f)), Const(Spectrum(.95f, .85f, .7f)), ConstF(0.25f), ConstF(1.5f), true);
    for (int i = 0; i < 12; ++i) prims.push_back(make_shared<GeometricPrimitive>(cube[i], mcube, nullptr));
    // the tall block: its "top" face sits at y = h while the sides reach 2h (RayTracing.cpp:166-167)
    auto block = BoxTriangles(bs, Translate(Vector3f(2.7f, 0.f, 1.0f)) * RotateY(-15) * Transform(), 12, h, h, h * 2);
    for (int i = 0; i < 12; ++i) prims.push_back(make_shared<GeometricPrimitive>(block[i], i < 10 ? mwhite : mirror, nullptr));
    if (!envMap.empty())
        lights.push_back(make_shared<InfiniteAreaLight>(RotateY(30) * RotateX(-90), Spectrum(1.5f, 1.2f, 1.0f), 1, envMap));
    bs.scene = make_shared<Scene>(make_shared<BVHAccel>(prims), lights);
    bs.lookat = LookAt(Point3f(2.5f, 2.5f, 11.8f), Point3f(2.5f, 2.5f, 0.0f), Vector3f(0.0f, 1.0f, 0.0f));
}

// "cornell_metal": the Cornell box with the branches no shipped scene reaches.
//   tall block   MetalMaterial exactly as the reference constructs (and never attaches) it, RayTracing.cpp:57-63:
//                eta (0.18, 0.15, 0.81), k 0.11, roughness 0.2 on both axes, no remap             material.cpp:51-67, reflection.cpp:23-42
//   short cube   MetalMaterial with anisotropic roughness 0.05 / 0.4 through RoughnessToAlpha       microfacet.h:74-78
//   ceiling light, first triangle: wound the other way round (geometric normal up) in a mesh whose vertex normals point down,
//                so Triangle::Sample turns the sampled normal to face them                          triangle.cpp:237-241
//   ceiling light, second triangle: wound the other way round with reverseOrientation = true        triangle.cpp:242-243
//   a small two-sided emitter standing in the room                                                  light.h:73-75
void BuildCornellMetal(BuiltScene &bs) {
    std::vector<std::shared_ptr<Primitive>> prims; std::vector<std::shared_ptr<Light>> lights;
    std::shared_ptr<Material> mlight = Matte(Spectrum(1.0f, 1.0f, 1.0f));
    {
        const float y = 5 - 0.01; float len = 0.5;
        Transform o2w = Translate(Vector3f(2.5f, 0.0f, 2.5f)) * Transform();
        const Transform *po = bs.Keep(o2w), *pi = bs.Keep(Inverse(o2w));
        int idx[3] = {0, 1, 2};
        Point3f A[3] = {Point3f(-len, y, len), Point3f(len, y, len), Point3f(-len, y, -len)};      // Cross(p1 - p0, p2 - p0) points up
        Normal3f NA[3] = {Normal3f(0.f, -1.f, 0.f), Normal3f(0.f, -1.f, 0.f), Normal3f(0.f, -1.f, 0.f)};
        auto t0 = CreateTriangleMesh(po, pi, false, 1, idx, 3, A, nullptr, NA, nullptr, nullptr);
        Point3f B[3] = {Point3f(len, y, len), Point3f(len, y, -len), Point3f(-len, y, -len)};        // up as well; reverseOrientation turns the sample's
        auto t1 = CreateTriangleMesh(po, pi, true, 1, idx, 3, B, nullptr, nullptr, nullptr, nullptr);
        for (auto &tri : {t0[0], t1[0]}) {
            std::shared_ptr<AreaLight> area = make_shared<DiffuseAreaLight>(*po, Spectrum(20.f), 1, tri, false);
            lights.push_back(area);
            prims.push_back(make_shared<GeometricPrimitive>(tri, mlight, area));
        }
        // two-sided emitter: a vertical triangle in the plane x = 2.2
        Transform o2 = Translate(Vector3f(2.2f, 3.2f, 3.4f)) * RotateY(25) * Transform();
        const Transform *qo = bs.Keep(o2), *qi = bs.Keep(Inverse(o2));
        Point3f Cc[3] = {Point3f(0.f, 0.f, -0.35f), Point3f(0.f, 0.f, 0.35f), Point3f(0.f, 0.6f, 0.f)};
        auto t2 = CreateTriangleMesh(qo, qi, false, 1, idx, 3, Cc, nullptr, nullptr, nullptr, nullptr);
        std::shared_ptr<AreaLight> area2 = make_shared<DiffuseAreaLight>(*qo, Spectrum(9.f, 7.f, 4.f), 1, t2[0], true);
        lights.push_back(area2);
        prims.push_back(make_shared<GeometricPrimitive>(t2[0], mlight, area2));
    }
    std::shared_ptr<Material> mirror = make_shared<MirrorMaterial>(Const(Spectrum(1.f)));
    std::shared_ptr<Material> mgreen = Matte(Spectrum(.12f, .45f, .15f)), mred = Matte(Spectrum(.65f, .05f, .05f)), mwhite = Matte(Spectrum(.73f, .73f, .73f));
    std::shared_ptr<Material> metal = make_shared<MetalMaterial>(Const(Spectrum(0.18f, 0.15f, 0.81f)), Const(Spectrum(0.11f, 0.11f, 0.11f)), ConstF(0.2f),
                                                                 ConstF(0.2f), ConstF(0.2f), false);
    std::shared_ptr<Material> brushed = make_shared<MetalMaterial>(Const(Spectrum(0.2f, 0.92f, 1.1f)), Const(Spectrum(3.9f, 2.45f, 2.14f)), ConstF(0.3f),
                                                                   ConstF(0.05f), ConstF(0.4f), true);
    Float L = 5;
    auto room = BoxTriangles(bs, Transform(), 10, L, L, L);
    for (int i = 0; i < 10; ++i) prims.push_back(make_shared<GeometricPrimitive>(room[i], i < 6 ? mwhite : (i < 8 ? mred : mgreen), nullptr));
    Float h = 1.486f;
    auto cube = BoxTriangles(bs, Translate(Vector3f(0.7f, 0.f, 2.8f)) * RotateY(18) * Transform(), 12, h, h, h);
    for (int i = 0; i < 12; ++i) prims.push_back(make_shared<GeometricPrimitive>(cube[i], brushed, nullptr));
    auto block = BoxTriangles(bs, Translate(Vector3f(2.7f, 0.f, 1.0f)) * RotateY(-15) * Transform(), 12, h, h, h * 2);
    for (int i = 0; i < 12; ++i) prims.push_back(make_shared<GeometricPrimitive>(block[i], i < 10 ? metal : mirror, nullptr));
    bs.scene = make_shared<Scene>(make_shared<BVHAccel>(prims), lights);
    bs.lookat = LookAt(Point3f(2.5f, 2.5f, 11.8f), Point3f(2.5f, 2.5f, 0.0f), Vector3f(0.0f, 1.0f, 0.0f));
}

// rand()-driven helpers of the reference's entry file (RayTracing.cpp:30-44).  The reference
// passes several rand() calls as arguments of one constructor call; the draws below are
// sequenced last-argument-first, which is what g++ and MSVC do for such calls.
double Rnd() { return rand() / (RAND_MAX + 1.0); }
double Rnd(double lo, double hi) { return lo + (hi - lo) * Rnd(); }
Vector3f RndVec() { double z = Rnd(), y = Rnd(), x = Rnd(); return Vector3f(x, y, z); }
Vector3f RndVec(double lo, double hi) { double z = Rnd(lo, hi), y = Rnd(lo, hi), x = Rnd(lo, hi); return Vector3f(x, y, z); }

void AddSphere(BuiltScene &bs, std::vector<std::shared_ptr<Primitive>> &prims, const Vector3f &c, Float r, const std::shared_ptr<Material> &m) {
    Transform o2w = Translate(c) * Transform();
    const Transform *po = bs.Keep(o2w), *pi = bs.Keep(Inverse(o2w));
    prims.push_back(make_shared<GeometricPrimitive>(make_shared<Sphere>(po, pi, false, r), m, nullptr));
}

// variant 2 ("spheres3"): the extended scene plus a sphere whose texture is a 300 x 200 PNG (not powers of two: resampled to
// 512 x 256 on load, mipmap.h:70-121) looked up through UVMapping2D(2, 3, 0.25, 0.5) (texture.cpp:5-13)
// envOnly ("spheres_env"): the scene's only light is an InfiniteAreaLight, so there is no sky term (integrator.cpp:249-254 needs
// an empty light list) and escaped rays carry nothing; all light arrives through next-event estimation.
void BuildSpheres(BuiltScene &bs, bool extended, const std::string &assets, bool oddTexture = false, bool envOnly = false) {
    srand(1);   // glibc's default seed, so repeated builds in one process give the same layout
    std::vector<std::shared_ptr<Primitive>> prims;
    AddSphere(bs, prims, Vector3f(0.f, -1000.f, 0.f), 1000.0, Matte(Spectrum(0.5, 0.5, 0.5)));
    for (int a = -11; a < 11; a++)
        for (int b = -11; b < 11; b++) {
            double choose_mat = Rnd();
            double cz = b + 0.8 * Rnd(), cx = a + 0.8 * Rnd();
            Point3f center(cx, 0.2, cz);
            if ((center - Point3f(4, 0.2, 0)).Length() > 1.0) {
                if (choose_mat < 0.8) {
                    Vector3f second = RndVec(), first = RndVec();
                    AddSphere(bs, prims, Vector3f(center), 0.2, Matte(Spectrum(first * second)));
                } else {
                    Spectrum albedo = RndVec(0.5, 1);
                    AddSphere(bs, prims, Vector3f(center), 0.2, make_shared<MirrorMaterial>(Const(albedo)));
                }
            }
        }
    std::shared_ptr<Material> big = Matte(Spectrum(0.4, 0.2, 0.1));
    if (extended) {
        shared_ptr<Texture<Spectrum>> earth = make_shared<ImageTexture<RGBSpectrum, Spectrum>>(
            make_unique<UVMapping2D>(1.f, 1.f, 0.f, 0.f), assets + "/earthmap.jpg", false, 8.f, ImageWrap::Repeat, 1.f, false);
        big = make_shared<MatteMaterial>(earth, ConstF(0.0f));
    }
    AddSphere(bs, prims, Vector3f(-4, 1, 0), 1.0, big);
    AddSphere(bs, prims, Vector3f(4, 1, 0), 1.0, make_shared<MirrorMaterial>(Const(Spectrum(0.7, 0.6, 0.5))));
    if (extended)
        AddSphere(bs, prims, Vector3f(0, 1, 0), 1.0, make_shared<GlassMaterial>(Const(Spectrum(1.f)), Const(Spectrum(1.f)), ConstF(1.5f)));
    if (oddTexture) {
        shared_ptr<Texture<Spectrum>> pat = make_shared<ImageTexture<RGBSpectrum, Spectrum>>(
            make_unique<UVMapping2D>(2.f, 3.f, 0.25f, 0.5f), assets + "/pattern_300x200.png", false, 8.f, ImageWrap::Repeat, 1.f, false);
        AddSphere(bs, prims, Vector3f(6.5f, 0.8f, 2.2f), 0.8, make_shared<MatteMaterial>(pat, ConstF(0.0f)));
    }
    std::vector<std::shared_ptr<Light>> lights;
    if (envOnly) lights.push_back(make_shared<InfiniteAreaLight>(RotateX(-90), Spectrum(1.f), 1, assets + "/env_48x20.hdr"));
    bs.scene = make_shared<Scene>(make_shared<BVHAccel>(prims), lights);
    bs.lookat = LookAt(Point3f(13, 2, 3), Point3f(0.f, 0.f, 0.f), Vector3f(0.0f, 1.0f, 0.0f));
}

// room_albedo: the reference's room is white (1); "helmet_grey" takes 0.8, so that no throughput sits exactly on Li's roulette threshold
void BuildHelmet(BuiltScene &bs, const std::string &assets, const std::string &model = "/DamagedHelmet/glTF/DamagedHelmet.gltf", Float room_albedo = 1.0f) {
    std::vector<std::shared_ptr<Primitive>> prims; std::vector<std::shared_ptr<Light>> lights;
    std::shared_ptr<Material> m1 = Matte(Spectrum(room_albedo, room_albedo, room_albedo));
    AddCeilingLight(bs, prims, lights, m1, 25.f);
    Float L = 5;
    auto room = BoxTriangles(bs, Transform(), 10, L, L, L);
    for (int i = 0; i < 10; ++i) prims.push_back(make_shared<GeometricPrimitive>(room[i], m1, nullptr));
    Float ts = 2.5f;
    Transform o2w = Translate(Vector3f(ts, ts, ts)) * RotateY(-90) * RotateX(-90);
    bs.loaders.emplace_back(new ModelLoad);
    ModelLoad &ML = *bs.loaders.back();
    ML.loadModel(assets + model, o2w);
    if (ML.meshes.empty()) throw std::runtime_error("helmet: could not load " + assets + model);
    ML.buildTextureModel(o2w, prims);
    bs.scene = make_shared<Scene>(make_shared<BVHAccel>(prims), lights);
    bs.lookat = LookAt(Point3f(1.f + ts, 0.7f + ts, -1.f + ts), Point3f(-0.5f + ts, -0.2f + ts, 0.3f + ts), Vector3f(0.0f, 1.0f, 0.0f));
}

// MetalRoughSphere() (RayTracing.cpp:194-216): the glTF sample model under RotateZ(180) RotateY(180) RotateX(-90), one
// MetalRoughnessMaterial per mesh from the model's two textures, no light (so the sky gradient lights the scene), eye (0, 0, 50).
void BuildMRSpheres(BuiltScene &bs, const std::string &assets) {
    std::vector<std::shared_ptr<Primitive>> prims; std::vector<std::shared_ptr<Light>> lights;
    Transform o2w = RotateZ(180) * RotateY(180) * RotateX(-90) * Transform();
    bs.loaders.emplace_back(new ModelLoad);
    ModelLoad &ML = *bs.loaders.back();
    const std::string model = assets + "/MetalRoughSpheres/glTF/MetalRoughSpheres.gltf";
    ML.loadModel(model, o2w);
    if (ML.meshes.empty()) throw std::runtime_error("mrspheres: could not load " + model + " (its geometry buffer MetalRoughSpheres0.bin is not part of the reference; tests/mrspheres_bin.py writes a stand-in)");
    ML.buildTextureModel(o2w, prims);
    bs.scene = make_shared<Scene>(make_shared<BVHAccel>(prims), lights);
    bs.lookat = LookAt(Point3f(0.f, 0.f, 50.0f), Point3f(0.f, 0.f, 0.0f), Vector3f(0.0f, 1.0f, 0.0f));
}

// PCG32 (O'Neill 2014), sequence 0xC0FFEE.
struct Pcg {
    uint64_t state = 0, inc = 0;
    explicit Pcg(uint64_t seq) { inc = (seq << 1u) | 1u; next(); state += 0x853c49e6748fea9bULL; next(); }
    uint32_t next() { uint64_t o = state; state = o * 0x5851f42d4c957f2dULL + inc; uint32_t x = (uint32_t)(((o >> 18u) ^ o) >> 27u); uint32_t r = (uint32_t)(o >> 59u); return (x >> r) | (x << ((~r + 1u) & 31)); }
    float uniform() { return std::min(0.99999994f, next() * 2.3283064365386963e-10f); }
};
void BuildSoup(BuiltScene &bs, int64_t n) {
    if (n <= 0) n = 100000;
    auto fs = std::make_unique<FlatScene>();
    fs->tri_p.resize(9 * (size_t)n); fs->tri_uv.resize(6 * (size_t)n);
    fs->prim_kind.assign(n, RT_PRIM_TRIANGLE); fs->prim_geom.resize(n); fs->prim_material.assign(n, 0); fs->prim_light.assign(n, -1);
    Pcg rng(0xC0FFEE);
    const float duv[6] = {0, 0, 1, 0, 1, 1};
    for (int64_t i = 0; i < n; ++i) {
        float c[3]; for (int k = 0; k < 3; ++k) c[k] = 100.f * rng.uniform();
        for (int v = 0; v < 3; ++v) for (int k = 0; k < 3; ++k) fs->tri_p[9 * i + 3 * v + k] = c[k] + (0.3f * rng.uniform() - 0.15f);
        memcpy(&fs->tri_uv[6 * i], duv, sizeof(duv));
        fs->prim_geom[i] = (int32_t)i;
    }
    rt_texture t; memset(&t, 0, sizeof(t)); t.kind = RT_TEX_CONST; t.value[0] = t.value[1] = t.value[2] = 0.5f; fs->textures.push_back(t);
    rt_material m; memset(&m, 0, sizeof(m)); m.kind = RT_MAT_MATTE; m.tex[0] = 0; m.tex[1] = m.tex[2] = m.tex[3] = -1; fs->materials.push_back(m);
    fs->Bind();
    std::vector<float> bounds(6 * (size_t)n);
    rt_prim_bounds(&fs->desc, bounds.data());
    fs->nodes.resize(2 * (size_t)n); fs->prim_order.resize(n);
    int32_t nn = 0;
    if (rt_bvh_build((int32_t)n, bounds.data(), fs->nodes.data(), &nn, fs->prim_order.data()) != 0) throw std::runtime_error("soup: rt_bvh_build failed");
    fs->nodes.resize(nn); fs->nodes.shrink_to_fit();
    fs->Bind();
    bs.flat = std::move(fs);
    bs.lookat = LookAt(Point3f(-60.f, 50.f, -60.f), Point3f(50.f, 50.f, 50.f), Vector3f(0.0f, 1.0f, 0.0f));
}

}  // namespace

extern "C" {

struct rth_scene { BuiltScene b; };

const char *rth_last_error(void) { return g_error.c_str(); }

rth_scene *rth_scene_build(const char *name, const char *assets_dir, int64_t param) {
    std::string nm = name ? name : "", assets = assets_dir ? assets_dir : ".";
    rth_scene *h = new rth_scene;
    try {
        if (nm == "cornell") BuildCornell(h->b);
        else if (nm == "cornell_plastic") BuildCornell(h->b, true);
        else if (nm == "sphere1") BuildSpheres(h->b, false, assets);
        else if (nm == "spheres2") BuildSpheres(h->b, true, assets);
        else if (nm == "spheres3") BuildSpheres(h->b, true, assets, true);
        else if (nm == "cornell_metal") BuildCornellMetal(h->b);
        else if (nm == "cornell_roughglass") BuildCornell(h->b, false, "", true);
        else if (nm == "cornell_env") BuildCornell(h->b, false, assets + "/env_48x20.hdr");
        else if (nm == "spheres_env") BuildSpheres(h->b, true, assets, false, true);
        else if (nm == "helmet") BuildHelmet(h->b, assets);
        else if (nm == "helmet_glb") BuildHelmet(h->b, assets, "/DamagedHelmet/glTF-Binary/DamagedHelmet.glb");   // the same model as one binary container
        else if (nm == "helmet_grey") BuildHelmet(h->b, assets, "/DamagedHelmet/glTF/DamagedHelmet.gltf", 0.8f);
        else if (nm == "mrspheres") BuildMRSpheres(h->b, assets);
        else if (nm == "soup") BuildSoup(h->b, param);
        else throw std::runtime_error("unknown scene '" + nm + "'");
        if (!h->b.flat) h->b.flat = FlattenScene(*h->b.scene);
    } catch (std::exception &e) { g_error = e.what(); delete h; return nullptr; }
    return h;
}
const rt_scene_desc *rth_scene_desc(rth_scene *h) { return &h->b.flat->desc; }
// File an image texture was loaded from ("" for constants, registered images and scenes without an object graph).
const char *rth_scene_texture_source(rth_scene *h, int index) {
    const std::vector<std::string> &src = h->b.flat->texture_sources;
    return (index >= 0 && index < (int)src.size()) ? src[index].c_str() : "";
}
// Constructor arguments of infinite light `index`: its .hdr file and scale (L receives three floats).
const char *rth_scene_env_source(rth_scene *h, int index, float *L) {
    const auto &src = h->b.flat->env_sources;
    if (index < 0 || index >= (int)src.size()) return "";
    if (L) memcpy(L, src[index].L, sizeof(src[index].L));
    return src[index].path.c_str();
}
int rth_scene_camera(rth_scene *h, int width, int height, rt_camera *out) {
    Transform c2w = Inverse(h->b.lookat);
    std::unique_ptr<PerspectiveCamera> cam(CreatePerspectiveCamera(c2w, h->b.nearPlane, h->b.fov, width, height));
    return cam->Describe(out) ? 0 : -1;
}
void rth_scene_free(rth_scene *h) { delete h; }
void rth_register_image(const char *name, int w, int h, const uint8_t *rgb8) { RegisterImage(name, w, h, rgb8); }
// Load a .gltf / .glb with ModelLoad and hand back its triangles as the loader leaves them (object space of ObjectToWorld =
// identity): 9 floats of positions and 6 of uv per triangle, and the texture names of the first mesh.  Pass null outputs to
// learn n_tris first.  0 = ok.
int rth_model_probe(const char *path, int apply_node_transforms, int *n_tris, float *tri_p, float *tri_uv, char *names, int names_cap) {
    if (!path || !n_tris) return -1;
    ModelLoad ML; ML.applyNodeTransforms = apply_node_transforms != 0;
    Transform id;
    try { ML.loadModel(path, id); } catch (std::exception &e) { g_error = e.what(); return -1; }     // no exception crosses the C boundary
    if (ML.meshes.empty()) return -1;
    int n = 0;
    for (auto &m : ML.meshes) {
        for (int t = 0; t < m->nTriangles; ++t, ++n) {
            if (!tri_p) continue;
            for (int k = 0; k < 3; ++k) {
                int v = m->vertexIndices[3 * t + k];
                tri_p[9 * n + 3 * k] = m->p[v].x; tri_p[9 * n + 3 * k + 1] = m->p[v].y; tri_p[9 * n + 3 * k + 2] = m->p[v].z;
                if (tri_uv) { tri_uv[6 * n + 2 * k] = m->uv ? m->uv[v].x : 0.f; tri_uv[6 * n + 2 * k + 1] = m->uv ? m->uv[v].y : 0.f; }
            }
        }
    }
    *n_tris = n;
    if (names && names_cap > 0) snprintf(names, (size_t)names_cap, "%s|%s", ML.diffTexName.empty() ? "" : ML.diffTexName[0].c_str(), ML.specTexName.empty() ? "" : ML.specTexName[0].c_str());
    return 0;
}
// Radiance .hdr files: write w*h*3 floats (rows top-down, flat RGBE); read back what the reference's loader would hand its
// InfiniteAreaLight (rows bottom-up when flip is set).  Call rth_read_hdr with out == nullptr for the size.  0 = ok.
int rth_write_hdr(const char *path, int w, int h, const float *rgb) { return (path && rgb && WriteHDR(path, w, h, rgb)) ? 0 : -1; }
int rth_read_hdr(const char *path, int flip, int *w, int *h, float *out) {
    if (!path || !w || !h) return -1;
    std::vector<float> rgb;
    try { if (!ReadHDR(path, w, h, &rgb, flip != 0)) return -1; } catch (std::exception &e) { g_error = e.what(); return -1; }
    if (out) memcpy(out, rgb.data(), rgb.size() * sizeof(float));
    return 0;
}
// Decode a PNG / JPEG held in memory.  Call with out == nullptr for the size, then with a w*h*3 buffer.  0 = ok.
int rth_decode_image(const uint8_t *data, size_t size, int *w, int *h, uint8_t *out) {
    static thread_local std::vector<uint8_t> last; static thread_local const uint8_t *last_data = nullptr; static thread_local int lw = 0, lh = 0;
    if (!data || !w || !h) return -1;
    if (last_data != data || last.empty()) {
        bool ok = false;
        try { ok = DecodeImageRGB8(data, size, &lw, &lh, &last); } catch (std::exception &e) { g_error = e.what(); }
        if (!ok) { last.clear(); last_data = nullptr; return -1; }
        last_data = data;
    }
    *w = lw; *h = lh;
    if (out) { memcpy(out, last.data(), last.size()); last.clear(); last_data = nullptr; }
    return 0;
}

// Render a named scene the way the reference's main() does (RayTracing.cpp:410-434): through
// PathIntegrator::Render, i.e. the drop-in path.  out_sum: w*h*3 floats.
int rth_render_like_main(rth_scene *h, int width, int height, int strata_x, int strata_y, int max_depth, const char *png_path,
                         float *out_sum, rt_render_stats *stats) {
    try {
        if (!h->b.scene) throw std::runtime_error("scene has no object graph");
        Transform c2w = Inverse(h->b.lookat);
        shared_ptr<Camera> cam = shared_ptr<Camera>(CreatePerspectiveCamera(c2w, h->b.nearPlane, h->b.fov, width, height));
        shared_ptr<Sampler> ss = make_unique<StratifiedSampler>(strata_x, strata_y, true, 1);
        Bounds2i pixelBounds;
        auto pi = make_shared<PathIntegrator>(max_depth, cam, ss, pixelBounds, 1.f);
        pi->outputPath = png_path ? png_path : "";
        shared_ptr<Integrator> integ = pi;
        integ->Render(*h->b.scene, width, height);
        if (out_sum) memcpy(out_sum, pi->lastSum.data(), pi->lastSum.size() * sizeof(float));
        if (stats) *stats = pi->lastStats;
    } catch (std::exception &e) { g_error = e.what(); return -1; }
    return 0;
}

// Known-answer helpers for the transform code: ops are applied as M = op[0] * op[1] * ... ;
// kind 0 = Translate(x,y,z), 1 = RotateX(x), 2 = RotateY(x), 3 = RotateZ(x), 4 = Scale(x,y,z).
int rth_kat_transform(int n_ops, const int *kinds, const float *args, int n_pts, const float *pts, float *out_pts, float *out_m, float *out_minv) {
    Transform t;
    for (int i = 0; i < n_ops; ++i) {
        const float *a = args + 3 * i; Transform o;
        switch (kinds[i]) { case 0: o = Translate(Vector3f(a[0], a[1], a[2])); break; case 1: o = RotateX(a[0]); break; case 2: o = RotateY(a[0]); break;
                            case 3: o = RotateZ(a[0]); break; case 4: o = Scale(a[0], a[1], a[2]); break; default: return -1; }
        t = (i == 0) ? o : t * o;
    }
    for (int i = 0; i < n_pts; ++i) { Point3f p = t(Point3f(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2])); out_pts[3 * i] = p.x; out_pts[3 * i + 1] = p.y; out_pts[3 * i + 2] = p.z; }
    if (out_m) memcpy(out_m, t.GetMatrix().m, 64);
    if (out_minv) memcpy(out_minv, t.GetInverseMatrix().m, 64);
    return 0;
}
int rth_kat_camera(const float *eye, const float *look, const float *up, float fov, float near_plane, int w, int h, rt_camera *out) {
    Transform la = LookAt(Point3f(eye[0], eye[1], eye[2]), Point3f(look[0], look[1], look[2]), Vector3f(up[0], up[1], up[2]));
    std::unique_ptr<PerspectiveCamera> cam(CreatePerspectiveCamera(Inverse(la), near_plane, fov, w, h));
    return cam->Describe(out) ? 0 : -1;
}

}  // extern "C"

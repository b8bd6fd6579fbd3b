// rt_host.h — host-side scene-building surface of the MI355X path-tracing backend.
//
// The reference renderer (sedcftyv/RayTracing) builds scenes by constructing a graph
// of shared_ptr objects (RayTracing/RayTracing.cpp:46-408) and renders through
// Integrator::Render (include/core/integrator.h:17-21).  The classes below keep the
// reference's names and constructor signatures so those scene functions compile
// unchanged, but they are plain recorders: they hold exactly the data the device
// needs, FlattenScene() turns the graph into the rt_scene_desc arrays of rt_amd.h,
// and PathIntegrator::Render hands that to the HIP kernels through the C ABI.
// Nothing here traces a ray on the CPU.
//
// Numerical conventions follow the reference where they change the flattened data:
//   Transform / Inverse / LookAt / Perspective / Rotate*   include/core/transform.{h,cpp}
//   TriangleMesh stores world-space vertices               include/core/triangle.cpp:8-36
//   PerspectiveCamera matrices                             include/core/camera.h:24-45, camera.cpp:60-79
#pragma once
#ifndef RT_AMD_HOST_H
#define RT_AMD_HOST_H

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "rt_amd.h"

// The reference's pbrt.h:19-24 exports these names at global scope and its scene
// scripts rely on that.
using std::vector;
using std::shared_ptr;
using std::unique_ptr;
using std::make_shared;
using std::make_unique;
using std::string;

typedef float Float;
static const Float Pi = 3.14159265358979323846;
static const Float InvPi = 0.31830988618379067154;
static constexpr Float Infinity = std::numeric_limits<Float>::infinity();
inline Float Radians(Float deg) { return (Pi / 180) * deg; }
inline Float Degrees(Float rad) { return (180 / Pi) * rad; }

// ------------------------------------------------------------------ tuples --
template <typename T> struct Point3;
template <typename T> struct Normal3;
template <typename T> struct Point2;

template <typename T> struct Vector2 {
    T x, y;
    Vector2() : x(0), y(0) {}
    Vector2(T a, T b) : x(a), y(b) {}
    T operator[](int i) const { return i == 0 ? x : y; }
    T &operator[](int i) { return i == 0 ? x : y; }
    Vector2 operator+(const Vector2 &v) const { return Vector2(x + v.x, y + v.y); }
    Vector2 operator-(const Vector2 &v) const { return Vector2(x - v.x, y - v.y); }
};

template <typename T> struct Vector3 {
    T x, y, z;
    Vector3() : x(0), y(0), z(0) {}
    Vector3(T a, T b, T c) : x(a), y(b), z(c) {}
    explicit Vector3(const Point3<T> &p);
    explicit Vector3(const Normal3<T> &n);
    T operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    T &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    Vector3 operator+(const Vector3 &v) const { return Vector3(x + v.x, y + v.y, z + v.z); }
    Vector3 operator-(const Vector3 &v) const { return Vector3(x - v.x, y - v.y, z - v.z); }
    Vector3 operator-() const { return Vector3(-x, -y, -z); }
    Vector3 operator*(T s) const { return Vector3(s * x, s * y, s * z); }
    Vector3 operator/(T f) const { Float inv = (Float)1 / f; return Vector3(x * inv, y * inv, z * inv); }
    Vector3 &operator+=(const Vector3 &v) { x += v.x; y += v.y; z += v.z; return *this; }
    Float LengthSquared() const { return x * x + y * y + z * z; }
    Float Length() const { return std::sqrt(LengthSquared()); }
};
template <typename T> Vector3<T> operator*(const Vector3<T> &v, const Vector3<T> &w) { return Vector3<T>(v.x * w.x, v.y * w.y, v.z * w.z); }
template <typename T, typename U> Vector3<T> operator*(U s, const Vector3<T> &v) { return v * (T)s; }
template <typename T> T Dot(const Vector3<T> &a, const Vector3<T> &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> Vector3<T> Cross(const Vector3<T> &a, const Vector3<T> &b) {
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;   // double intermediates, geometry.h:183-190
    return Vector3<T>((T)((ay * bz) - (az * by)), (T)((az * bx) - (ax * bz)), (T)((ax * by) - (ay * bx)));
}
template <typename T> Vector3<T> Normalize(const Vector3<T> &v) { return v / v.Length(); }

template <typename T> struct Point3 {
    T x, y, z;
    Point3() : x(0), y(0), z(0) {}
    Point3(T a, T b, T c) : x(a), y(b), z(c) {}
    explicit Point3(const Vector3<T> &v) : x(v.x), y(v.y), z(v.z) {}
    T operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    T &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    Point3 operator+(const Vector3<T> &v) const { return Point3(x + v.x, y + v.y, z + v.z); }
    Point3 operator+(const Point3 &p) const { return Point3(x + p.x, y + p.y, z + p.z); }
    Vector3<T> operator-(const Point3 &p) const { return Vector3<T>(x - p.x, y - p.y, z - p.z); }
    Point3 operator-(const Vector3<T> &v) const { return Point3(x - v.x, y - v.y, z - v.z); }
    Point3 operator*(T f) const { return Point3(f * x, f * y, f * z); }
    Point3 operator/(T f) const { Float inv = (Float)1 / f; return Point3(inv * x, inv * y, inv * z); }
};
template <typename T, typename U> Point3<T> operator*(U f, const Point3<T> &p) { return p * (T)f; }
template <typename T> Vector3<T>::Vector3(const Point3<T> &p) : x(p.x), y(p.y), z(p.z) {}

template <typename T> struct Point2 {
    T x, y;
    Point2() : x(0), y(0) {}
    Point2(T a, T b) : x(a), y(b) {}
    T operator[](int i) const { return i == 0 ? x : y; }
    T &operator[](int i) { return i == 0 ? x : y; }
};

template <typename T> struct Normal3 {
    T x, y, z;
    Normal3() : x(0), y(0), z(0) {}
    Normal3(T a, T b, T c) : x(a), y(b), z(c) {}
    explicit Normal3(const Vector3<T> &v) : x(v.x), y(v.y), z(v.z) {}
    T operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    T &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
template <typename T> Vector3<T>::Vector3(const Normal3<T> &n) : x(n.x), y(n.y), z(n.z) {}

typedef Vector2<Float> Vector2f;  typedef Vector2<int> Vector2i;
typedef Vector3<Float> Vector3f;  typedef Vector3<int> Vector3i;
typedef Point3<Float> Point3f;    typedef Point3<int> Point3i;
typedef Point2<Float> Point2f;    typedef Point2<int> Point2i;
typedef Normal3<Float> Normal3f;
using color = Vector3f;

template <typename T> struct Bounds2 {
    Point2<T> pMin, pMax;
    Bounds2() {
        T lo = std::numeric_limits<T>::lowest(), hi = std::numeric_limits<T>::max();
        pMin = Point2<T>(hi, hi); pMax = Point2<T>(lo, lo);
    }
};
typedef Bounds2<Float> Bounds2f;  typedef Bounds2<int> Bounds2i;

template <typename T> struct Bounds3 {
    Point3<T> pMin, pMax;
    Bounds3() {
        T lo = std::numeric_limits<T>::lowest(), hi = std::numeric_limits<T>::max();
        pMin = Point3<T>(hi, hi, hi); pMax = Point3<T>(lo, lo, lo);
    }
    Bounds3(const Point3<T> &a, const Point3<T> &b)
        : pMin(std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)),
          pMax(std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)) {}
};
typedef Bounds3<Float> Bounds3f;

// --------------------------------------------------------------- transform --
struct Matrix4x4 {
    Float m[4][4];
    Matrix4x4();
    Matrix4x4(Float t00, Float t01, Float t02, Float t03, Float t10, Float t11, Float t12, Float t13,
              Float t20, Float t21, Float t22, Float t23, Float t30, Float t31, Float t32, Float t33);
    static Matrix4x4 Mul(const Matrix4x4 &a, const Matrix4x4 &b);
};
Matrix4x4 Transpose(const Matrix4x4 &);
Matrix4x4 Inverse(const Matrix4x4 &);

class Transform {
public:
    Transform() {}
    Transform(const Matrix4x4 &mat) : m(mat), mInv(Inverse(mat)) {}
    Transform(const Matrix4x4 &mat, const Matrix4x4 &inv) : m(mat), mInv(inv) {}
    friend Transform Inverse(const Transform &t) { return Transform(t.mInv, t.m); }
    const Matrix4x4 &GetMatrix() const { return m; }
    const Matrix4x4 &GetInverseMatrix() const { return mInv; }
    Point3f operator()(const Point3f &p) const;
    Vector3f operator()(const Vector3f &v) const;
    Normal3f operator()(const Normal3f &n) const;
    Transform operator*(const Transform &t2) const;
    bool SwapsHandedness() const;
private:
    Matrix4x4 m, mInv;
};
Transform Translate(const Vector3f &delta);
Transform Scale(Float x, Float y, Float z);
Transform RotateX(Float theta);
Transform RotateY(Float theta);
Transform RotateZ(Float theta);
Transform Rotate(Float theta, const Vector3f &axis);
Transform LookAt(const Point3f &pos, const Point3f &look, const Vector3f &up);
Transform Perspective(Float fov, Float znear, Float zfar);

// ---------------------------------------------------------------- spectrum --
class RGBSpectrum {
public:
    RGBSpectrum(Float v = 0.f) { c[0] = c[1] = c[2] = v; }
    RGBSpectrum(Float r, Float g, Float b) { c[0] = r; c[1] = g; c[2] = b; }
    RGBSpectrum(Vector3f a) { c[0] = a[0]; c[1] = a[1]; c[2] = a[2]; }
    Float &operator[](int i) { return c[i]; }
    Float operator[](int i) const { return c[i]; }
    RGBSpectrum operator+(const RGBSpectrum &s) const { return RGBSpectrum(c[0] + s.c[0], c[1] + s.c[1], c[2] + s.c[2]); }
    RGBSpectrum operator-(const RGBSpectrum &s) const { return RGBSpectrum(c[0] - s.c[0], c[1] - s.c[1], c[2] - s.c[2]); }
    RGBSpectrum operator*(const RGBSpectrum &s) const { return RGBSpectrum(c[0] * s.c[0], c[1] * s.c[1], c[2] * s.c[2]); }
    RGBSpectrum operator*(Float a) const { return RGBSpectrum(c[0] * a, c[1] * a, c[2] * a); }
    friend RGBSpectrum operator*(Float a, const RGBSpectrum &s) { return s * a; }
    RGBSpectrum operator/(Float a) const { return RGBSpectrum(c[0] / a, c[1] / a, c[2] / a); }
    bool IsBlack() const { return c[0] == 0. && c[1] == 0. && c[2] == 0.; }
    static const int nSamples = 3;
private:
    Float c[3];
};
typedef RGBSpectrum Spectrum;

// ---------------------------------------------------------------- textures --
struct SurfaceInteraction;   // never materialised on the host; kept so signatures read like the reference's
enum class ImageWrap { Repeat, Black, Clamp };

// Every texture can describe itself as an rt_texture; image texels are owned by the texture.
class TextureBase {
public:
    virtual ~TextureBase() {}
    virtual void Describe(rt_texture *out) const = 0;
    virtual std::string Source() const { return ""; }
};
template <typename T> class Texture : public TextureBase {
public:
    virtual ~Texture() {}
};
template <typename T> class ConstantTexture : public Texture<T> {
public:
    ConstantTexture(const T &v) : value(v) {}
    void Describe(rt_texture *out) const override;
    T value;
};

class TextureMapping2D { public: virtual ~TextureMapping2D() {} };
class UVMapping2D : public TextureMapping2D {
public:
    UVMapping2D(Float su = 1, Float sv = 1, Float du = 0, Float dv = 0) : su(su), sv(sv), du(du), dv(dv) {}
    const Float su, sv, du, dv;
};

// Decoded image: float RGB, row-major, after the reference's load conventions
// (8-bit / 255, no gamma: imageio.cpp:19-45; Lanczos resample to powers of two: mipmap.h:70-121).
struct ImageData { int width = 0, height = 0; std::vector<float> rgb; std::string source; /* file it was decoded from, if any */ };
// 8-bit RGB pixels of a PNG (non-interlaced) or JPEG (baseline / progressive Huffman) held in memory.
bool DecodeImageRGB8(const uint8_t *data, size_t size, int *width, int *height, std::vector<uint8_t> *rgb);
bool DecodeJPEG(const uint8_t *data, size_t size, int *width, int *height, std::vector<uint8_t> *rgb);   // jpeg_decode.cpp
// Read an image file (PNG and JPEG natively; a ".png" that is missing falls back to its ".jpg" twin; images registered
// with RegisterImage by name come first).
// Returns nullptr when the file cannot be decoded.
std::shared_ptr<ImageData> LoadImageRGB(const std::string &filename);
// Make 8-bit RGB pixels available under `name` (used for formats this library does not decode itself).
void RegisterImage(const std::string &name, int width, int height, const uint8_t *rgb8);
std::shared_ptr<ImageData> ResampleToPow2(const std::shared_ptr<ImageData> &img);

template <typename Tmemory, typename Treturn> class ImageTexture : public Texture<Treturn> {
public:
    ImageTexture(std::unique_ptr<TextureMapping2D> mapping, const std::string &filename, bool doTrilinear, Float maxAniso,
                 ImageWrap wrapMode, Float scale, bool gamma);
    void Describe(rt_texture *out) const override;
    std::string Source() const override { return image ? image->source : ""; }
private:
    std::shared_ptr<ImageData> image;
    Float su = 1, sv = 1, du = 0, dv = 0;
};

// --------------------------------------------------------------- materials --
enum class TransportMode { Radiance, Importance };
class Material {
public:
    virtual ~Material() {}
    virtual int Kind() const = 0;
    // texture slots in rt_material order (see rt_amd.h), nullptr = absent
    virtual void Slots(const TextureBase *slots[4], int *remap) const = 0;
};
typedef std::shared_ptr<Texture<Spectrum>> SpectrumTex;
typedef std::shared_ptr<Texture<Float>> FloatTex;

class MatteMaterial : public Material {
public:
    MatteMaterial(const SpectrumTex &Kd, const FloatTex &sigma) : Kd(Kd), sigma(sigma) {}
    int Kind() const override { return RT_MAT_MATTE; }
    void Slots(const TextureBase *s[4], int *remap) const override { s[0] = Kd.get(); s[1] = s[2] = s[3] = nullptr; *remap = 0; }
private:
    SpectrumTex Kd; FloatTex sigma;
};
class MirrorMaterial : public Material {
public:
    MirrorMaterial(const SpectrumTex &r) : Kr(r) {}
    int Kind() const override { return RT_MAT_MIRROR; }
    void Slots(const TextureBase *s[4], int *remap) const override { s[0] = Kr.get(); s[1] = s[2] = s[3] = nullptr; *remap = 0; }
private:
    SpectrumTex Kr;
};
class GlassMaterial : public Material {
public:
    GlassMaterial(const SpectrumTex &Kr, const SpectrumTex &Kt, const FloatTex &index) : Kr(Kr), Kt(Kt), index(index) {}
    int Kind() const override { return RT_MAT_GLASS; }
    void Slots(const TextureBase *s[4], int *remap) const override { s[0] = Kr.get(); s[1] = Kt.get(); s[2] = index.get(); s[3] = nullptr; *remap = 0; }
private:
    SpectrumTex Kr, Kt; FloatTex index;
};
class PlasticMaterial : public Material {
public:
    PlasticMaterial(const SpectrumTex &Kd, const SpectrumTex &Ks, const FloatTex &roughness, bool remapRoughness)
        : Kd(Kd), Ks(Ks), roughness(roughness), remapRoughness(remapRoughness) {}
    int Kind() const override { return RT_MAT_PLASTIC; }
    void Slots(const TextureBase *s[4], int *remap) const override { s[0] = Kd.get(); s[1] = Ks.get(); s[2] = roughness.get(); s[3] = nullptr; *remap = remapRoughness; }
private:
    SpectrumTex Kd, Ks; FloatTex roughness; const bool remapRoughness;
};
class MetalMaterial : public Material {
public:
    MetalMaterial(const SpectrumTex &eta, const SpectrumTex &k, const FloatTex &rough, const FloatTex &urough,
                  const FloatTex &vrough, bool remapRoughness)
        : eta(eta), k(k), roughness(rough), uRoughness(urough), vRoughness(vrough), remapRoughness(remapRoughness) {}
    int Kind() const override { return RT_MAT_METAL; }
    void Slots(const TextureBase *s[4], int *remap) const override {
        s[0] = eta.get(); s[1] = k.get();
        s[2] = uRoughness ? uRoughness.get() : roughness.get();   // material.cpp:55-58
        s[3] = vRoughness ? vRoughness.get() : roughness.get();
        *remap = remapRoughness;
    }
private:
    SpectrumTex eta, k; FloatTex roughness, uRoughness, vRoughness; bool remapRoughness;
};
// Rough dielectric: MicrofacetReflection(Kr, TrowbridgeReitz(rough, rough), FresnelDielectric(1, index)) +
// MicrofacetTransmission(Kt, TrowbridgeReitz(rough, rough), 1, index, Radiance).  The reference has both lobes
// (include/core/reflection.cpp:241-266,311-355) but no material that assembles them; this one does, the way pbrt's glass does.
class RoughGlassMaterial : public Material {
public:
    RoughGlassMaterial(const SpectrumTex &Kr, const SpectrumTex &Kt, const FloatTex &roughness, const FloatTex &index, bool remapRoughness)
        : Kr(Kr), Kt(Kt), roughness(roughness), index(index), remapRoughness(remapRoughness) {}
    int Kind() const override { return RT_MAT_ROUGHGLASS; }
    void Slots(const TextureBase *s[4], int *remap) const override { s[0] = Kr.get(); s[1] = Kt.get(); s[2] = index.get(); s[3] = roughness.get(); *remap = remapRoughness; }
private:
    SpectrumTex Kr, Kt; FloatTex roughness, index; const bool remapRoughness;
};
class MetalRoughnessMaterial : public Material {
public:
    MetalRoughnessMaterial(const SpectrumTex &base_color, const SpectrumTex &metalroughness)
        : base_color(base_color), metalroughness(metalroughness) {}
    int Kind() const override { return RT_MAT_METALROUGH; }
    void Slots(const TextureBase *s[4], int *remap) const override { s[0] = base_color.get(); s[1] = metalroughness.get(); s[2] = s[3] = nullptr; *remap = 0; }
private:
    SpectrumTex base_color, metalroughness;
};

// ------------------------------------------------------------------ shapes --
class Shape {
public:
    Shape(const Transform *ObjectToWorld, const Transform *WorldToObject, bool reverseOrientation)
        : ObjectToWorld(ObjectToWorld), WorldToObject(WorldToObject), reverseOrientation(reverseOrientation),
          transformSwapsHandedness(ObjectToWorld->SwapsHandedness()) {}
    virtual ~Shape() {}
    virtual int Kind() const = 0;
    virtual Float Area() const = 0;
    const Transform *ObjectToWorld, *WorldToObject;
    const bool reverseOrientation, transformSwapsHandedness;
};

struct TriangleMesh {
    TriangleMesh(const Transform &ObjectToWorld, int nTriangles, const int *vertexIndices, int nVertices, const Point3f *P,
                 const Vector3f *S, const Normal3f *N, const Point2f *uv, const int *faceIndices);
    const int nTriangles, nVertices;
    std::vector<int> vertexIndices;
    std::unique_ptr<Point3f[]> p;     // world space
    std::unique_ptr<Normal3f[]> n;
    std::unique_ptr<Vector3f[]> s;
    std::unique_ptr<Point2f[]> uv;
    std::vector<int> faceIndices;
};

class Triangle : public Shape {
public:
    Triangle(const Transform *o2w, const Transform *w2o, bool reverseOrientation, const std::shared_ptr<TriangleMesh> &mesh, int triNumber)
        : Shape(o2w, w2o, reverseOrientation), mesh(mesh), v(&mesh->vertexIndices[3 * triNumber]) {}
    int Kind() const override { return RT_PRIM_TRIANGLE; }
    Float Area() const override;
    std::shared_ptr<TriangleMesh> mesh;
    const int *v;
};
std::vector<std::shared_ptr<Shape>> CreateTriangleMesh(const Transform *o2w, const Transform *w2o, bool reverseOrientation,
                                                       int nTriangles, const int *vertexIndices, int nVertices, const Point3f *p,
                                                       const Vector3f *s, const Normal3f *n, const Point2f *uv,
                                                       const int *faceIndices = nullptr);

class Sphere : public Shape {
public:
    Sphere(const Transform *o2w, const Transform *w2o, bool reverseOrientation, Float radius)
        : Shape(o2w, w2o, reverseOrientation), radius(radius) {}
    int Kind() const override { return RT_PRIM_SPHERE; }
    Float Area() const override { return 4 * Pi * radius * radius; }
    const Float radius;
};

// ------------------------------------------------------------------ lights --
class Light {
public:
    virtual ~Light() {}
};
class AreaLight : public Light {};

// Radiance .hdr files (RGBE, run-length encoded or flat): the format the reference's InfiniteAreaLight reads through stbi_loadf
// (infinite.cpp:16-18).  Floats are what that loader produces: mantissa byte * 2^(exponent - 136), three channels, and the rows
// bottom-up when flipVertically is set (stbi_set_flip_vertically_on_load(true), infinite.cpp:17).
bool ReadHDR(const std::string &path, int *width, int *height, std::vector<float> *rgb, bool flipVertically);
bool WriteHDR(const std::string &path, int width, int height, const float *rgb);      // flat RGBE, rows top-down

// The reference's InfiniteAreaLight (include/core/infinite.h, infinite.cpp:4-71): the constructor builds the radiance map
// (L * texel, then r + Sqrt(r), infinite.cpp:20-28), its MIP pyramid, and the Distribution2D over luminance * sin(theta)
// sampled at twice the map's resolution; Describe() hands all of it to rt_env_light.
class InfiniteAreaLight : public Light {
public:
    InfiniteAreaLight(const Transform &LightToWorld, const Spectrum &power, int nSamples, const std::string &texmap);
    void Describe(rt_env_light *out) const;      // world_center / world_radius are filled by FlattenScene (Preprocess, infinite.h:20-22)
    Transform LightToWorld, WorldToLight;
    std::string source; float scale[3];           // constructor arguments, kept for the test tooling
    int width = 0, height = 0;                    // level 0 of the map
    std::vector<float> texels;                    // width * height * 3
    int nu = 0, nv = 0;
    std::vector<float> func, cdf, funcInt, marginalCdf;
    float marginalInt = 0;
};
class DiffuseAreaLight : public AreaLight {
public:
    DiffuseAreaLight(const Transform &LightToWorld, const Spectrum &Le, int nSamples, const std::shared_ptr<Shape> &shape,
                     bool twoSided = false)
        : Lemit(Le), shape(shape), twoSided(twoSided), area(shape->Area()) {}
    const Spectrum Lemit;
    std::shared_ptr<Shape> shape;
    const bool twoSided;
    const Float area;
};

// -------------------------------------------------------------- primitives --
class Primitive {
public:
    virtual ~Primitive() {}
};
class GeometricPrimitive : public Primitive {
public:
    GeometricPrimitive(const std::shared_ptr<Shape> &shape, const std::shared_ptr<Material> &material,
                       const std::shared_ptr<AreaLight> &areaLight)
        : shape(shape), material(material), areaLight(areaLight) {}
    std::shared_ptr<Shape> shape;
    std::shared_ptr<Material> material;
    std::shared_ptr<AreaLight> areaLight;
};

// Owns every array an rt_scene_desc points at.
struct FlatScene {
    rt_scene_desc desc;
    std::vector<uint8_t> prim_kind, tri_has_n;
    std::vector<int32_t> prim_geom, prim_material, prim_light, prim_order;
    std::vector<float> tri_p, tri_uv, tri_n;
    std::vector<rt_sphere> spheres;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<rt_light> lights;
    std::vector<rt_env_light> env_lights;
    std::vector<rt_bvh_node> nodes;
    std::vector<std::shared_ptr<ImageData>> keepalive;
    std::vector<std::string> texture_sources;   // per texture: the file an image texture came from, else ""
    struct EnvSource { std::string path; float L[3]; };
    std::vector<EnvSource> env_sources;         // per infinite light: the .hdr file and the scale its constructor was given
    void Bind();   // point desc at the vectors
};

class Aggregate : public Primitive {
public:
    std::vector<std::shared_ptr<Primitive>> primitives;   // original push order
};
// Records the primitive list and the build parameters; the tree itself is built by rt_bvh_build_ex (the reference's
// recursiveBuild decisions, bvh.cpp:86-227) when the scene is flattened.
class BVHAccel : public Aggregate {
public:
    enum class SplitMethod { SAH, HLBVH, Middle, EqualCounts };
    BVHAccel(std::vector<std::shared_ptr<Primitive>> p, int maxPrimsInNode = 1, SplitMethod splitMethod = SplitMethod::Middle);
    int maxPrimsInNode;
    SplitMethod splitMethod;
};
class SimpleAccel : public Aggregate {
public:
    SimpleAccel(std::vector<std::shared_ptr<Primitive>> p) { primitives = std::move(p); }
};

class Scene {
public:
    Scene(std::shared_ptr<Primitive> aggregate, const std::vector<std::shared_ptr<Light>> &lights)
        : lights(lights), aggregate(aggregate) {}
    std::vector<std::shared_ptr<Light>> lights;
    std::vector<std::shared_ptr<Light>> infiniteLights;
    const std::shared_ptr<Primitive> &GetAggregate() const { return aggregate; }
private:
    std::shared_ptr<Primitive> aggregate;
};

// Flatten the object graph into rt_scene_desc arrays (original primitive order, BVH built by rt_bvh_build).
std::unique_ptr<FlatScene> FlattenScene(const Scene &scene);

// ------------------------------------------------------------------ camera --
struct CameraSample { Point2f pFilm; Point2f pLens; Float time; };
class Camera {
public:
    Camera(const Transform &c2w) : CameraToWorld(c2w) {}
    virtual ~Camera() {}
    virtual bool Describe(rt_camera *out) const = 0;
    Transform CameraToWorld;
};
class PerspectiveCamera : public Camera {
public:
    PerspectiveCamera(const Transform &CameraToWorld, const Bounds2f &screenWindow, Float lensRadius, Float focalDistance, Float fov,
                      int image_width, int image_height);
    bool Describe(rt_camera *out) const override;
    Transform CameraToScreen, RasterToCamera, ScreenToRaster, RasterToScreen;
    int width, height;
};
PerspectiveCamera *CreatePerspectiveCamera(const Transform &cam2world, Float focaldistance, Float fov, int image_width, int image_height);

// ----------------------------------------------------------------- sampler --
class Sampler {
public:
    Sampler(int64_t spp) : samplesPerPixel(spp) {}
    virtual ~Sampler() {}
    const int64_t samplesPerPixel;
    int xSamples = 1, ySamples = 1;
};
class StratifiedSampler : public Sampler {
public:
    StratifiedSampler(int xPixelSamples, int yPixelSamples, bool jitterSamples, int nSampledDimensions)
        : Sampler((int64_t)xPixelSamples * yPixelSamples) { xSamples = xPixelSamples; ySamples = yPixelSamples; }
};

// -------------------------------------------------------------- integrator --
class Integrator {
public:
    virtual ~Integrator() {}
    virtual void Render(const Scene &scene, int image_width, int image_height) = 0;
};
// Render() flattens the scene, runs the wavefront tracer on the GPU through the C ABI,
// tonemaps like write_color and writes image.png to the working directory, printing the
// elapsed seconds (integrator.cpp:340-412).  The float sum buffer of the last render is kept.
class PathIntegrator : public Integrator {
public:
    PathIntegrator(int maxDepth, std::shared_ptr<const Camera> camera, std::shared_ptr<Sampler> sampler, const Bounds2i &pixelBounds,
                   Float rrThreshold = 1, const std::string &lightSampleStrategy = "spatial")
        : maxDepth(maxDepth), camera(camera), sampler(sampler), rrThreshold(rrThreshold) {}
    void Render(const Scene &scene, int image_width, int image_height) override;
    // Render estimates with Li_re, as the reference's Render does (integrator.cpp:332,362).  The reference also defines Li
    // (integrator.cpp:175-229) and UniformSampleAllLights (:26-51) without calling either; these switches select them.
    bool estimateWithLi = false, sampleAllLights = false;
    std::vector<float> lastSum;      // width*height*3, per-pixel radiance sums
    rt_render_stats lastStats;
    uint32_t seedLo = 0x243F6A88u, seedHi = 0x85A308D3u;
    std::string outputPath = "image.png";
private:
    const int maxDepth;
    std::shared_ptr<const Camera> camera;
    std::shared_ptr<Sampler> sampler;
    const Float rrThreshold;
};

// ---------------------------------------------------------- model loading --
// glTF 2.0 (.gltf + external .bin) reader replacing the reference's Assimp-backed
// ModelLoad (include/core/modelload.{h,cpp}).  Import convention of the reference's
// ReadFile flags (modelload.cpp:9): positions and normals z -> -z (MakeLeftHanded),
// v -> 1-v (FlipUVs), node transforms ignored (modelload.cpp:18-25),
// baseColorTexture -> diffuse slot, metallicRoughnessTexture -> "unknown" slot.
class ModelLoad {
public:
    vector<shared_ptr<TriangleMesh>> meshes;
    string directory;
    vector<string> specTexName, diffTexName;
    bool applyNodeTransforms = false;   // extension (the reference never reads a node's transform, modelload.cpp:18-25)
    void loadModel(string path, const Transform &ObjectToWorld);
    void buildNoTextureModel(Transform &tri_Object2World, vector<shared_ptr<Primitive>> &prims, shared_ptr<Material> material);
    void buildTextureModel(Transform &tri_Object2World, vector<shared_ptr<Primitive>> &prims);
private:
    std::vector<std::unique_ptr<Transform>> keep;   // Shape holds Transform pointers
};
shared_ptr<Material> getDiffuseMaterial(string filename);
shared_ptr<Material> getMetalRoughnessMaterial(string diffFilename, string specFilename);

// PNG writer for the tonemapped frame (the reference uses stb_image_write, integrator.cpp:374,409).
bool WritePNG(const std::string &path, int width, int height, const uint8_t *rgb);

#endif  // RT_AMD_HOST_H

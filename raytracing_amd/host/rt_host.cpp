// rt_host.cpp — implementation of the host scene surface declared in rt_host.h.
#include "rt_host.h"

#include <chrono>
#include <fstream>
#include <functional>
#include <sstream>
#include <stdexcept>
#include <zlib.h>

// ------------------------------------------------------------- matrices ----
Matrix4x4::Matrix4x4() {
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m[i][j] = (i == j) ? 1.f : 0.f;
}
Matrix4x4::Matrix4x4(Float t00, Float t01, Float t02, Float t03, Float t10, Float t11, Float t12, Float t13, Float t20, Float t21,
                     Float t22, Float t23, Float t30, Float t31, Float t32, Float t33) {
    Float v[16] = {t00, t01, t02, t03, t10, t11, t12, t13, t20, t21, t22, t23, t30, t31, t32, t33};
    memcpy(m, v, sizeof(v));
}
Matrix4x4 Matrix4x4::Mul(const Matrix4x4 &a, const Matrix4x4 &b) {
    Matrix4x4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}
Matrix4x4 Transpose(const Matrix4x4 &a) {
    Matrix4x4 r;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[j][i];
    return r;
}

// Gauss-Jordan elimination with full pivoting, in the operation order of the
// reference (transform.cpp:34-82) so that inverted camera matrices are bit-identical.
Matrix4x4 Inverse(const Matrix4x4 &src) {
    int colOf[4], rowOf[4], used[4] = {0, 0, 0, 0};
    Float a[4][4];
    memcpy(a, src.m, sizeof(a));
    for (int step = 0; step < 4; ++step) {
        int prow = 0, pcol = 0;
        Float best = 0.f;
        for (int r = 0; r < 4; ++r) {
            if (used[r] == 1) continue;
            for (int c = 0; c < 4; ++c) {
                if (used[c] == 0) {
                    if (std::abs(a[r][c]) >= best) { best = Float(std::abs(a[r][c])); prow = r; pcol = c; }
                } else if (used[c] > 1)
                    fprintf(stderr, "Inverse: matrix has no inverse (a pivot column was chosen twice)\n");
            }
        }
        ++used[pcol];
        if (prow != pcol) for (int k = 0; k < 4; ++k) std::swap(a[prow][k], a[pcol][k]);
        rowOf[step] = prow; colOf[step] = pcol;
        if (a[pcol][pcol] == 0.f) fprintf(stderr, "Inverse: matrix has no inverse (zero pivot)\n");
        Float pivinv = 1. / a[pcol][pcol];
        a[pcol][pcol] = 1.;
        for (int k = 0; k < 4; ++k) a[pcol][k] *= pivinv;
        for (int r = 0; r < 4; ++r) {
            if (r == pcol) continue;
            Float save = a[r][pcol];
            a[r][pcol] = 0;
            for (int k = 0; k < 4; ++k) a[r][k] -= a[pcol][k] * save;
        }
    }
    for (int step = 3; step >= 0; --step)
        if (rowOf[step] != colOf[step])
            for (int k = 0; k < 4; ++k) std::swap(a[k][rowOf[step]], a[k][colOf[step]]);
    Matrix4x4 r;
    memcpy(r.m, a, sizeof(a));
    return r;
}

Point3f Transform::operator()(const Point3f &p) const {
    Float x = p.x, y = p.y, z = p.z;
    Float xp = m.m[0][0] * x + m.m[0][1] * y + m.m[0][2] * z + m.m[0][3];
    Float yp = m.m[1][0] * x + m.m[1][1] * y + m.m[1][2] * z + m.m[1][3];
    Float zp = m.m[2][0] * x + m.m[2][1] * y + m.m[2][2] * z + m.m[2][3];
    Float wp = m.m[3][0] * x + m.m[3][1] * y + m.m[3][2] * z + m.m[3][3];
    if (wp == 1) return Point3f(xp, yp, zp);
    return Point3f(xp, yp, zp) / wp;
}
Vector3f Transform::operator()(const Vector3f &v) const {
    Float x = v.x, y = v.y, z = v.z;
    return Vector3f(m.m[0][0] * x + m.m[0][1] * y + m.m[0][2] * z, m.m[1][0] * x + m.m[1][1] * y + m.m[1][2] * z,
                    m.m[2][0] * x + m.m[2][1] * y + m.m[2][2] * z);
}
Normal3f Transform::operator()(const Normal3f &n) const {
    Float x = n.x, y = n.y, z = n.z;
    return Normal3f(mInv.m[0][0] * x + mInv.m[1][0] * y + mInv.m[2][0] * z, mInv.m[0][1] * x + mInv.m[1][1] * y + mInv.m[2][1] * z,
                    mInv.m[0][2] * x + mInv.m[1][2] * y + mInv.m[2][2] * z);
}
Transform Transform::operator*(const Transform &t2) const { return Transform(Matrix4x4::Mul(m, t2.m), Matrix4x4::Mul(t2.mInv, mInv)); }
bool Transform::SwapsHandedness() const {
    Float det = m.m[0][0] * (m.m[1][1] * m.m[2][2] - m.m[1][2] * m.m[2][1]) - m.m[0][1] * (m.m[1][0] * m.m[2][2] - m.m[1][2] * m.m[2][0]) +
                m.m[0][2] * (m.m[1][0] * m.m[2][1] - m.m[1][1] * m.m[2][0]);
    return det < 0;
}
Transform Translate(const Vector3f &d) {
    return Transform(Matrix4x4(1, 0, 0, d.x, 0, 1, 0, d.y, 0, 0, 1, d.z, 0, 0, 0, 1), Matrix4x4(1, 0, 0, -d.x, 0, 1, 0, -d.y, 0, 0, 1, -d.z, 0, 0, 0, 1));
}
Transform Scale(Float x, Float y, Float z) {
    return Transform(Matrix4x4(x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1), Matrix4x4(1 / x, 0, 0, 0, 0, 1 / y, 0, 0, 0, 0, 1 / z, 0, 0, 0, 0, 1));
}
Transform RotateX(Float theta) {
    Float s = std::sin(Radians(theta)), c = std::cos(Radians(theta));
    Matrix4x4 m(1, 0, 0, 0, 0, c, -s, 0, 0, s, c, 0, 0, 0, 0, 1);
    return Transform(m, Transpose(m));
}
Transform RotateY(Float theta) {
    Float s = std::sin(Radians(theta)), c = std::cos(Radians(theta));
    Matrix4x4 m(c, 0, s, 0, 0, 1, 0, 0, -s, 0, c, 0, 0, 0, 0, 1);
    return Transform(m, Transpose(m));
}
Transform RotateZ(Float theta) {
    Float s = std::sin(Radians(theta)), c = std::cos(Radians(theta));
    Matrix4x4 m(c, -s, 0, 0, s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1);
    return Transform(m, Transpose(m));
}
Transform Rotate(Float theta, const Vector3f &axis) {
    Vector3f a = Normalize(axis);
    Float s = std::sin(Radians(theta)), c = std::cos(Radians(theta));
    Matrix4x4 m;
    m.m[0][0] = a.x * a.x + (1 - a.x * a.x) * c; m.m[0][1] = a.x * a.y * (1 - c) - a.z * s; m.m[0][2] = a.x * a.z * (1 - c) + a.y * s; m.m[0][3] = 0;
    m.m[1][0] = a.x * a.y * (1 - c) + a.z * s; m.m[1][1] = a.y * a.y + (1 - a.y * a.y) * c; m.m[1][2] = a.y * a.z * (1 - c) - a.x * s; m.m[1][3] = 0;
    m.m[2][0] = a.x * a.z * (1 - c) - a.y * s; m.m[2][1] = a.y * a.z * (1 - c) + a.x * s; m.m[2][2] = a.z * a.z + (1 - a.z * a.z) * c; m.m[2][3] = 0;
    return Transform(m, Transpose(m));
}
Transform LookAt(const Point3f &pos, const Point3f &look, const Vector3f &up) {
    Matrix4x4 c2w;
    c2w.m[0][3] = pos.x; c2w.m[1][3] = pos.y; c2w.m[2][3] = pos.z; c2w.m[3][3] = 1;
    Vector3f dir = Normalize(look - pos);
    if (Cross(Normalize(up), dir).Length() == 0) {
        fprintf(stderr, "LookAt: the up vector is parallel to the viewing direction; returning the identity\n");
        return Transform();
    }
    Vector3f right = Normalize(Cross(Normalize(up), dir));
    Vector3f newUp = Cross(dir, right);
    c2w.m[0][0] = right.x; c2w.m[1][0] = right.y; c2w.m[2][0] = right.z; c2w.m[3][0] = 0.;
    c2w.m[0][1] = newUp.x; c2w.m[1][1] = newUp.y; c2w.m[2][1] = newUp.z; c2w.m[3][1] = 0.;
    c2w.m[0][2] = dir.x; c2w.m[1][2] = dir.y; c2w.m[2][2] = dir.z; c2w.m[3][2] = 0.;
    return Transform(Inverse(c2w), c2w);
}
Transform Perspective(Float fov, Float n, Float f) {
    Matrix4x4 persp(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, f / (f - n), -f * n / (f - n), 0, 0, 1, 0);
    Float invTanAng = 1 / std::tan(Radians(fov) / 2);
    return Scale(invTanAng, invTanAng, 1) * Transform(persp);
}

// --------------------------------------------------------------- textures --
template <> void ConstantTexture<Float>::Describe(rt_texture *out) const {
    memset(out, 0, sizeof(*out)); out->kind = RT_TEX_CONST; out->value[0] = out->value[1] = out->value[2] = value;
}
template <> void ConstantTexture<Spectrum>::Describe(rt_texture *out) const {
    memset(out, 0, sizeof(*out)); out->kind = RT_TEX_CONST; out->value[0] = value[0]; out->value[1] = value[1]; out->value[2] = value[2];
}
template class ConstantTexture<Float>;
template class ConstantTexture<Spectrum>;

namespace {
struct Raw8 { int w, h; std::vector<uint8_t> rgb; };
std::map<std::string, Raw8> &Registry() { static std::map<std::string, Raw8> r; return r; }
std::string NormalizePath(std::string p) { for (char &c : p) if (c == '\\') c = '/'; return p; }
std::string BaseName(const std::string &p) { size_t k = p.find_last_of('/'); return k == std::string::npos ? p : p.substr(k + 1); }

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
int paeth(int a, int b, int c) { int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }

// Non-interlaced PNG, 8 or 16 bits per channel, all colour types -> 8-bit RGB.
bool DecodePNG(const std::vector<uint8_t> &buf, Raw8 *out) {
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (buf.size() < 8 || memcmp(buf.data(), sig, 8) != 0) return false;
    size_t pos = 8; uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    while (pos + 12 <= buf.size()) {
        uint32_t len = be32(&buf[pos]); const uint8_t *type = &buf[pos + 4]; const uint8_t *data = &buf[pos + 8];
        if (pos + 12 + len > buf.size()) return false;
        if (!memcmp(type, "IHDR", 4)) { w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12]; }
        else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + len;
    }
    if (!w || !h || interlace || (depth != 8 && depth != 16)) return false;
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch || (ctype == 3 && depth != 8)) return false;
    size_t bpp = (size_t)ch * depth / 8, stride = bpp * w;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf rawLen = raw.size();
    if (uncompress(raw.data(), &rawLen, idat.data(), idat.size()) != Z_OK || rawLen != raw.size()) return false;
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t *src = &raw[(stride + 1) * y]; int ft = src[0]; ++src;
        uint8_t *dst = &img[stride * y]; const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            int a = i >= bpp ? dst[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0, x = src[i];
            switch (ft) { case 0: break; case 1: x += a; break; case 2: x += b; break; case 3: x += (a + b) / 2; break; case 4: x += paeth(a, b, c); break; default: return false; }
            dst[i] = (uint8_t)x;
        }
    }
    out->w = (int)w; out->h = (int)h; out->rgb.resize((size_t)w * h * 3);
    size_t step = depth / 8;
    for (size_t p = 0; p < (size_t)w * h; ++p) {
        const uint8_t *px = &img[p * bpp]; uint8_t r, g, b;
        if (ctype == 0 || ctype == 4) r = g = b = px[0];
        else if (ctype == 3) { size_t k = px[0]; if (3 * k + 2 >= plte.size()) return false; r = plte[3 * k]; g = plte[3 * k + 1]; b = plte[3 * k + 2]; }
        else { r = px[0]; g = px[step]; b = px[2 * step]; }
        out->rgb[3 * p] = r; out->rgb[3 * p + 1] = g; out->rgb[3 * p + 2] = b;
    }
    return true;
}
Float LanczosW(Float x, Float tau = 2) {   // texture.cpp:15-23
    x = std::abs(x);
    if (x < 1e-5f) return 1;
    if (x > 1.f) return 0;
    x *= Pi;
    Float s = std::sin(x * tau) / (x * tau);
    Float l = std::sin(x) / x;
    return s * l;
}
struct RW { int first; Float w[4]; };
std::vector<RW> ResampleWeights(int oldRes, int newRes) {   // mipmap.h:31-47
    std::vector<RW> wt(newRes); Float filterwidth = 2.f;
    for (int i = 0; i < newRes; ++i) {
        Float center = (i + .5f) * oldRes / newRes;
        wt[i].first = (int)std::floor((center - filterwidth) + 0.5f);
        for (int j = 0; j < 4; ++j) { Float pos = wt[i].first + j + .5f; wt[i].w[j] = LanczosW((pos - center) / filterwidth); }
        Float inv = 1 / (wt[i].w[0] + wt[i].w[1] + wt[i].w[2] + wt[i].w[3]);
        for (int j = 0; j < 4; ++j) wt[i].w[j] *= inv;
    }
    return wt;
}
int ModI(int a, int b) { int r = a - (a / b) * b; return r < 0 ? r + b : r; }
int RoundUpPow2(int v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }
bool IsPow2(int v) { return v && !(v & (v - 1)); }
}  // namespace

bool DecodeImageRGB8(const uint8_t *data, size_t size, int *width, int *height, std::vector<uint8_t> *rgb) {
    if (size >= 8 && data[0] == 137 && data[1] == 'P') {
        Raw8 r; std::vector<uint8_t> buf(data, data + size);
        if (!DecodePNG(buf, &r)) return false;
        *width = r.w; *height = r.h; *rgb = std::move(r.rgb);
        return true;
    }
    if (size >= 3 && data[0] == 0xFF && data[1] == 0xD8) return DecodeJPEG(data, size, width, height, rgb);
    return false;
}

void RegisterImage(const std::string &name, int width, int height, const uint8_t *rgb8) {
    Raw8 r; r.w = width; r.h = height; r.rgb.assign(rgb8, rgb8 + (size_t)width * height * 3);
    Registry()[BaseName(NormalizePath(name))] = std::move(r);
}

std::shared_ptr<ImageData> LoadImageRGB(const std::string &filename) {
    std::string path = NormalizePath(filename);
    Raw8 raw; const Raw8 *src = nullptr; std::string from;
    auto it = Registry().find(BaseName(path));
    if (it != Registry().end()) src = &it->second;
    else {
        // The file itself; failing that its JPEG twin (DamagedHelmet.gltf names Default_albedo.png, the model ships Default_albedo.jpg).
        std::vector<std::string> tries = {path};
        size_t dot = path.find_last_of('.');
        if (dot != std::string::npos && path.substr(dot) == ".png") { tries.push_back(path.substr(0, dot) + ".jpg"); tries.push_back(path.substr(0, dot) + ".jpeg"); }
        for (const std::string &t : tries) {
            std::ifstream f(t, std::ios::binary);
            if (!f) continue;
            std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            if (DecodeImageRGB8(buf.data(), buf.size(), &raw.w, &raw.h, &raw.rgb)) { src = &raw; from = t; break; }
        }
    }
    if (!src) return nullptr;
    auto img = std::make_shared<ImageData>();
    img->source = from;
    img->width = src->w; img->height = src->h; img->rgb.resize(src->rgb.size());
    for (size_t i = 0; i < src->rgb.size(); ++i) img->rgb[i] = src->rgb[i] / 255.f;   // imageio.cpp:36-38
    return img;
}

// mipmap.h:70-121 (Repeat wrap): separable 4-tap Lanczos up-sampling to the next powers of two.
std::shared_ptr<ImageData> ResampleToPow2(const std::shared_ptr<ImageData> &img) {
    if (IsPow2(img->width) && IsPow2(img->height)) return img;
    int W = img->width, H = img->height, W2 = RoundUpPow2(W), H2 = RoundUpPow2(H);
    auto out = std::make_shared<ImageData>(); out->width = W2; out->height = H2; out->rgb.assign((size_t)W2 * H2 * 3, 0.f);
    out->source = img->source;
    std::vector<RW> sw = ResampleWeights(W, W2);
    for (int t = 0; t < H; ++t)
        for (int s = 0; s < W2; ++s)
            for (int c = 0; c < 3; ++c) {
                Float acc = 0.f;
                for (int j = 0; j < 4; ++j) { int o = ModI(sw[s].first + j, W); acc += sw[s].w[j] * img->rgb[3 * ((size_t)t * W + o) + c]; }
                out->rgb[3 * ((size_t)t * W2 + s) + c] = acc;
            }
    std::vector<RW> tw = ResampleWeights(H, H2);
    std::vector<Float> work((size_t)H2 * 3);
    for (int s = 0; s < W2; ++s) {
        for (int t = 0; t < H2; ++t)
            for (int c = 0; c < 3; ++c) {
                Float acc = 0.f;
                for (int j = 0; j < 4; ++j) { int o = ModI(tw[t].first + j, H); acc += tw[t].w[j] * out->rgb[3 * ((size_t)o * W2 + s) + c]; }
                work[3 * t + c] = acc;
            }
        for (int t = 0; t < H2; ++t)
            for (int c = 0; c < 3; ++c) { Float v = work[3 * t + c]; out->rgb[3 * ((size_t)t * W2 + s) + c] = v < 0.f ? 0.f : v; }
    }
    return out;
}

template <typename M, typename R>
ImageTexture<M, R>::ImageTexture(std::unique_ptr<TextureMapping2D> mapping, const std::string &filename, bool, Float, ImageWrap, Float scale, bool) {
    static std::map<std::string, std::shared_ptr<ImageData>> cache;   // imagemap.cpp:8-13
    if (UVMapping2D *uv = dynamic_cast<UVMapping2D *>(mapping.get())) { su = uv->su; sv = uv->sv; du = uv->du; dv = uv->dv; }
    std::string key = NormalizePath(filename);
    auto it = cache.find(key);
    if (it != cache.end()) { image = it->second; return; }
    std::shared_ptr<ImageData> img = LoadImageRGB(key);
    if (!img) {   // imagemap.cpp:16-22
        printf("Creating a constant grey texture to replace \"%s\".", filename.c_str());
        img = std::make_shared<ImageData>(); img->width = img->height = 1; img->rgb.assign(3, 0.5f);
    }
    if (scale != 1.f) for (float &v : img->rgb) v = scale * v;
    image = ResampleToPow2(img);
    cache[key] = image;
}
template <typename M, typename R> void ImageTexture<M, R>::Describe(rt_texture *out) const {
    memset(out, 0, sizeof(*out));
    out->kind = RT_TEX_IMAGE; out->width = image->width; out->height = image->height; out->texels = image->rgb.data();
    out->su = su; out->sv = sv; out->du = du; out->dv = dv;
}
template class ImageTexture<RGBSpectrum, Spectrum>;

// ------------------------------------------------------------------ shapes --
TriangleMesh::TriangleMesh(const Transform &o2w, int nTriangles, const int *vi, int nVertices, const Point3f *P, const Vector3f *S,
                           const Normal3f *N, const Point2f *UV, const int *fIndices)
    : nTriangles(nTriangles), nVertices(nVertices), vertexIndices(vi, vi + 3 * nTriangles) {
    p.reset(new Point3f[nVertices]);
    for (int i = 0; i < nVertices; ++i) p[i] = o2w(P[i]);
    if (UV) { uv.reset(new Point2f[nVertices]); memcpy(uv.get(), UV, nVertices * sizeof(Point2f)); }
    if (N) { n.reset(new Normal3f[nVertices]); for (int i = 0; i < nVertices; ++i) n[i] = o2w(N[i]); }
    if (S) { s.reset(new Vector3f[nVertices]); for (int i = 0; i < nVertices; ++i) s[i] = o2w(S[i]); }
    if (fIndices) faceIndices = std::vector<int>(fIndices, fIndices + nTriangles);
}
Float Triangle::Area() const {   // triangle.cpp:222-227
    const Point3f &p0 = mesh->p[v[0]], &p1 = mesh->p[v[1]], &p2 = mesh->p[v[2]];
    return 0.5 * Cross(p1 - p0, p2 - p0).Length();
}
std::vector<std::shared_ptr<Shape>> CreateTriangleMesh(const Transform *o2w, const Transform *w2o, bool rev, int nTriangles, const int *vi,
                                                       int nVertices, const Point3f *p, const Vector3f *s, const Normal3f *n,
                                                       const Point2f *uv, const int *faceIndices) {
    auto mesh = std::make_shared<TriangleMesh>(*o2w, nTriangles, vi, nVertices, p, s, n, uv, faceIndices);
    std::vector<std::shared_ptr<Shape>> tris; tris.reserve(nTriangles);
    for (int i = 0; i < nTriangles; ++i) tris.push_back(std::make_shared<Triangle>(o2w, w2o, rev, mesh, i));
    return tris;
}

BVHAccel::BVHAccel(std::vector<std::shared_ptr<Primitive>> p, int maxPrimsInNode, SplitMethod splitMethod)
    : maxPrimsInNode(maxPrimsInNode), splitMethod(splitMethod) {
    if (maxPrimsInNode < 1) throw std::runtime_error("BVHAccel: maxPrimsInNode must be at least 1");
    primitives = std::move(p);
}

// ---------------------------------------------------------------- flatten --
void FlatScene::Bind() {
    memset(&desc, 0, sizeof(desc));
    desc.n_prims = (int32_t)prim_kind.size();
    desc.prim_kind = prim_kind.data(); desc.prim_geom = prim_geom.data(); desc.prim_material = prim_material.data(); desc.prim_light = prim_light.data();
    desc.n_tris = (int32_t)(tri_p.size() / 9); desc.tri_p = tri_p.data(); desc.tri_uv = tri_uv.data();
    desc.tri_n = tri_n.empty() ? nullptr : tri_n.data(); desc.tri_has_n = tri_has_n.empty() ? nullptr : tri_has_n.data();
    desc.n_spheres = (int32_t)spheres.size(); desc.spheres = spheres.empty() ? nullptr : spheres.data();
    desc.n_materials = (int32_t)materials.size(); desc.materials = materials.data();
    desc.n_textures = (int32_t)textures.size(); desc.textures = textures.data();
    desc.n_lights = (int32_t)lights.size(); desc.lights = lights.empty() ? nullptr : lights.data();
    desc.n_nodes = (int32_t)nodes.size(); desc.bvh_nodes = nodes.empty() ? nullptr : nodes.data();
    desc.prim_order = prim_order.empty() ? nullptr : prim_order.data();
    desc.n_env_lights = (int32_t)env_lights.size(); desc.env_lights = env_lights.empty() ? nullptr : env_lights.data();
}

std::unique_ptr<FlatScene> FlattenScene(const Scene &scene) {
    auto fs = std::make_unique<FlatScene>();
    const Aggregate *agg = dynamic_cast<const Aggregate *>(scene.GetAggregate().get());
    if (!agg) throw std::runtime_error("FlattenScene: the scene aggregate must be a BVHAccel or SimpleAccel");
    std::map<const Material *, int> matIndex;
    std::map<const TextureBase *, int> texIndex;
    std::map<const Light *, int> lightIndex;
    for (size_t i = 0; i < scene.lights.size(); ++i) lightIndex[scene.lights[i].get()] = (int)i;
    fs->lights.resize(scene.lights.size());
    for (size_t i = 0; i < scene.lights.size(); ++i)
        if (const InfiniteAreaLight *env = dynamic_cast<const InfiniteAreaLight *>(scene.lights[i].get())) {
            rt_env_light e; env->Describe(&e);
            rt_light &L = fs->lights[i];
            L.prim = -1; L.kind = RT_LIGHT_INFINITE; L.env = (int32_t)fs->env_lights.size();
            fs->env_lights.push_back(e);
            FlatScene::EnvSource src; src.path = env->source; memcpy(src.L, env->scale, sizeof(src.L)); fs->env_sources.push_back(src);
        }
    bool anyNormals = false;
    auto texOf = [&](const TextureBase *t) -> int {
        if (!t) return -1;
        auto it = texIndex.find(t);
        if (it != texIndex.end()) return it->second;
        rt_texture d; t->Describe(&d);
        int id = (int)fs->textures.size(); fs->textures.push_back(d); fs->texture_sources.push_back(t->Source()); texIndex[t] = id; return id;
    };
    for (size_t i = 0; i < agg->primitives.size(); ++i) {
        const GeometricPrimitive *gp = dynamic_cast<const GeometricPrimitive *>(agg->primitives[i].get());
        if (!gp) throw std::runtime_error("FlattenScene: only GeometricPrimitive is supported");
        int kind = gp->shape->Kind();
        fs->prim_kind.push_back((uint8_t)kind);
        if (kind == RT_PRIM_TRIANGLE) {
            const Triangle *tri = static_cast<const Triangle *>(gp->shape.get());
            const TriangleMesh &m = *tri->mesh;
            fs->prim_geom.push_back((int32_t)(fs->tri_p.size() / 9));
            for (int k = 0; k < 3; ++k) { const Point3f &p = m.p[tri->v[k]]; fs->tri_p.push_back(p.x); fs->tri_p.push_back(p.y); fs->tri_p.push_back(p.z); }
            if (m.uv) for (int k = 0; k < 3; ++k) { fs->tri_uv.push_back(m.uv[tri->v[k]].x); fs->tri_uv.push_back(m.uv[tri->v[k]].y); }
            else { const float d[6] = {0, 0, 1, 0, 1, 1}; fs->tri_uv.insert(fs->tri_uv.end(), d, d + 6); }   // triangle.h:52-56
            if (m.n) { anyNormals = true; for (int k = 0; k < 3; ++k) { const Normal3f &n = m.n[tri->v[k]]; fs->tri_n.push_back(n.x); fs->tri_n.push_back(n.y); fs->tri_n.push_back(n.z); } fs->tri_has_n.push_back(1); }
            else { fs->tri_n.insert(fs->tri_n.end(), 9, 0.f); fs->tri_has_n.push_back(0); }
        } else {
            const Sphere *sp = static_cast<const Sphere *>(gp->shape.get());
            rt_sphere s; s.radius = sp->radius;
            memcpy(s.o2w, sp->ObjectToWorld->GetMatrix().m, sizeof(s.o2w));
            memcpy(s.w2o, sp->WorldToObject->GetMatrix().m, sizeof(s.w2o));
            fs->prim_geom.push_back((int32_t)fs->spheres.size()); fs->spheres.push_back(s);
        }
        const Material *mat = gp->material.get();
        if (!mat) throw std::runtime_error("FlattenScene: primitive without material");
        auto mi = matIndex.find(mat);
        if (mi == matIndex.end()) {
            rt_material rm; memset(&rm, 0, sizeof(rm)); rm.kind = mat->Kind();
            const TextureBase *slots[4]; int remap = 0; mat->Slots(slots, &remap);
            for (int k = 0; k < 4; ++k) rm.tex[k] = texOf(slots[k]);
            rm.remap_roughness = remap;
            int id = (int)fs->materials.size(); fs->materials.push_back(rm); mi = matIndex.emplace(mat, id).first;
        }
        fs->prim_material.push_back(mi->second);
        int li = -1;
        if (gp->areaLight) {
            auto lit = lightIndex.find(gp->areaLight.get());
            if (lit == lightIndex.end()) throw std::runtime_error("FlattenScene: area light of a primitive is not in scene.lights");
            li = lit->second;
            const DiffuseAreaLight *dl = static_cast<const DiffuseAreaLight *>(gp->areaLight.get());
            if (kind != RT_PRIM_TRIANGLE) throw std::runtime_error("FlattenScene: area lights must be triangles");
            rt_light &L = fs->lights[li];
            L.prim = (int32_t)i; L.Le[0] = dl->Lemit[0]; L.Le[1] = dl->Lemit[1]; L.Le[2] = dl->Lemit[2]; L.two_sided = dl->twoSided; L.area = dl->area;
            L.flip_normal = (gp->shape->reverseOrientation ^ gp->shape->transformSwapsHandedness) ? 1 : 0;
        }
        fs->prim_light.push_back(li);
    }
    if (!anyNormals) { fs->tri_n.clear(); fs->tri_has_n.clear(); }
    for (const rt_texture &t : fs->textures) (void)t;
    fs->Bind();
    int n = fs->desc.n_prims;
    if (n > 0) {
        std::vector<float> bounds(6 * (size_t)n);
        if (rt_prim_bounds(&fs->desc, bounds.data()) != 0) throw std::runtime_error("rt_prim_bounds failed");
        fs->nodes.resize(2 * (size_t)n); fs->prim_order.resize(n);
        int32_t nn = 0;
        const BVHAccel *bvh = dynamic_cast<const BVHAccel *>(agg);      // a SimpleAccel gets the default tree
        int maxPrims = bvh ? bvh->maxPrimsInNode : 1, method = bvh ? (int)bvh->splitMethod : RT_SPLIT_MIDDLE;
        if (rt_bvh_build_ex(n, bounds.data(), maxPrims, method, fs->nodes.data(), &nn, fs->prim_order.data()) != 0) throw std::runtime_error("rt_bvh_build_ex failed");
        fs->nodes.resize(nn);
        // InfiniteAreaLight::Preprocess (infinite.h:20-22, called by the Scene constructor, scene.cpp:6-13): the bounding sphere of
        // the scene's world bound (Bounds3::BoundingSphere, geometry.h:801-804), which is the root node's box
        for (rt_env_light &e : fs->env_lights) {
            Point3f lo(fs->nodes[0].bmin[0], fs->nodes[0].bmin[1], fs->nodes[0].bmin[2]), hi(fs->nodes[0].bmax[0], fs->nodes[0].bmax[1], fs->nodes[0].bmax[2]);
            Point3f c = (lo + hi) / 2;
            bool inside = c.x >= lo.x && c.x <= hi.x && c.y >= lo.y && c.y <= hi.y && c.z >= lo.z && c.z <= hi.z;
            e.world_center[0] = c.x; e.world_center[1] = c.y; e.world_center[2] = c.z;
            e.world_radius = inside ? (c - hi).Length() : 0;
        }
        fs->Bind();
    }
    return fs;
}

// ------------------------------------------------------------------ camera --
PerspectiveCamera::PerspectiveCamera(const Transform &c2w, const Bounds2f &sw, Float, Float focalDistance, Float fov, int w, int h)
    : Camera(c2w), CameraToScreen(Perspective(fov, focalDistance, 1000.f)), width(w), height(h) {
    ScreenToRaster = Scale(w, h, 1) * Scale(1 / (sw.pMax.x - sw.pMin.x), 1 / (sw.pMin.y - sw.pMax.y), 1) * Translate(Vector3f(-sw.pMin.x, -sw.pMax.y, 0));
    RasterToScreen = Inverse(ScreenToRaster);
    RasterToCamera = Inverse(CameraToScreen) * RasterToScreen;
}
bool PerspectiveCamera::Describe(rt_camera *out) const {
    memcpy(out->raster_to_camera, RasterToCamera.GetMatrix().m, sizeof(out->raster_to_camera));
    memcpy(out->camera_to_world, CameraToWorld.GetMatrix().m, sizeof(out->camera_to_world));
    out->width = width; out->height = height;
    return true;
}
PerspectiveCamera *CreatePerspectiveCamera(const Transform &cam2world, Float focaldistance, Float fov, int w, int h) {
    Float frame = (float)w / (float)h;
    Bounds2f screen;
    if (frame > 1.f) { screen.pMin.x = -frame; screen.pMax.x = frame; screen.pMin.y = -1.f; screen.pMax.y = 1.f; }
    else { screen.pMin.x = -1.f; screen.pMax.x = 1.f; screen.pMin.y = -1.f / frame; screen.pMax.y = 1.f / frame; }
    return new PerspectiveCamera(cam2world, screen, 0.0f, focaldistance, fov, w, h);
}

// -------------------------------------------------------------- integrator --
void PathIntegrator::Render(const Scene &scene, int image_width, int image_height) {
    auto t0 = std::chrono::steady_clock::now();
    std::unique_ptr<FlatScene> fs = FlattenScene(scene);
    rt_camera cam;
    if (!camera->Describe(&cam)) throw std::runtime_error("PathIntegrator::Render: unsupported camera");
    cam.width = image_width; cam.height = image_height;
    rt_render_opts o; memset(&o, 0, sizeof(o));
    o.strata_x = sampler->xSamples; o.strata_y = sampler->ySamples; o.sample_begin = 0; o.sample_end = (int)sampler->samplesPerPixel;
    o.max_depth = maxDepth; o.seed_lo = seedLo; o.seed_hi = seedHi; o.rr_threshold = rrThreshold;
    o.flags = (estimateWithLi ? RT_RENDER_LI : 0) | (sampleAllLights ? RT_RENDER_ALL_LIGHTS : 0);
    lastSum.assign((size_t)image_width * image_height * 3, 0.f);
    // Every GPU the process sees takes a share of the samples of every pixel (RT_AMD_DEVICES=n caps the number); one GPU renders alone.
    // Under a one-process-per-GPU launcher (torchrun and its like export LOCAL_RANK / WORLD_SIZE) every rank would otherwise drive all
    // GPUs at once, so such a process keeps to one device — its own, by LOCAL_RANK — unless RT_AMD_DEVICES says otherwise.
    int nDev = rt_device_count(), firstDev = 0;
    const bool launched = getenv("LOCAL_RANK") || getenv("WORLD_SIZE");
    if (launched) {
        if (const char *lr = getenv("LOCAL_RANK")) { int r = atoi(lr); if (r >= 0 && r < nDev) firstDev = r; }
        nDev = nDev > 0 ? 1 : 0;
    }
    if (const char *cap = getenv("RT_AMD_DEVICES")) { int c = atoi(cap); if (c >= 1 && c <= rt_device_count()) { nDev = c; if (c > 1) firstDev = 0; } }
    if (nDev > 1) {
        std::vector<int32_t> devs(nDev);
        for (int i = 0; i < nDev; ++i) devs[i] = i;
        rt_multi *m = nullptr;
        if (rt_multi_create(devs.data(), nDev, &fs->desc, &m) != 0) throw std::runtime_error(rt_multi_last_error(nullptr));
        int rc = rt_multi_render(m, &cam, &o, lastSum.data(), nullptr, &lastStats);
        std::string err = rc ? rt_multi_last_error(m) : "";
        rt_multi_destroy(m);
        if (rc) throw std::runtime_error(err);
    } else {
        rt_ctx *ctx = nullptr;
        if (rt_init(firstDev, &ctx) != 0) throw std::runtime_error(std::string("rt_init: ") + rt_last_error(nullptr));
        rt_scene *sc = nullptr;
        if (rt_scene_create(ctx, &fs->desc, &sc) != 0) { std::string e = rt_last_error(ctx); rt_shutdown(ctx); throw std::runtime_error("rt_scene_create: " + e); }
        int rc = rt_render(sc, &cam, &o, lastSum.data(), &lastStats);
        std::string err = rc ? rt_last_error(ctx) : "";
        rt_scene_destroy(sc); rt_shutdown(ctx);
        if (rc) throw std::runtime_error("rt_render: " + err);
    }
    std::vector<uint8_t> px((size_t)image_width * image_height * 3);
    rt_tonemap_u8(lastSum.data(), image_width, image_height, (int)sampler->samplesPerPixel, px.data());
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << secs << std::endl;
    if (!outputPath.empty()) WritePNG(outputPath, image_width, image_height, px.data());
    std::cerr << "\nDone.\n";
}

bool WritePNG(const std::string &path, int w, int h, const uint8_t *rgb) {
    std::vector<uint8_t> raw((size_t)(3 * w + 1) * h);
    for (int y = 0; y < h; ++y) { raw[(size_t)(3 * w + 1) * y] = 0; memcpy(&raw[(size_t)(3 * w + 1) * y + 1], rgb + (size_t)3 * w * y, 3 * (size_t)w); }
    uLongf clen = compressBound(raw.size()); std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), raw.size(), 6) != Z_OK) return false;
    FILE *f = fopen(path.c_str(), "wb"); if (!f) return false;
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10}; fwrite(sig, 1, 8, f);
    auto chunk = [&](const char *type, const uint8_t *data, uint32_t len) {
        uint8_t hdr[8] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len, (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3]};
        fwrite(hdr, 1, 8, f); if (len) fwrite(data, 1, len, f);
        uLong crc = crc32(0L, hdr + 4, 4); if (len) crc = crc32(crc, data, len);
        uint8_t c[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc}; fwrite(c, 1, 4, f);
    };
    uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w, (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h, 8, 2, 0, 0, 0};
    chunk("IHDR", ihdr, 13); chunk("IDAT", comp.data(), (uint32_t)clen); chunk("IEND", nullptr, 0);
    fclose(f); return true;
}

// ------------------------------------------------------------ glTF loader --
namespace {
struct JVal {
    enum T { Null, Bool, Num, Str, Arr, Obj } t = Null;
    double num = 0; bool b = false; std::string s; std::vector<JVal> a; std::vector<std::pair<std::string, JVal>> o;
    const JVal *get(const std::string &k) const { for (auto &kv : o) if (kv.first == k) return &kv.second; return nullptr; }
    int geti(const std::string &k, int def = -1) const { const JVal *v = get(k); return (v && v->t == Num) ? (int)v->num : def; }
};
struct JParser {
    const std::string &s; size_t p = 0; explicit JParser(const std::string &s) : s(s) {}
    void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\n' || s[p] == '\r' || s[p] == '\t')) ++p; }
    JVal parse() {
        ws(); JVal v; if (p >= s.size()) throw std::runtime_error("json: eof");
        char c = s[p];
        if (c == '{') { v.t = JVal::Obj; ++p; ws(); if (s[p] == '}') { ++p; return v; }
            while (true) { ws(); JVal k = parse(); ws(); if (s[p++] != ':') throw std::runtime_error("json: ':'"); JVal x = parse(); v.o.emplace_back(k.s, std::move(x)); ws(); if (s[p] == ',') { ++p; continue; } if (s[p] == '}') { ++p; break; } throw std::runtime_error("json: '}'"); } }
        else if (c == '[') { v.t = JVal::Arr; ++p; ws(); if (s[p] == ']') { ++p; return v; }
            while (true) { v.a.push_back(parse()); ws(); if (s[p] == ',') { ++p; continue; } if (s[p] == ']') { ++p; break; } throw std::runtime_error("json: ']'"); } }
        else if (c == '"') { v.t = JVal::Str; ++p; while (p < s.size() && s[p] != '"') { if (s[p] == '\\' && p + 1 < s.size()) { ++p; char e = s[p]; v.s += (e == 'n' ? '\n' : e == 't' ? '\t' : e); } else v.s += s[p]; ++p; } ++p; }
        else if (!s.compare(p, 4, "true")) { v.t = JVal::Bool; v.b = true; p += 4; }
        else if (!s.compare(p, 5, "false")) { v.t = JVal::Bool; p += 5; }
        else if (!s.compare(p, 4, "null")) { p += 4; }
        else { size_t e = p; while (e < s.size() && (isdigit((unsigned char)s[e]) || s[e] == '-' || s[e] == '+' || s[e] == '.' || s[e] == 'e' || s[e] == 'E')) ++e; v.t = JVal::Num; v.num = atof(s.substr(p, e - p).c_str()); p = e; }
        return v;
    }
};
std::vector<uint8_t> ReadFileBytes(const std::string &path) {
    std::ifstream f(path, std::ios::binary); if (!f) return {};
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
}  // namespace

namespace {
std::vector<uint8_t> Base64(const std::string &t, size_t from) {
    std::vector<uint8_t> out; uint32_t acc = 0; int bits = 0;
    for (size_t i = from; i < t.size(); ++i) {
        char c = t[i]; int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A'; else if (c >= 'a' && c <= 'z') v = c - 'a' + 26; else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62; else if (c == '/' || c == '_') v = 63; else continue;
        acc = (acc << 6) | (uint32_t)v; bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)(acc >> bits)); }
    }
    return out;
}
uint32_t le32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
// Column-major 4x4 (glTF) helpers for node transforms, in double.
struct M4 { double m[16]; };
M4 M4Identity() { M4 r; for (int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.0 : 0.0; return r; }
M4 M4Mul(const M4 &a, const M4 &b) {
    M4 r;
    for (int c = 0; c < 4; ++c) for (int rr = 0; rr < 4; ++rr) { double v = 0; for (int k = 0; k < 4; ++k) v += a.m[k * 4 + rr] * b.m[c * 4 + k]; r.m[c * 4 + rr] = v; }
    return r;
}
M4 NodeLocal(const JVal &nd) {       // glTF 2.0: `matrix`, or translation * rotation * scale
    M4 r = M4Identity();
    if (const JVal *mx = nd.get("matrix")) { if (mx->a.size() == 16) for (int i = 0; i < 16; ++i) r.m[i] = mx->a[i].num; return r; }
    double t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, sc[3] = {1, 1, 1};
    if (const JVal *v = nd.get("translation")) if (v->a.size() == 3) for (int i = 0; i < 3; ++i) t[i] = v->a[i].num;
    if (const JVal *v = nd.get("rotation")) if (v->a.size() == 4) for (int i = 0; i < 4; ++i) q[i] = v->a[i].num;
    if (const JVal *v = nd.get("scale")) if (v->a.size() == 3) for (int i = 0; i < 3; ++i) sc[i] = v->a[i].num;
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w),      // column 0
                   2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w),      // column 1
                   2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y)};     // column 2
    for (int c = 0; c < 3; ++c) for (int rr = 0; rr < 3; ++rr) r.m[c * 4 + rr] = R[c * 3 + rr] * sc[c];
    r.m[12] = t[0]; r.m[13] = t[1]; r.m[14] = t[2];
    return r;
}
}  // namespace

// Reads .gltf (external or base64 buffers and images) and .glb (binary container, images in buffer views).  What is taken from
// the file and how follows the reference's Assimp-based loader (modelload.cpp:7-125, import flags :8-9): positions and normals
// with z negated, v flipped, the node hierarchy walked meshes-first — and, like it, node transforms are NOT applied unless
// applyNodeTransforms is set (an extension: the node's matrix / TRS chain is then applied to the raw positions before the z flip).
void ModelLoad::loadModel(string path, const Transform &ObjectToWorld) {
    path = NormalizePath(path);
    std::vector<uint8_t> text = ReadFileBytes(path);
    if (text.empty()) { std::cout << "ERROR::GLTF::cannot read " << path << std::endl; return; }
    size_t slash = path.find_last_of('/');
    directory = slash == std::string::npos ? "." : path.substr(0, slash);
    const std::string fileName = BaseName(path);
    std::vector<uint8_t> glbBin; bool isGlb = false;
    std::string js;
    if (text.size() >= 20 && le32(text.data()) == 0x46546C67u) {         // "glTF": 12-byte header, then chunks (length, type, data)
        isGlb = true;
        size_t pos = 12, total = std::min<size_t>(text.size(), le32(text.data() + 8));
        while (pos + 8 <= total) {
            uint32_t len = le32(&text[pos]), type = le32(&text[pos + 4]);
            if (pos + 8 + (size_t)len > total) break;
            if (type == 0x4E4F534Au && js.empty()) js.assign(text.begin() + pos + 8, text.begin() + pos + 8 + len);
            else if (type == 0x004E4942u && glbBin.empty()) glbBin.assign(text.begin() + pos + 8, text.begin() + pos + 8 + len);
            pos += 8 + (((size_t)len + 3) & ~(size_t)3);
        }
        if (js.empty()) { std::cout << "ERROR::GLB::no JSON chunk in " << path << std::endl; return; }
    } else js.assign(text.begin(), text.end());
    JVal root;
    try { root = JParser(js).parse(); } catch (std::exception &e) { std::cout << "ERROR::GLTF::" << e.what() << std::endl; return; }
    std::vector<std::vector<uint8_t>> buffers;
    if (const JVal *bs = root.get("buffers")) for (const JVal &b : bs->a) {
        const JVal *uri = b.get("uri");
        if (!uri) buffers.push_back(isGlb && buffers.empty() ? glbBin : std::vector<uint8_t>());
        else if (uri->s.compare(0, 5, "data:") == 0) { size_t c = uri->s.find(','); buffers.push_back(c == std::string::npos ? std::vector<uint8_t>() : Base64(uri->s, c + 1)); }
        else buffers.push_back(ReadFileBytes(directory + "/" + uri->s));
    }
    const JVal *views = root.get("bufferViews"), *accs = root.get("accessors"), *jmeshes = root.get("meshes");
    if (!views || !accs || !jmeshes) { std::cout << "ERROR::GLTF::incomplete scene" << std::endl; return; }
    // Start of an accessor's data, or nullptr unless every element lies inside both its buffer view and its buffer.
    auto accessor = [&](int idx, int *count, int *ctype, int *ncomp, int *stride) -> const uint8_t * {
        if (idx < 0 || idx >= (int)accs->a.size()) return nullptr;
        const JVal &a = accs->a[idx];
        const int vi = a.geti("bufferView", -1);
        if (vi < 0 || vi >= (int)views->a.size()) return nullptr;
        const JVal &v = views->a[vi];
        const int bi = v.geti("buffer", 0);
        if (bi < 0 || bi >= (int)buffers.size()) return nullptr;
        *count = a.geti("count", 0); *ctype = a.geti("componentType", 0);
        if (*count <= 0) return nullptr;
        const std::string ty = a.get("type") ? a.get("type")->s : "SCALAR";
        *ncomp = ty == "SCALAR" ? 1 : ty == "VEC2" ? 2 : ty == "VEC3" ? 3 : 4;
        int csize = (*ctype == 5126 || *ctype == 5125) ? 4 : (*ctype == 5123 || *ctype == 5122) ? 2 : 1;
        int bs = v.geti("byteStride", 0);
        if (bs < 0 || v.geti("byteOffset", 0) < 0 || a.geti("byteOffset", 0) < 0 || v.geti("byteLength", 0) < 0) return nullptr;
        *stride = bs ? bs : csize * *ncomp;
        const std::vector<uint8_t> &buf = buffers[bi];
        const size_t voff = (size_t)v.geti("byteOffset", 0), off = voff + (size_t)a.geti("byteOffset", 0);
        const size_t vend = v.get("byteLength") ? voff + (size_t)v.geti("byteLength", 0) : buf.size();
        const size_t last = off + (size_t)*stride * (size_t)(*count - 1) + (size_t)csize * (size_t)*ncomp;    // one past the last element's last byte
        if (last > std::min(buf.size(), vend)) return nullptr;
        return buf.data() + off;
    };
    auto texName = [&](const JVal *matl, const char *slot) -> std::string {
        if (!matl) return "";
        const JVal *pbr = matl->get("pbrMetallicRoughness"); if (!pbr) return "";
        const JVal *t = pbr->get(slot); if (!t) return "";
        const JVal *texs = root.get("textures"), *imgs = root.get("images"); if (!texs || !imgs) return "";
        int ti = t->geti("index"); if (ti < 0 || ti >= (int)texs->a.size()) return "";
        int src = texs->a[ti].geti("source"); if (src < 0 || src >= (int)imgs->a.size()) return "";
        const JVal &im = imgs->a[src];
        const JVal *uri = im.get("uri");
        if (uri && uri->s.compare(0, 5, "data:") != 0) return uri->s;
        // embedded image (buffer view or data URI): decode now and make it known under a name of its own
        std::string name = fileName + "#image" + std::to_string(src);
        std::vector<uint8_t> bytes;
        if (uri) { size_t c = uri->s.find(','); if (c != std::string::npos) bytes = Base64(uri->s, c + 1); }
        else {
            int vi = im.geti("bufferView");
            if (vi >= 0 && vi < (int)views->a.size()) {
                const JVal &v = views->a[vi]; int bi = v.geti("buffer", 0);
                size_t off = (size_t)v.geti("byteOffset", 0), len = (size_t)v.geti("byteLength", 0);
                if (bi >= 0 && bi < (int)buffers.size() && off + len <= buffers[bi].size()) bytes.assign(buffers[bi].begin() + off, buffers[bi].begin() + off + len);
            }
        }
        int w = 0, h = 0; std::vector<uint8_t> rgb;
        if (bytes.empty() || !DecodeImageRGB8(bytes.data(), bytes.size(), &w, &h, &rgb)) return "";
        RegisterImage(name, w, h, rgb.data());
        return name;
    };
    auto addMesh = [&](const JVal &mesh, const M4 &world) {
        const bool xf = applyNodeTransforms;
        // normals go through the cofactor matrix of the upper 3x3 (= inverse transpose up to a positive or negative scale)
        const double *a = world.m;
        const double cof[9] = {a[5] * a[10] - a[6] * a[9], a[6] * a[8] - a[4] * a[10], a[4] * a[9] - a[5] * a[8],
                               a[2] * a[9] - a[1] * a[10], a[0] * a[10] - a[2] * a[8], a[1] * a[8] - a[0] * a[9],
                               a[1] * a[6] - a[2] * a[5], a[2] * a[4] - a[0] * a[6], a[0] * a[5] - a[1] * a[4]};
        const double det = a[0] * cof[0] + a[1] * cof[1] + a[2] * cof[2];
        const JVal *prims = mesh.get("primitives"); if (!prims) return;
        for (const JVal &pr : prims->a) {
            const JVal *attr = pr.get("attributes"); if (!attr) continue;
            int n = 0, ct = 0, nc = 0, st = 0;
            const uint8_t *pos = accessor(attr->geti("POSITION"), &n, &ct, &nc, &st);
            if (!pos || ct != 5126 || nc != 3) continue;
            int nVerts = n, pst = st;
            std::vector<Point3f> P(nVerts); std::vector<Normal3f> N; std::vector<Point2f> UV;
            for (int i = 0; i < nVerts; ++i) {
                float v[3]; memcpy(v, pos + (size_t)i * pst, 12);
                if (xf) { double x = v[0], y = v[1], z = v[2]; v[0] = (float)(a[0] * x + a[4] * y + a[8] * z + a[12]); v[1] = (float)(a[1] * x + a[5] * y + a[9] * z + a[13]); v[2] = (float)(a[2] * x + a[6] * y + a[10] * z + a[14]); }
                P[i] = Point3f(v[0], v[1], -v[2]);
            }
            const uint8_t *nor = accessor(attr->geti("NORMAL"), &n, &ct, &nc, &st);
            if (nor && ct == 5126 && nc == 3 && n == nVerts) {
                N.resize(nVerts);
                for (int i = 0; i < nVerts; ++i) {
                    float v[3]; memcpy(v, nor + (size_t)i * st, 12);
                    if (xf) {
                        double x = v[0], y = v[1], z = v[2], sgn = det < 0 ? -1.0 : 1.0;
                        double nx = sgn * (cof[0] * x + cof[3] * y + cof[6] * z), ny = sgn * (cof[1] * x + cof[4] * y + cof[7] * z), nz = sgn * (cof[2] * x + cof[5] * y + cof[8] * z);
                        double len = std::sqrt(nx * nx + ny * ny + nz * nz);
                        if (len > 0) { v[0] = (float)(nx / len); v[1] = (float)(ny / len); v[2] = (float)(nz / len); }
                    }
                    N[i] = Normal3f(v[0], v[1], -v[2]);
                }
            }
            const uint8_t *tc = accessor(attr->geti("TEXCOORD_0"), &n, &ct, &nc, &st);
            if (tc && ct == 5126 && nc == 2 && n == nVerts) { UV.resize(nVerts); for (int i = 0; i < nVerts; ++i) { float v[2]; memcpy(v, tc + (size_t)i * st, 8); UV[i] = Point2f(v[0], 1 - v[1]); } }
            std::vector<int> idx;
            int ia = pr.geti("indices");
            if (ia >= 0) {
                const uint8_t *ip = accessor(ia, &n, &ct, &nc, &st); if (!ip) continue;
                idx.resize(n);
                for (int i = 0; i < n; ++i) {
                    if (ct == 5123) { uint16_t v; memcpy(&v, ip + (size_t)i * st, 2); idx[i] = v; }
                    else if (ct == 5125) { uint32_t v; memcpy(&v, ip + (size_t)i * st, 4); idx[i] = (int)v; }
                    else idx[i] = ip[(size_t)i * st];
                }
            } else { idx.resize(nVerts); for (int i = 0; i < nVerts; ++i) idx[i] = i; }
            idx.resize(idx.size() / 3 * 3);
            bool idxOk = true;
            for (int v : idx) if (v < 0 || v >= nVerts) { idxOk = false; break; }
            if (!idxOk) { std::cout << "ERROR::GLTF::a primitive's indices point outside its vertices; skipped" << std::endl; continue; }
            int nTris = (int)idx.size() / 3;
            meshes.push_back(std::make_shared<TriangleMesh>(ObjectToWorld, nTris, idx.data(), nVerts, P.data(), nullptr, N.empty() ? nullptr : N.data(),
                                                            UV.empty() ? nullptr : UV.data(), nullptr));
            const JVal *mats = root.get("materials"); int mi = pr.geti("material");
            const JVal *matl = (mats && mi >= 0 && mi < (int)mats->a.size()) ? &mats->a[mi] : nullptr;
            diffTexName.push_back(texName(matl, "baseColorTexture"));
            specTexName.push_back(texName(matl, "metallicRoughnessTexture"));
        }
    };
    // node walk: meshes of a node first, then its children (modelload.cpp:18-25); node transforms are not applied.
    const JVal *nodes = root.get("nodes"), *scenes = root.get("scenes");
    std::vector<int> stack;
    if (nodes && scenes && !scenes->a.empty()) {
        const JVal &sc = scenes->a[std::max(0, root.geti("scene", 0))];
        std::function<void(int, const M4 &, int)> walk = [&](int ni, const M4 &parent, int depth) {
            if (ni < 0 || ni >= (int)nodes->a.size() || depth > 64) return;
            const JVal &nd = nodes->a[ni];
            M4 world = M4Mul(parent, NodeLocal(nd));
            int mi = nd.geti("mesh"); if (mi >= 0 && mi < (int)jmeshes->a.size()) addMesh(jmeshes->a[mi], world);
            if (const JVal *ch = nd.get("children")) for (const JVal &c : ch->a) walk((int)c.num, world, depth + 1);
        };
        if (const JVal *rn = sc.get("nodes")) for (const JVal &r : rn->a) walk((int)r.num, M4Identity(), 0);
    } else for (const JVal &m : jmeshes->a) addMesh(m, M4Identity());
}

void ModelLoad::buildNoTextureModel(Transform &o2w, vector<shared_ptr<Primitive>> &prims, shared_ptr<Material> material) {
    keep.emplace_back(new Transform(o2w)); const Transform *po = keep.back().get();
    keep.emplace_back(new Transform(Inverse(o2w))); const Transform *pi = keep.back().get();
    for (size_t i = 0; i < meshes.size(); ++i)
        for (int j = 0; j < meshes[i]->nTriangles; ++j)
            prims.push_back(make_shared<GeometricPrimitive>(make_shared<Triangle>(po, pi, false, meshes[i], j), material, nullptr));
}
void ModelLoad::buildTextureModel(Transform &o2w, vector<shared_ptr<Primitive>> &prims) {
    keep.emplace_back(new Transform(o2w)); const Transform *po = keep.back().get();
    keep.emplace_back(new Transform(Inverse(o2w))); const Transform *pi = keep.back().get();
    for (size_t i = 0; i < meshes.size(); ++i) {
        string f1 = directory + "/" + diffTexName[i], f2 = directory + "/" + specTexName[i];
        shared_ptr<Material> material = (diffTexName[i] != "" && specTexName[i] != "") ? getMetalRoughnessMaterial(f1, f2) : getDiffuseMaterial(f1);
        for (int j = 0; j < meshes[i]->nTriangles; ++j)
            prims.push_back(make_shared<GeometricPrimitive>(make_shared<Triangle>(po, pi, false, meshes[i], j), material, nullptr));
    }
}
shared_ptr<Material> getDiffuseMaterial(string filename) {
    shared_ptr<Texture<Spectrum>> Kt = make_shared<ImageTexture<RGBSpectrum, Spectrum>>(make_unique<UVMapping2D>(1.f, 1.f, 0.f, 0.f), filename, false, 8.f, ImageWrap::Repeat, 1.f, false);
    return make_shared<MatteMaterial>(Kt, make_shared<ConstantTexture<Float>>(0.0f));
}
shared_ptr<Material> getMetalRoughnessMaterial(string diffFilename, string specFilename) {
    shared_ptr<Texture<Spectrum>> base = make_shared<ImageTexture<Spectrum, Spectrum>>(make_unique<UVMapping2D>(1.f, 1.f, 0.f, 0.f), diffFilename, false, 8.f, ImageWrap::Repeat, 1.f, false);
    shared_ptr<Texture<Spectrum>> mr = make_shared<ImageTexture<Spectrum, Spectrum>>(make_unique<UVMapping2D>(1.f, 1.f, 0.f, 0.f), specFilename, false, 8.f, ImageWrap::Repeat, 1.f, false);
    return make_shared<MetalRoughnessMaterial>(base, mr);
}

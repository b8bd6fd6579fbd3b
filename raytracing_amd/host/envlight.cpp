// envlight.cpp — host side of the infinite area light: Radiance .hdr reader / writer, the MIP pyramid and the piecewise-constant
// 2D distribution that the reference's InfiniteAreaLight constructor builds (include/core/infinite.cpp:4-71), in its operation
// order, so that the tables handed to the device are the reference's bit for bit.
#include "rt_host.h"

#include <fstream>

namespace {

int ModI(int a, int b) { int r = a - (a / b) * b; return r < 0 ? r + b : r; }

// One level of MIPMap's pyramid (mipmap.h:125-139): RGB floats, Repeat wrap (MIPMap's default, mipmap.h:22-25).
struct Level {
    int w = 0, h = 0; std::vector<float> rgb;
    const float *texel(int s, int t) const { return &rgb[3 * ((size_t)ModI(t, h) * w + ModI(s, w))]; }      // MIPMap::Texel, mipmap.h:152-171
};
// MIPMap::triangle (mipmap.h:190-200)
void Triangle(const Level &l, float st0, float st1, float out[3]) {
    float s = st0 * l.w - 0.5f, t = st1 * l.h - 0.5f;
    int s0 = (int)std::floor(s), t0 = (int)std::floor(t);
    float ds = s - s0, dt = t - t0;
    const float *a = l.texel(s0, t0), *b = l.texel(s0, t0 + 1), *c = l.texel(s0 + 1, t0), *d = l.texel(s0 + 1, t0 + 1);
    for (int k = 0; k < 3; ++k) out[k] = (1 - ds) * (1 - dt) * a[k] + (1 - ds) * dt * b[k] + ds * (1 - dt) * c[k] + ds * dt * d[k];
}
int Log2Int(uint32_t v) { return v ? 31 - __builtin_clz(v) : 0; }                                           // pbrt.h:188-192

// MIPMap::Lookup(st, width) (mipmap.h:173-187): trilinear between two pyramid levels.
void Lookup(const std::vector<Level> &pyr, float st0, float st1, float width, float out[3]) {
    const int nLevels = (int)pyr.size();
    const float invLog2 = 1.442695040888963387004650940071;
    float level = nLevels - 1 + std::log(std::max(width, (float)1e-8)) * invLog2;
    if (level < 0) { Triangle(pyr[0], st0, st1, out); return; }
    if (level >= nLevels - 1) { const float *t = pyr[nLevels - 1].texel(0, 0); out[0] = t[0]; out[1] = t[1]; out[2] = t[2]; return; }
    int iLevel = (int)std::floor(level);
    float delta = level - iLevel, a[3], b[3];
    Triangle(pyr[iLevel], st0, st1, a); Triangle(pyr[iLevel + 1], st0, st1, b);
    for (int k = 0; k < 3; ++k) out[k] = (1 - delta) * a[k] + delta * b[k];                                  // Lerp, spectrum.h:244-246
}

// Distribution1D's constructor (sampling.h:16-28) over f[0..n): cdf (n + 1 values) and funcInt.
float BuildCdf(const float *f, int n, float *cdf) {
    cdf[0] = 0;
    for (int i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + f[i - 1] / n;
    float funcInt = cdf[n];
    if (funcInt == 0) { for (int i = 1; i < n + 1; ++i) cdf[i] = float(i) / float(n); }
    else { for (int i = 1; i < n + 1; ++i) cdf[i] /= funcInt; }
    return funcInt;
}

}  // namespace

// stb_image's Radiance reader restated (stbi__hdr_load / stbi__hdr_convert): header lines up to an empty one, of which
// FORMAT=32-bit_rle_rgbe must be present; "-Y h +X w"; scanlines flat, or run-length encoded when they start with 2 2 and a
// width between 8 and 32767.
bool ReadHDR(const std::string &path, int *width, int *height, std::vector<float> *rgb, bool flip) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    size_t p = 0;
    auto line = [&]() { std::string s; while (p < d.size() && d[p] != '\n') s += (char)d[p++]; if (p < d.size()) ++p; return s; };
    std::string first = line();
    if (first != "#?RADIANCE" && first != "#?RGBE") return false;
    bool fmt = false;
    while (p < d.size()) { std::string s = line(); if (s.empty()) break; if (s == "FORMAT=32-bit_rle_rgbe") fmt = true; }
    if (!fmt) return false;
    std::string res = line();
    int w = 0, h = 0;
    if (sscanf(res.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0 || (size_t)w * h > ((size_t)1 << 28)) return false;
    std::vector<uint8_t> rgbe((size_t)w * h * 4);
    // stbi__hdr_load: images narrower than 8 or wider than 32767 pixels are flat; otherwise every scanline starts with 2 2 and its
    // length (high bit clear), which must equal the width ("invalid decoded scanline length" if not).  A first scanline without that
    // header means the whole file is flat, its first four bytes being pixel 0; the same thing on a later scanline sends stb's loop back to
    // pixel 1 of row 0 (its own comment: "this makes no sense") — such a file is refused here.
    bool flat = w < 8 || w >= 32768;
    for (int y = 0; y < h && !flat; ++y) {
        uint8_t *row = &rgbe[(size_t)y * w * 4];
        if (p + 4 > d.size()) return false;
        if (d[p] != 2 || d[p + 1] != 2 || (d[p + 2] & 0x80)) {
            if (y == 0) { flat = true; break; }
            return false;
        }
        if (((d[p + 2] << 8) | d[p + 3]) != w) return false;
        p += 4;
        for (int k = 0; k < 4; ++k) {
            int i = 0;
            while (i < w) {
                if (p >= d.size()) return false;
                int count = d[p++];
                if (count > 128) { count -= 128; if (p >= d.size() || i + count > w) return false; uint8_t v = d[p++]; for (int z = 0; z < count; ++z) row[4 * (i++) + k] = v; }
                else { if (count == 0 || p + count > d.size() || i + count > w) return false; for (int z = 0; z < count; ++z) row[4 * (i++) + k] = d[p++]; }
            }
        }
    }
    if (flat) {
        if (p + (size_t)4 * w * h > d.size()) return false;
        memcpy(rgbe.data(), &d[p], (size_t)4 * w * h);
    }
    rgb->assign((size_t)w * h * 3, 0.f);
    for (int y = 0; y < h; ++y) {
        const uint8_t *row = &rgbe[(size_t)(flip ? h - 1 - y : y) * w * 4];
        float *out = &(*rgb)[(size_t)y * w * 3];
        for (int x = 0; x < w; ++x) {
            const uint8_t *in = row + 4 * x;
            if (in[3] != 0) {
                float f1 = (float)ldexp(1.0f, in[3] - (int)(128 + 8));
                out[3 * x] = in[0] * f1; out[3 * x + 1] = in[1] * f1; out[3 * x + 2] = in[2] * f1;
            }
        }
    }
    *width = w; *height = h;
    return true;
}

bool WriteHDR(const std::string &path, int w, int h, const float *rgb) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n", h, w);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        float r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2], m = std::max(r, std::max(g, b));
        uint8_t px[4] = {0, 0, 0, 0};
        if (m >= 1e-32f) { int e; float sc = std::frexp(m, &e) * 256.0f / m; px[0] = (uint8_t)(r * sc); px[1] = (uint8_t)(g * sc); px[2] = (uint8_t)(b * sc); px[3] = (uint8_t)(e + 128); }
        fwrite(px, 1, 4, f);
    }
    return fclose(f) == 0;
}

InfiniteAreaLight::InfiniteAreaLight(const Transform &l2w, const Spectrum &L, int, const std::string &texmap)
    : LightToWorld(l2w), WorldToLight(Inverse(l2w)), source(texmap) {
    scale[0] = L[0]; scale[1] = L[1]; scale[2] = L[2];
    auto img = std::make_shared<ImageData>();
    std::vector<float> data; int w = 0, h = 0;
    std::string path = texmap; for (char &c : path) if (c == '\\') c = '/';
    if (texmap != "" && ReadHDR(path, &w, &h, &data, true)) {
        img->width = w; img->height = h; img->rgb.resize((size_t)w * h * 3);
        for (size_t i = 0; i < (size_t)w * h; ++i)
            for (int c = 0; c < 3; ++c) { float r = L[c] * data[3 * i + c]; img->rgb[3 * i + c] = r + std::sqrt(r); }      // infinite.cpp:23-27
    } else {
        if (texmap != "") printf("InfiniteAreaLight: cannot read \"%s\"; using the constant radiance\n", texmap.c_str());
        img->width = img->height = 1; img->rgb = {L[0], L[1], L[2]};                                                          // infinite.cpp:43-47
    }
    img = ResampleToPow2(img);                                                                                                // mipmap.h:70-121
    std::vector<Level> pyr(1 + Log2Int((uint32_t)std::max(img->width, img->height)));
    pyr[0].w = img->width; pyr[0].h = img->height; pyr[0].rgb = img->rgb;
    for (size_t i = 1; i < pyr.size(); ++i) {                                                                                 // mipmap.h:128-139
        Level &dst = pyr[i]; const Level &src = pyr[i - 1];
        dst.w = std::max(1, src.w / 2); dst.h = std::max(1, src.h / 2); dst.rgb.resize((size_t)dst.w * dst.h * 3);
        for (int t = 0; t < dst.h; ++t)
            for (int s = 0; s < dst.w; ++s) {
                const float *a = src.texel(2 * s, 2 * t), *b = src.texel(2 * s + 1, 2 * t), *c = src.texel(2 * s, 2 * t + 1), *d = src.texel(2 * s + 1, 2 * t + 1);
                for (int k = 0; k < 3; ++k) dst.rgb[3 * ((size_t)t * dst.w + s) + k] = .25f * (((a[k] + b[k]) + c[k]) + d[k]);
            }
    }
    width = pyr[0].w; height = pyr[0].h; texels = pyr[0].rgb;
    // scalar image at twice the resolution: luminance of a filtered lookup, times sin(theta) (infinite.cpp:53-66)
    nu = 2 * width; nv = 2 * height;
    func.resize((size_t)nu * nv);
    float fwidth = 0.5f / std::min(nu, nv);
    for (int64_t v = 0; v < nv; ++v) {
        float vp = (v + .5f) / (float)nv;
        float sinTheta = std::sin(Pi * (v + .5f) / nv);
        for (int u = 0; u < nu; ++u) {
            float up = (u + .5f) / (float)nu, c[3];
            Lookup(pyr, up, vp, fwidth, c);
            float y = 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2];                                                // RGBSpectrum::y, spectrum.h:238-241
            y *= sinTheta;
            func[u + v * nu] = y;
        }
    }
    // Distribution2D (sampling.cpp:113-123)
    cdf.resize((size_t)nv * (nu + 1)); funcInt.resize(nv); marginalCdf.resize(nv + 1);
    for (int v = 0; v < nv; ++v) funcInt[v] = BuildCdf(&func[(size_t)v * nu], nu, &cdf[(size_t)v * (nu + 1)]);
    marginalInt = BuildCdf(funcInt.data(), nv, marginalCdf.data());
}

void InfiniteAreaLight::Describe(rt_env_light *o) const {
    memset(o, 0, sizeof(*o));
    memcpy(o->light_to_world, LightToWorld.GetMatrix().m, 64); memcpy(o->world_to_light, WorldToLight.GetMatrix().m, 64);
    o->width = width; o->height = height; o->texels = texels.data();
    o->nu = nu; o->nv = nv; o->func = func.data(); o->cdf = cdf.data(); o->func_int = funcInt.data(); o->marginal_cdf = marginalCdf.data();
    o->marginal_int = marginalInt;
}

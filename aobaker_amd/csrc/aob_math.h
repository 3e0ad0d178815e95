// aob_math.h — device-side arithmetic of the bake path (gfx950).
//
// Arithmetic contract: IEEE-754 binary32, round-to-nearest-even,
// no contraction except where fmaf is written (the library is compiled with
// -ffp-contract=off), correctly rounded division and sqrt (hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt), denormals kept.  The evaluation
// order of every expression below is part of the contract: the parity tests
// expect bit-identical rays, hits and sums.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aob {

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 make_f3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 sub(f3 a, f3 b) { return make_f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ f3 cross(f3 a, f3 b)
{
    return make_f3(fmaf(a.y, b.z, -(a.z * b.y)),
                   fmaf(a.z, b.x, -(a.x * b.z)),
                   fmaf(a.x, b.y, -(a.y * b.x)));
}
__device__ __forceinline__ f3 normalize(f3 v)
{
    float l = sqrtf(dot(v, v));
    return make_f3(v.x / l, v.y / l, v.z / l);
}
__device__ __forceinline__ bool finite3(f3 v) { return isfinite(v.x) && isfinite(v.y) && isfinite(v.z); }

// D3D ftoi — the (int) casts of ProcessIntersectionsComputeShader.hlsl:87-88:
// truncate toward zero, NaN -> 0, saturate.
__device__ __forceinline__ int32_t ftoi(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}

enum { FP16_OFF = 0, FP16_RTZ = 1, FP16_RNE = 2 };

// float32 -> float16 bits with the stated rounding, integer-only (the RGBA16F
// NormalV render target of the reference, BakeEngine.cpp:446).
__device__ __forceinline__ uint32_t f32_to_f16(float v, int mode)
{
    uint32_t u = __float_as_uint(v);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return sign | 0x7e00u;
    if (a == 0x7f800000u) return sign | 0x7c00u;
    if (a < 0x00800000u) return sign;
    int32_t e = (int32_t)(a >> 23) - 127;
    uint32_t m = (a & 0x7fffffu) | 0x800000u;
    if (e > 15) return sign | (mode == FP16_RTZ ? 0x7bffu : 0x7c00u);
    uint32_t shift, base;
    if (e >= -14) { shift = 13; base = (uint32_t)(e + 15) << 10; m &= 0x7fffffu; }
    else {
        if (e < -25) return sign;
        shift = (uint32_t)(-e - 1); base = 0;
    }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t h = base + q;
    if (mode != FP16_RTZ) {
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) h += 1;
    }
    return sign | h;
}

__device__ __forceinline__ float f16_to_f32(uint32_t h)
{
    uint32_t sign = (h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = sign;
        else {
            int s = __clz((int)m) - 21;          // shifts needed to bring bit 10 up
            m = (m << s) & 0x3ffu;
            u = sign | ((uint32_t)(127 - 15 - s + 1) << 23) | (m << 13);
        }
    } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
    else u = sign | ((e + 112u) << 23) | (m << 13);
    return __uint_as_float(u);
}

__device__ __forceinline__ float quantise_fp16(float v, int mode)
{
    return mode == FP16_OFF ? v : f16_to_f32(f32_to_f16(v, mode));
}

// kernels/CL/common.cl:220-232
__device__ __forceinline__ float safe_inv1(float d)
{
    const float ooeps = 1e-8f;
    return 1.0f / (fabsf(d) > ooeps ? d : copysignf(ooeps, d));
}

// kernels/CL/common.cl:177-218, USE_SAFE_MATH branch (invd = 1.f / denom).
// Returns t, or t_max on a miss.
__device__ __forceinline__ float intersect_triangle(f3 o, f3 d, f3 v1, f3 v2, f3 v3, float t_max)
{
    f3 e1 = sub(v2, v1);
    f3 e2 = sub(v3, v1);
    f3 s1 = cross(d, e2);
    float denom = dot(s1, e1);
    if (denom == 0.0f) return t_max;
    float invd = 1.0f / denom;
    f3 dd = sub(o, v1);
    float b1 = dot(dd, s1) * invd;
    f3 s2 = cross(dd, e1);
    float b2 = dot(d, s2) * invd;
    float temp = dot(e2, s2) * invd;
    if (b1 < 0.0f || b1 > 1.0f || b2 < 0.0f || b1 + b2 > 1.0f || temp < 0.0f || temp > t_max)
        return t_max;
    return temp;
}

// kernels/CL/intersect_bvh2_lds.cl:54-63 (mad == fmaf)
__device__ __forceinline__ void intersect_bbox(f3 pmin, f3 pmax, f3 invdir, f3 oxinvdir, float t_max,
                                               float& t0, float& t1)
{
    float fx = fmaf(pmax.x, invdir.x, oxinvdir.x), nx = fmaf(pmin.x, invdir.x, oxinvdir.x);
    float fy = fmaf(pmax.y, invdir.y, oxinvdir.y), ny = fmaf(pmin.y, invdir.y, oxinvdir.y);
    float fz = fmaf(pmax.z, invdir.z, oxinvdir.z), nz = fmaf(pmin.z, invdir.z, oxinvdir.z);
    float tmaxx = fmaxf(fx, nx), tminx = fminf(fx, nx);
    float tmaxy = fmaxf(fy, ny), tminy = fminf(fy, ny);
    float tmaxz = fmaxf(fz, nz), tminz = fminf(fz, nz);
    t1 = fminf(fminf(fminf(tmaxx, tmaxy), tmaxz), t_max);
    t0 = fmaxf(fmaxf(fmaxf(tminx, tminy), tminz), 0.0f);
}

__device__ __forceinline__ uint32_t pcg_hash(uint32_t v)
{
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}

// sin/cos of 2*pi*u, u in [0,1): quadrant reduction + fixed polynomials.
// This IS the sampler's definition, not an approximation to be swapped for ocml sincos.
__device__ __forceinline__ void sincos2pi(float u, float& s_out, float& c_out)
{
    float t = u * 4.0f;
    int q = (int)t;
    float f = t - (float)q;
    float x = f * 1.57079637f;
    float x2 = x * x;
    float ps = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, -2.50521084e-8f, 2.75573192e-6f),
                 -1.98412698e-4f), 8.33333333e-3f), -1.66666667e-1f), 1.0f);
    float s = x * ps;
    float c = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 2.08767570e-9f, -2.75573192e-7f),
                 2.48015873e-5f), -1.38888889e-3f), 4.16666667e-2f), -0.5f), 1.0f);
    q &= 3;
    s_out = q == 0 ? s : (q == 1 ? c : (q == 2 ? -s : -c));
    c_out = q == 0 ? c : (q == 1 ? -s : (q == 2 ? -c : s));
}

struct CosineGrid { int32_t sx, sy; float inv_sx, inv_sy; };

// COSINE sampler: stratified (sx x sy grid over the unit square), jittered by a
// counter-based hash of (seed, texel, sample), mapped to a cosine-weighted
// direction in the local frame (x, z tangent; y normal).
// (cx, cy) = (i % sx, i / sx): the caller that walks i upwards keeps them as counters instead of dividing
__device__ __forceinline__ f3 cosine_sample_cell(uint32_t base, int32_t i, int32_t cx, int32_t cy, const CosineGrid& g)
{
    const float one_minus = 0.99999994f;
    uint32_t a = pcg_hash(base + 2u * (uint32_t)i);
    uint32_t b = pcg_hash(base + 2u * (uint32_t)i + 1u);
    float j1 = (float)(a >> 8) * 5.9604644775390625e-8f;
    float j2 = (float)(b >> 8) * 5.9604644775390625e-8f;
    float u1 = fminf(((float)cx + j1) * g.inv_sx, one_minus);
    float u2 = fminf(((float)cy + j2) * g.inv_sy, one_minus);
    if (cy >= g.sy) { u1 = j1; u2 = j2; }      // samples beyond the sx*sy grid are plain random
    float r = sqrtf(u1);
    float y = sqrtf(1.0f - u1);
    float s, c;
    sincos2pi(u2, s, c);
    return make_f3(r * c, y, r * s);
}

__device__ __forceinline__ f3 cosine_sample(uint32_t base, int32_t i, const CosineGrid& g)
{
    const int32_t cy = i / g.sx;
    return cosine_sample_cell(base, i, i - cy * g.sx, cy, g);
}

// host side of the COSINE sampler: sx = floor(sqrt(spp)), sy = spp / sx; samples beyond sx*sy are unstratified
inline void cosine_grid(int32_t spp, CosineGrid& g)
{
    int32_t a = 1;
    while ((int64_t)(a + 1) * (a + 1) <= spp) a++;
    g.sx = a; g.sy = spp / a;
    g.inv_sx = 1.0f / (float)g.sx; g.inv_sy = 1.0f / (float)g.sy;
}

inline uint32_t host_pcg(uint32_t v)
{
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}

} // namespace aob

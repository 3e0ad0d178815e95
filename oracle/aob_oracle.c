/*
 * aob_oracle.c — CPU restatement of the reference AO bake path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY (see aob_oracle.h).  PARITY UNPINNED against the
 * reference binary; pinned by closed-form tests and the MSVC rand() sequence.
 *
 * Arithmetic contract (shared with the HIP kernels, stated in DESIGN.md §4):
 * IEEE-754 binary32, round-to-nearest-even, no contraction except where fmaf
 * is written, correctly rounded / and sqrtf, denormals kept.  Build with
 * -ffp-contract=off and without -ffast-math (oracle/Makefile does).
 */
#include "aob_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* small vector helpers; the evaluation order is part of the contract        */

typedef struct { float x, y, z; } f3;

static inline f3 f3_make(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline f3 f3_sub(f3 a, f3 b) { return f3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline float f3_dot(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline f3 f3_cross(f3 a, f3 b)
{
    return f3_make(fmaf(a.y, b.z, -(a.z * b.y)),
                   fmaf(a.z, b.x, -(a.x * b.z)),
                   fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline f3 f3_normalize(f3 v)
{
    float l = sqrtf(f3_dot(v, v));
    return f3_make(v.x / l, v.y / l, v.z / l);
}
static inline int f3_finite(f3 v) { return isfinite(v.x) && isfinite(v.y) && isfinite(v.z); }

/* ------------------------------------------------------------------------ */
/* scalar primitives                                                         */

uint32_t orc_msvc_rand(uint32_t* state)
{
    *state = *state * 214013u + 2531011u;
    return (*state >> 16) & 0x7fffu;
}

int32_t orc_ftoi(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}

uint16_t orc_f32_to_f16(float v, int mode)
{
    uint32_t u; memcpy(&u, &v, 4);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                 /* NaN */
    if (a == 0x7f800000u) return (uint16_t)(sign | 0x7c00u);                /* inf */
    int32_t e = (int32_t)(a >> 23) - 127;
    uint32_t m = (a & 0x7fffffu) | 0x800000u;                               /* 24-bit significand */
    if (a < 0x00800000u) return (uint16_t)sign;                             /* f32 denormal -> 0 */
    if (e > 15) {
        /* overflow: RTZ saturates to the largest finite half, RNE goes to inf */
        return (uint16_t)(sign | (mode == AOB_FP16_RTZ ? 0x7bffu : 0x7c00u));
    }
    uint32_t shift, base;
    if (e >= -14) { shift = 13; base = (uint32_t)(e + 15) << 10; m &= 0x7fffffu; }
    else {
        /* half denormal: value = m * 2^(e-23); unit = 2^-24 -> shift = -e - 1 */
        if (e < -25) return (uint16_t)sign;
        shift = (uint32_t)(-e - 1); base = 0;
    }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t h = base + q;          /* carries ripple into the exponent correctly */
    if (mode != AOB_FP16_RTZ) {
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) h += 1;
    }
    return (uint16_t)(sign | h);
}

float orc_f16_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = sign;
        else {
            /* normalise the denormal */
            int s = 0;
            while (!(m & 0x400u)) { m <<= 1; s++; }
            m &= 0x3ffu;
            u = sign | ((uint32_t)(127 - 15 - s + 1) << 23) | (m << 13);
        }
    } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
    else u = sign | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4);
    return f;
}

/* kernels/CL/common.cl:177-218 (USE_SAFE_MATH branch: invd = 1.f / denom) */
static inline float intersect_triangle(f3 o, f3 d, f3 v1, f3 v2, f3 v3, float t_max)
{
    f3 e1 = f3_sub(v2, v1);
    f3 e2 = f3_sub(v3, v1);
    f3 s1 = f3_cross(d, e2);
    float denom = f3_dot(s1, e1);
    if (denom == 0.0f) return t_max;
    float invd = 1.0f / denom;
    f3 dd = f3_sub(o, v1);
    float b1 = f3_dot(dd, s1) * invd;
    f3 s2 = f3_cross(dd, e1);
    float b2 = f3_dot(d, s2) * invd;
    float temp = f3_dot(e2, s2) * invd;
    if (b1 < 0.0f || b1 > 1.0f || b2 < 0.0f || b1 + b2 > 1.0f || temp < 0.0f || temp > t_max)
        return t_max;
    return temp;
}

float orc_intersect_triangle(const float o[3], const float d[3],
                             const float v1[3], const float v2[3], const float v3[3], float t_max)
{
    return intersect_triangle(f3_make(o[0], o[1], o[2]), f3_make(d[0], d[1], d[2]),
                              f3_make(v1[0], v1[1], v1[2]), f3_make(v2[0], v2[1], v2[2]),
                              f3_make(v3[0], v3[1], v3[2]), t_max);
}

/* kernels/CL/common.cl:220-232 */
static inline float safe_inv1(float d)
{
    const float ooeps = 1e-8f;
    return 1.0f / (fabsf(d) > ooeps ? d : copysignf(ooeps, d));
}
static inline f3 safe_invdir(f3 d) { return f3_make(safe_inv1(d.x), safe_inv1(d.y), safe_inv1(d.z)); }

void orc_safe_invdir(const float d[3], float invdir[3])
{
    f3 r = safe_invdir(f3_make(d[0], d[1], d[2]));
    invdir[0] = r.x; invdir[1] = r.y; invdir[2] = r.z;
}

/* kernels/CL/intersect_bvh2_lds.cl:54-63 (mad == fmaf here) */
static inline void intersect_bbox(const float pmin[3], const float pmax[3], f3 invdir, f3 oxinvdir,
                                  float t_max, float* t0, float* t1)
{
    float fx = fmaf(pmax[0], invdir.x, oxinvdir.x), nx = fmaf(pmin[0], invdir.x, oxinvdir.x);
    float fy = fmaf(pmax[1], invdir.y, oxinvdir.y), ny = fmaf(pmin[1], invdir.y, oxinvdir.y);
    float fz = fmaf(pmax[2], invdir.z, oxinvdir.z), nz = fmaf(pmin[2], invdir.z, oxinvdir.z);
    float tmaxx = fmaxf(fx, nx), tminx = fminf(fx, nx);
    float tmaxy = fmaxf(fy, ny), tminy = fminf(fy, ny);
    float tmaxz = fmaxf(fz, nz), tminz = fminf(fz, nz);
    *t1 = fminf(fminf(fminf(tmaxx, tmaxy), tmaxz), t_max);
    *t0 = fmaxf(fmaxf(fmaxf(tminx, tminy), tminz), 0.0f);
}

void orc_intersect_bbox(const float pmin[3], const float pmax[3], const float invdir[3],
                        const float oxinvdir[3], float t_max, float out[2])
{
    intersect_bbox(pmin, pmax, f3_make(invdir[0], invdir[1], invdir[2]),
                   f3_make(oxinvdir[0], oxinvdir[1], oxinvdir[2]), t_max, &out[0], &out[1]);
}

/* ------------------------------------------------------------------------ */
/* samplers                                                                  */

/* BakeEngine.cpp:312-333 — note y = 1 - sqrt(x^2+z^2), not cos(theta). */
void orc_hemisphere_table(float* out, uint64_t count)
{
    uint32_t state = 1u;
    const float two_pi = 2.0f * 3.14159265358979323846f;   /* RadeonRays/math/mathutils.h:37 */
    for (uint64_t i = 0; i < count; i++) {
        float u = (float)orc_msvc_rand(&state) / 32767.0f;
        float v = (float)orc_msvc_rand(&state) / 32767.0f;
        float sintheta = sqrtf(-u * (u - 2.0f));
        float phi = two_pi * v;
        float x = sintheta * cosf(phi);
        float z = sintheta * sinf(phi);
        float len = sqrtf(x * x + z * z);
        out[3 * i + 0] = x;
        out[3 * i + 1] = 1.0f - len;
        out[3 * i + 2] = z;
    }
}

static uint64_t chunk_texel_max(const AobOptions* opt)
{
    /* BakeEngine.cpp:359-364 */
    uint64_t spp = opt->sample_count > 0 ? (uint64_t)opt->sample_count : 1;
    uint64_t c = (uint64_t)(opt->ray_chunk_size > 0 ? opt->ray_chunk_size : 0) / spp;
    if (c > 65535) c = 65535;
    if (c < 1) c = 1;
    return c;
}

uint64_t orc_compat_table_entries(const AobOptions* opt)
{
    return chunk_texel_max(opt) * (uint64_t)(opt->sample_count > 0 ? opt->sample_count : 0);
}

static inline uint32_t pcg_hash(uint32_t v)
{
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}

void orc_sincos2pi(float u, float* s_out, float* c_out)
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
    switch (q & 3) {
    case 0: *s_out = s;  *c_out = c;  break;
    case 1: *s_out = c;  *c_out = -s; break;
    case 2: *s_out = -s; *c_out = -c; break;
    default: *s_out = -c; *c_out = s; break;
    }
}

static void cosine_grid(int32_t spp, int32_t* sx, int32_t* sy)
{
    /* sx = floor(sqrt(spp)), sy = spp / sx: the first sx*sy samples are one per cell of an
     * sx x sy grid, the remaining spp - sx*sy are plain random (an incomplete last grid row
     * would bias the estimate) */
    int32_t a = 1;
    while ((int64_t)(a + 1) * (a + 1) <= spp) a++;
    *sx = a;
    *sy = spp / a;
}

typedef struct { int32_t sx, sy; float inv_sx, inv_sy; } CosineGrid;

static inline f3 cosine_sample(uint32_t base, int32_t i, const CosineGrid* g)
{
    const float one_minus = 0.99999994f;
    uint32_t a = pcg_hash(base + 2u * (uint32_t)i);
    uint32_t b = pcg_hash(base + 2u * (uint32_t)i + 1u);
    float j1 = (float)(a >> 8) * 5.9604644775390625e-8f;
    float j2 = (float)(b >> 8) * 5.9604644775390625e-8f;
    int32_t cx = i % g->sx, cy = i / g->sx;
    float u1, u2;
    if (cy < g->sy) {
        u1 = fminf(((float)cx + j1) * g->inv_sx, one_minus);
        u2 = fminf(((float)cy + j2) * g->inv_sy, one_minus);
    } else { u1 = j1; u2 = j2; }
    float r = sqrtf(u1);
    float y = sqrtf(1.0f - u1);
    float s, c;
    orc_sincos2pi(u2, &s, &c);
    return f3_make(r * c, y, r * s);
}

void orc_cosine_sample(uint32_t seed, uint32_t texel, int32_t i, int32_t sample_count, float out[3])
{
    CosineGrid g;
    cosine_grid(sample_count, &g.sx, &g.sy);
    g.inv_sx = 1.0f / (float)g.sx; g.inv_sy = 1.0f / (float)g.sy;
    uint32_t base = pcg_hash(texel + pcg_hash(seed));
    f3 d = cosine_sample(base, i, &g);
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}

/* ------------------------------------------------------------------------ */
/* scene -> world-space geometry                                             */

static inline const float* attr_ptr(const void* base, uint32_t stride, uint32_t i)
{
    return (const float*)((const uint8_t*)base + (size_t)stride * i);
}

/* RasterizeVertexShader.hlsl:32: worldpos = mul(float4(pos,1), model) (row vector) */
static inline f3 transform_point(const float* m, const float* p)
{
    f3 r;
    float* o = &r.x;
    for (int j = 0; j < 3; j++) {
        float t = p[0] * m[0 * 4 + j];
        t = fmaf(p[1], m[1 * 4 + j], t);
        t = fmaf(p[2], m[2 * 4 + j], t);
        o[j] = t + m[3 * 4 + j];
    }
    return r;
}

/* RasterizeVertexShader.hlsl:30: normalize(mul(float4(norm,0), model)) — the model
 * matrix itself, not its inverse transpose. */
static inline f3 transform_normal(const float* m, const float* n)
{
    f3 r;
    float* o = &r.x;
    for (int j = 0; j < 3; j++) {
        float t = n[0] * m[0 * 4 + j];
        t = fmaf(n[1], m[1 * 4 + j], t);
        t = fmaf(n[2], m[2 * 4 + j], t);
        o[j] = t;
    }
    return f3_normalize(r);
}

uint64_t orc_scene_triangle_count(const AobSceneDesc* scene)
{
    uint64_t n = 0;
    for (uint32_t i = 0; i < scene->mesh_count; i++) n += scene->meshes[i].index_count / 3;
    return n;
}

/* BakeEngine.cpp:370-398: every section becomes a shape; RadeonRays applies the
 * shape transform to the vertices, so the BVH holds world-space triangles.
 * Shape ids start at 1 (RadeonRays/radeon_rays.h:238). */
void orc_scene_triangles(const AobSceneDesc* scene, OrcTriangle* out)
{
    uint64_t k = 0;
    for (uint32_t mi = 0; mi < scene->mesh_count; mi++) {
        const AobMeshDesc* m = &scene->meshes[mi];
        uint32_t ntri = m->index_count / 3;
        for (uint32_t t = 0; t < ntri; t++, k++) {
            for (int c = 0; c < 3; c++) {
                uint32_t vi = m->indices[3 * t + c];
                f3 w = transform_point(m->model, attr_ptr(m->positions, m->position_stride, vi));
                out[k].v[c][0] = w.x; out[k].v[c][1] = w.y; out[k].v[c][2] = w.z;
            }
            out[k].shape_id = mi + 1;
            out[k].prim_id = t;
        }
    }
}

/* ------------------------------------------------------------------------ */
/* LBVH build (kernels/CL/build_hlbvh.cl), 63-bit Morton codes                */

static inline uint64_t expand21(uint32_t v)
{
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

static inline uint32_t morton_axis(float c, float smin, float extent)
{
    /* build_hlbvh.cl:122 (center - scene_min) / scene_extents, then :71-73 clamp */
    float n = extent > 0.0f ? (c - smin) / extent : 0.0f;
    float q = fminf(fmaxf(n * 2097152.0f, 0.0f), 2097151.0f);
    return (uint32_t)q;
}

typedef struct { uint64_t code; uint32_t idx; } MortonKey;

static int morton_cmp(const void* a, const void* b)
{
    const MortonKey* x = (const MortonKey*)a; const MortonKey* y = (const MortonKey*)b;
    if (x->code != y->code) return x->code < y->code ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* build_hlbvh.cl:131-150 */
static inline int lbvh_delta(const MortonKey* keys, int64_t n, int64_t i1, int64_t i2)
{
    int64_t left = i1 < i2 ? i1 : i2, right = i1 < i2 ? i2 : i1;
    if (left < 0 || right >= n) return -1;
    uint64_t lc = keys[left].code, rc = keys[right].code;
    if (lc != rc) return __builtin_clzll(lc ^ rc);
    const uint32_t x = (uint32_t)left ^ (uint32_t)right;
    return 64 + (x ? __builtin_clz(x) : 32);          /* OpenCL's clz(0) is 32 (i1 == i2: the search probes its own position) */
}

typedef struct { float pmin[3], pmax[3]; } Box;

static inline void box_union(Box* d, const Box* a, const Box* b)
{
    for (int k = 0; k < 3; k++) { d->pmin[k] = fminf(a->pmin[k], b->pmin[k]); d->pmax[k] = fmaxf(a->pmax[k], b->pmax[k]); }
}

/* merged-box surface area as the PLOC builder of the HIP library evaluates it */
static inline float merged_area(const Box* a, const Box* b)
{
    const float dx = fmaxf(a->pmax[0], b->pmax[0]) - fminf(a->pmin[0], b->pmin[0]);
    const float dy = fmaxf(a->pmax[1], b->pmax[1]) - fminf(a->pmin[1], b->pmin[1]);
    const float dz = fmaxf(a->pmax[2], b->pmax[2]) - fminf(a->pmin[2], b->pmin[2]);
    return fmaf(dz, dx, fmaf(dy, dz, dx * dy));
}

/* bvh_builder 2 (bvh_build.hip build_sah_top): the top of the tree, over the m clusters PLOC has made so far (`cl`, in list order;
 * a cluster is a node id with its box in `bounds`), top-down by binned SAH — the reference stack's own lever for traversal cost
 * (RadeonRays/radeon_rays.h:286-294: bvh.builder = "sah").  Every list range [b, e) of two or more clusters:
 *   - the box of the clusters' centres c = (pmin + pmax) * 0.5; on every axis with a positive extent 16 bins,
 *     bin = min(15, (int)((c - cmin) / extent * 16)), with the clusters' count and box per bin;
 *   - the 3 x 15 planes between bins, in axis then bin order: cost = area(left) * count(left) + area(right) * count(right),
 *     area = dx * dy + dy * dz + dz * dx; the first plane of the lowest cost that leaves clusters on both sides wins;
 *   - the range is partitioned at that plane KEEPING THE LIST ORDER on both sides (no such plane: cut in the middle of the list);
 *   - the node's id is a function of where its range was cut: with the left part [b, mid) the position k = mid - 1 is unique
 *     among all nodes, and lies in 0 .. m-2 — the ids PLOC's own numbering (the k-th node made is n-2-k) has not used.  The root
 *     must be node 0: the root takes id 0 and the one node cut at k = 0 takes the root's k instead.  No order of construction
 *     enters: ranges can be split level by level, or each by a thread of its own. */
#define ORC_SAH_TOP_CLUSTERS 131072
#define ORC_SAH_BINS 16
typedef struct { int64_t begin, end; uint32_t parent; int side; } TopRange;
/* the cut of one range: returns mid (b < mid < e) and leaves the range's clusters partitioned in tmp[b, e); *node_box = their box */
static int64_t sah_cut(const uint32_t* cl, uint32_t* tmp, int64_t b, int64_t e, const Box* bounds, Box* node_box)
{
    const int64_t cnt = e - b;
    float cmin[3] = { INFINITY, INFINITY, INFINITY }, cmax[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int k = 0; k < 3; k++) { node_box->pmin[k] = INFINITY; node_box->pmax[k] = -INFINITY; }
    for (int64_t i = b; i < e; i++) {
        box_union(node_box, node_box, &bounds[cl[i]]);
        for (int k = 0; k < 3; k++) {
            const float c = (bounds[cl[i]].pmin[k] + bounds[cl[i]].pmax[k]) * 0.5f;
            cmin[k] = fminf(cmin[k], c); cmax[k] = fmaxf(cmax[k], c);
        }
    }
    int best_axis = -1, best_split = 0; float best_cost = INFINITY;
    for (int ax = 0; ax < 3; ax++) {
        const float ext = cmax[ax] - cmin[ax];
        if (!(ext > 0.0f)) continue;
        Box bb[ORC_SAH_BINS]; uint32_t bc[ORC_SAH_BINS];
        for (int q = 0; q < ORC_SAH_BINS; q++) { bc[q] = 0; for (int k = 0; k < 3; k++) { bb[q].pmin[k] = INFINITY; bb[q].pmax[k] = -INFINITY; } }
        for (int64_t i = b; i < e; i++) {
            const float c = (bounds[cl[i]].pmin[ax] + bounds[cl[i]].pmax[ax]) * 0.5f;
            int q = (int)((c - cmin[ax]) / ext * (float)ORC_SAH_BINS); if (q > ORC_SAH_BINS - 1) q = ORC_SAH_BINS - 1;
            bc[q]++; box_union(&bb[q], &bb[q], &bounds[cl[i]]);
        }
        float right_area[ORC_SAH_BINS]; uint32_t right_cnt[ORC_SAH_BINS];
        Box acc; for (int k = 0; k < 3; k++) { acc.pmin[k] = INFINITY; acc.pmax[k] = -INFINITY; }
        uint32_t c = 0;
        for (int q = ORC_SAH_BINS - 1; q > 0; q--) {
            c += bc[q]; if (bc[q]) box_union(&acc, &acc, &bb[q]);
            const float dx = acc.pmax[0] - acc.pmin[0], dy = acc.pmax[1] - acc.pmin[1], dz = acc.pmax[2] - acc.pmin[2];
            right_area[q] = c ? dx * dy + dy * dz + dz * dx : 0.0f; right_cnt[q] = c;
        }
        for (int k = 0; k < 3; k++) { acc.pmin[k] = INFINITY; acc.pmax[k] = -INFINITY; }
        c = 0;
        for (int q = 0; q < ORC_SAH_BINS - 1; q++) {
            c += bc[q]; if (bc[q]) box_union(&acc, &acc, &bb[q]);
            if (c == 0 || right_cnt[q + 1] == 0) continue;
            const float dx = acc.pmax[0] - acc.pmin[0], dy = acc.pmax[1] - acc.pmin[1], dz = acc.pmax[2] - acc.pmin[2];
            const float cost = (dx * dy + dy * dz + dz * dx) * (float)c + right_area[q + 1] * (float)right_cnt[q + 1];
            if (cost < best_cost) { best_cost = cost; best_axis = ax; best_split = q + 1; }
        }
    }
    if (best_axis < 0) { for (int64_t i = b; i < e; i++) tmp[i] = cl[i]; return b + cnt / 2; }
    const float ext = cmax[best_axis] - cmin[best_axis];
    int64_t at = b, mid = b;
    for (int pass = 0; pass < 2; pass++) {          /* stable: the left clusters in list order, then the right ones */
        for (int64_t i = b; i < e; i++) {
            const float c = (bounds[cl[i]].pmin[best_axis] + bounds[cl[i]].pmax[best_axis]) * 0.5f;
            int q = (int)((c - cmin[best_axis]) / ext * (float)ORC_SAH_BINS); if (q > ORC_SAH_BINS - 1) q = ORC_SAH_BINS - 1;
            if ((q < best_split) == (pass == 0)) tmp[at++] = cl[i];
        }
        if (pass == 0) mid = at;
    }
    return mid;
}
static void sah_top(uint32_t* cl, uint32_t* tmp, int64_t m, Box* bounds, uint32_t* left, uint32_t* right, uint32_t* parent)
{
    TopRange* todo = (TopRange*)malloc(sizeof(TopRange) * (size_t)m);
    int64_t top = 0;
    int64_t root_k = -1;
    todo[top].begin = 0; todo[top].end = m; todo[top].parent = 0xffffffffu; todo[top].side = 0; top++;
    while (top) {
        const TopRange t = todo[--top];
        Box bx;
        const int64_t mid = sah_cut(cl, tmp, t.begin, t.end, bounds, &bx);
        for (int64_t i = t.begin; i < t.end; i++) cl[i] = tmp[i];
        const int64_t k = mid - 1;
        if (t.parent == 0xffffffffu) root_k = k;
        const uint32_t id = t.parent == 0xffffffffu ? 0u : (k == 0 ? (uint32_t)root_k : (uint32_t)k);
        bounds[id] = bx;
        if (t.parent != 0xffffffffu) { if (t.side == 0) left[t.parent] = id; else right[t.parent] = id; parent[id] = t.parent; }
        const int64_t rb[2] = { t.begin, mid }, re[2] = { mid, t.end };
        for (int s2 = 1; s2 >= 0; s2--) {
            if (re[s2] - rb[s2] == 1) { const uint32_t c = cl[rb[s2]]; if (s2 == 0) left[id] = c; else right[id] = c; parent[c] = id; }
            else { todo[top].begin = rb[s2]; todo[top].end = re[s2]; todo[top].parent = id; todo[top].side = s2; top++; }
        }
    }
    free(todo);
}

uint64_t orc_bvh_build(const OrcTriangle* tris, uint64_t n, AobBvhNode* nodes, uint32_t* depth_out)
{
    return orc_bvh_build_ex(tris, n, nodes, depth_out, 0, 0);
}

uint64_t orc_bvh_build_ex(const OrcTriangle* tris, uint64_t n, AobBvhNode* nodes, uint32_t* depth_out, int builder, int adaptive)
{
    if (builder < 0) builder = 1;      /* "by workload" without a workload: PLOC (orc_resolve_bvh_builder settles it for a bake) */
    if (depth_out) *depth_out = 0;
    if (n == 0) return 0;
    Box* pb = (Box*)malloc(sizeof(Box) * n);
    Box scene;
    for (int k = 0; k < 3; k++) { scene.pmin[k] = INFINITY; scene.pmax[k] = -INFINITY; }
    for (uint64_t i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) {
            pb[i].pmin[k] = fminf(fminf(tris[i].v[0][k], tris[i].v[1][k]), tris[i].v[2][k]);
            pb[i].pmax[k] = fmaxf(fmaxf(tris[i].v[0][k], tris[i].v[1][k]), tris[i].v[2][k]);
            scene.pmin[k] = fminf(scene.pmin[k], pb[i].pmin[k]);
            scene.pmax[k] = fmaxf(scene.pmax[k], pb[i].pmax[k]);
        }
    }
    MortonKey* keys = (MortonKey*)malloc(sizeof(MortonKey) * n);
    float ext[3];
    for (int k = 0; k < 3; k++) ext[k] = scene.pmax[k] - scene.pmin[k];
    for (uint64_t i = 0; i < n; i++) {
        uint32_t q[3];
        for (int k = 0; k < 3; k++) {
            float c = (pb[i].pmax[k] + pb[i].pmin[k]) * 0.5f;        /* build_hlbvh.cl:118 */
            q[k] = morton_axis(c, scene.pmin[k], ext[k]);
        }
        if (adaptive) {
            /* AobOptions::morton_adaptive: the bits go to the axes in the order a spatial-median builder would split
             * them (radeon_rays.h:283-287) — always the axis whose cell is currently longest, ties to the lower axis */
            float e[3] = { ext[0], ext[1], ext[2] };
            int used[3] = { 0, 0, 0 };
            uint64_t code = 0;
            for (int bit = 0; bit < 63; bit++) {
                int ax = -1; float best = -1.0f;
                for (int k = 0; k < 3; k++) if (used[k] < 21 && e[k] > best) { best = e[k]; ax = k; }
                code = code << 1 | ((q[ax] >> (20 - used[ax])) & 1u);
                used[ax]++; e[ax] *= 0.5f;
            }
            keys[i].code = code;
        } else
        keys[i].code = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
        keys[i].idx = (uint32_t)i;
    }
    qsort(keys, n, sizeof(MortonKey), morton_cmp);

    uint64_t node_count = 2 * n - 1;
    memset(nodes, 0, sizeof(AobBvhNode) * node_count);
    if (n == 1) {
        const OrcTriangle* t = &tris[0];
        memcpy(nodes[0].lmin_or_v0, t->v[0], 12); memcpy(nodes[0].lmax_or_v1, t->v[1], 12);
        memcpy(nodes[0].rmin_or_v2, t->v[2], 12);
        nodes[0].addr_left = AOB_INVALID_ADDR; nodes[0].shape_id = t->shape_id; nodes[0].prim_id = t->prim_id;
        free(pb); free(keys);
        return 1;
    }

    int64_t N = (int64_t)n;
    uint32_t* left = (uint32_t*)malloc(4 * (n - 1));
    uint32_t* right = (uint32_t*)malloc(4 * (n - 1));
    uint32_t* parent = (uint32_t*)malloc(4 * node_count);
    Box* bounds = (Box*)malloc(sizeof(Box) * node_count);
    parent[0] = 0xffffffffu;
    uint32_t max_depth = 0;
    if (builder == 1 || builder == 2) {
        /* PLOC: the sequential statement of the library's parallel builder.  Clusters in Morton order look
         * `radius` places left and right for the neighbour with the smallest merged surface area (ties: the
         * lower position); mutual nearest neighbours merge; the list is compacted; repeat.  Node ids are final
         * from the start: leaf at position p is n-1+p, the k-th inner node created is n-2-k (root = 0). */
        const int64_t radius = 2;       /* kPlocRadius of the library */
        for (int64_t p = 0; p < N; p++) bounds[N - 1 + p] = pb[keys[p].idx];
        uint32_t* cl = (uint32_t*)malloc(4 * n); uint32_t* nxt = (uint32_t*)malloc(4 * n); uint32_t* nn = (uint32_t*)malloc(4 * n);
        for (int64_t p = 0; p < N; p++) cl[p] = (uint32_t)(N - 1 + p);
        int64_t m = N; uint32_t created = 0; int iter = 0;
        /* builder 2: clustering stops once the list is short enough for the top of the tree to be built over the clusters
         * top-down, by binned SAH (below) — where a SAH tree differs most from an agglomerated one */
        const int64_t stop = builder == 2 ? ORC_SAH_TOP_CLUSTERS : 1;
        while (m > stop) {
            const int forced = iter >= 256;
            for (int64_t i = 0; i < m; i++) {
                if (forced) { nn[i] = (uint32_t)(((i ^ 1) < m) ? (i ^ 1) : i); continue; }
                const int64_t lo = i > radius ? i - radius : 0, hi = i + radius < m - 1 ? i + radius : m - 1;
                float best = INFINITY; int64_t bj = i;
                for (int64_t j = lo; j <= hi; j++) {
                    if (j == i) continue;
                    const float a = merged_area(&bounds[cl[i]], &bounds[cl[j]]);
                    if (a < best || bj == i) { best = a; bj = j; }
                }
                nn[i] = (uint32_t)bj;
            }
            int64_t pos = 0; uint32_t mk = 0;
            for (int64_t i = 0; i < m; i++) {
                const int64_t j = nn[i];
                const int mutual = j != i && (int64_t)nn[j] == i;
                if (mutual && i > j) continue;
                uint32_t id = cl[i];
                if (mutual) {
                    const uint32_t a = cl[i], b2 = cl[j];
                    id = (uint32_t)(N - 2) - (created + mk); mk++;
                    box_union(&bounds[id], &bounds[a], &bounds[b2]);
                    left[id] = a; right[id] = b2; parent[a] = id; parent[b2] = id;
                }
                nxt[pos++] = id;
            }
            uint32_t* t = cl; cl = nxt; nxt = t;
            m = pos; created += mk; iter++;
        }
        if (m > 1) sah_top(cl, nxt, m, bounds, left, right, parent);
        parent[0] = 0xffffffffu;
        for (int64_t p = 0; p < N; p++) {
            uint32_t dpt = 0;
            for (uint32_t a = (uint32_t)(N - 1 + p); parent[a] != 0xffffffffu; a = parent[a]) dpt++;
            if (dpt > max_depth) max_depth = dpt;
        }
        free(cl); free(nxt); free(nn);
    } else {
    /* build_hlbvh.cl:152-261: span, split, child links */
    for (int64_t idx = 0; idx < N - 1; idx++) {
        int dnext = lbvh_delta(keys, N, idx, idx + 1), dprev = lbvh_delta(keys, N, idx, idx - 1);
        int d = dnext >= dprev ? 1 : -1;
        int delta_min = lbvh_delta(keys, N, idx, idx - d);
        int64_t lmax = 2;
        while (lbvh_delta(keys, N, idx, idx + lmax * d) > delta_min) lmax *= 2;
        int64_t l = 0, t = lmax;
        do {
            t /= 2;
            if (lbvh_delta(keys, N, idx, idx + (l + t) * d) > delta_min) l += t;
        } while (t > 1);
        int64_t sx = idx < idx + l * d ? idx : idx + l * d;
        int64_t sy = idx < idx + l * d ? idx + l * d : idx;
        int64_t lo = sx, hi = sy;
        int num_identical = lbvh_delta(keys, N, lo, hi);
        do {
            int64_t ns = (hi + lo) / 2;
            if (lbvh_delta(keys, N, lo, ns) > num_identical) lo = ns; else hi = ns;
        } while (hi > lo + 1);
        int64_t split = lo;
        uint32_t c1 = (split == sx) ? (uint32_t)(N - 1 + split) : (uint32_t)split;
        uint32_t c2 = (split + 1 == sy) ? (uint32_t)(N - 1 + split + 1) : (uint32_t)(split + 1);
        left[idx] = c1; right[idx] = c2;
        parent[c1] = (uint32_t)idx; parent[c2] = (uint32_t)idx;
    }
    /* leaves */
    for (int64_t p = 0; p < N; p++) bounds[N - 1 + p] = pb[keys[p].idx];
    /* refit (build_hlbvh.cl:264-314), serial: children before parents via visit counts */
    uint8_t* flags = (uint8_t*)calloc(n - 1, 1);
    for (int64_t p = 0; p < N; p++) {
        uint32_t idx = (uint32_t)(N - 1 + p);
        uint32_t dpt = 0;
        for (uint32_t a = idx; parent[a] != 0xffffffffu; a = parent[a]) dpt++;
        if (dpt > max_depth) max_depth = dpt;
        do {
            idx = parent[idx];
            if (flags[idx]++ == 0) break;
            box_union(&bounds[idx], &bounds[left[idx]], &bounds[right[idx]]);
        } while (idx != 0);
    }
    free(flags);
    }
    if (depth_out) *depth_out = max_depth;
    /* emit the 64-byte traversal nodes (intersect_bvh2_lds.cl:40-52) */
    for (int64_t i = 0; i < N - 1; i++) {
        AobBvhNode* nd = &nodes[i];
        const Box* bl = &bounds[left[i]]; const Box* br = &bounds[right[i]];
        memcpy(nd->lmin_or_v0, bl->pmin, 12); memcpy(nd->lmax_or_v1, bl->pmax, 12);
        memcpy(nd->rmin_or_v2, br->pmin, 12); memcpy(nd->rmax, br->pmax, 12);
        nd->addr_left = left[i]; nd->addr_right = right[i];
        nd->shape_id = 0; nd->prim_id = 0;
    }
    for (int64_t p = 0; p < N; p++) {
        AobBvhNode* nd = &nodes[N - 1 + p];
        const OrcTriangle* t = &tris[keys[p].idx];
        memcpy(nd->lmin_or_v0, t->v[0], 12); memcpy(nd->lmax_or_v1, t->v[1], 12);
        memcpy(nd->rmin_or_v2, t->v[2], 12);
        nd->addr_left = AOB_INVALID_ADDR; nd->addr_right = 0;
        nd->shape_id = t->shape_id; nd->prim_id = t->prim_id;
    }
    free(bounds); free(parent); free(right); free(left); free(keys); free(pb);
    return node_count;
}

/* ------------------------------------------------------------------------ */
/* UV rasterisation                                                          */

#define SUBPIXEL 256
#define COORD_LIMIT 67108864.0f   /* 2^26 sub-pixels: |screen coordinate| < 2^18 px */

static inline int64_t edge_fn(int64_t ax, int64_t ay, int64_t bx, int64_t by, int64_t px, int64_t py)
{
    return (bx - ax) * (py - ay) - (by - ay) * (px - ax);
}
/* D3D11 top-left rule for an edge a->b of a triangle whose interior is edge_fn > 0 */
static inline int edge_top_left(int64_t ax, int64_t ay, int64_t bx, int64_t by)
{
    int64_t dx = bx - ax, dy = by - ay;
    return dy < 0 || (dy == 0 && dx > 0);
}

/* the largest value of the edge function a -> b over a texel square, above its value at the centre */
static inline int64_t edge_reach(int64_t ax, int64_t ay, int64_t bx, int64_t by)
{
    const int64_t dx = bx - ax, dy = by - ay;
    return 128 * ((dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy));
}

typedef struct { f3 wpos, wnrm; float u, v; } RasterVertex;

static inline float quantise_fp16(float v, int mode)
{
    return mode == AOB_FP16_OFF ? v : orc_f16_to_f32(orc_f32_to_f16(v, mode));
}

uint64_t orc_rasterize(const AobSceneDesc* scene, int32_t material, const AobOptions* opt,
                       AobSurfel* surfels, uint32_t* section_index, uint64_t capacity)
{
    return orc_rasterize_ex(scene, material, opt, surfels, section_index, NULL, capacity);
}

uint64_t orc_rasterize_ex(const AobSceneDesc* scene, int32_t material, const AobOptions* opt,
                          AobSurfel* surfels, uint32_t* section_index, uint32_t* surfel_mesh, uint64_t capacity)
{
    if (material < 0 || (uint32_t)material >= scene->material_count) return 0;
    const int32_t W = scene->materials[material].width, H = scene->materials[material].height;
    if (W <= 0 || H <= 0) return 0;
    uint32_t* cover = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)W * H);
    /* AOB_LAYERS_ALL: every layer of the stencil peel (BakeEngine.cpp:531-532) is kept, each as a section of its own;
     * prev holds, per texel, the triangle index + 1 of its fragment in the layer before */
    const int peel = opt->raster_layers == AOB_LAYERS_ALL && opt->raster_mode != AOB_RASTER_CONSERVATIVE;
    uint32_t* prev = peel ? (uint32_t*)calloc((size_t)W * H, sizeof(uint32_t)) : NULL;
    uint64_t total = 0;
    for (uint32_t mi = 0; mi < scene->mesh_count; mi++) {
        const AobMeshDesc* m = &scene->meshes[mi];
        if (m->material != material) continue;                       /* BakeEngine.cpp:523 */
        if (peel) memset(prev, 0, sizeof(uint32_t) * (size_t)W * H);
        for (int layer = 0; layer < (peel ? 255 : 1); layer++) {
        memset(cover, 0, sizeof(uint32_t) * (size_t)W * H);
        uint32_t ntri = m->index_count / 3;
        uint64_t layer_count = 0;
        /* pass 1: the stencil layer peel keeps, per texel, the LAST fragment drawn
         * (BakeEngine.cpp:130-140,531-588): highest triangle index covering the centre */
        for (uint32_t t = 0; t < ntri; t++) {
            int64_t X[3], Y[3];
            int ok = 1;
            for (int c = 0; c < 3; c++) {
                const float* uv = attr_ptr(m->uvs, m->uv_stride, m->indices[3 * t + c]);
                /* RasterizeVertexShader.hlsl:29 + viewport: pixel = (u*W, v*H) */
                float sx = uv[0] * (float)W * (float)SUBPIXEL, sy = uv[1] * (float)H * (float)SUBPIXEL;
                if (!(fabsf(sx) < COORD_LIMIT) || !(fabsf(sy) < COORD_LIMIT)) { ok = 0; break; }
                X[c] = (int64_t)rintf(sx); Y[c] = (int64_t)rintf(sy);
            }
            if (!ok) continue;
            int64_t area = edge_fn(X[0], Y[0], X[1], Y[1], X[2], Y[2]);
            if (area == 0) continue;
            if (area < 0) { int64_t tx = X[1], ty = Y[1]; X[1] = X[2]; Y[1] = Y[2]; X[2] = tx; Y[2] = ty; }
            int64_t minx = X[0] < X[1] ? X[0] : X[1]; if (X[2] < minx) minx = X[2];
            int64_t maxx = X[0] > X[1] ? X[0] : X[1]; if (X[2] > maxx) maxx = X[2];
            int64_t miny = Y[0] < Y[1] ? Y[0] : Y[1]; if (Y[2] < miny) miny = Y[2];
            int64_t maxy = Y[0] > Y[1] ? Y[0] : Y[1]; if (Y[2] > maxy) maxy = Y[2];
            /* texels whose centre lies in the bounding box — or, conservative, whose open square overlaps it */
            const int cons = opt->raster_mode == AOB_RASTER_CONSERVATIVE;
            int64_t px0 = cons ? minx >> 8 : (minx - 128 + 255) >> 8, px1 = cons ? (maxx - 1) >> 8 : (maxx - 128) >> 8;
            int64_t py0 = cons ? miny >> 8 : (miny - 128 + 255) >> 8, py1 = cons ? (maxy - 1) >> 8 : (maxy - 128) >> 8;
            if (px0 < 0) px0 = 0; if (py0 < 0) py0 = 0;
            if (px1 > W - 1) px1 = W - 1; if (py1 > H - 1) py1 = H - 1;
            int64_t b0 = edge_top_left(X[0], Y[0], X[1], Y[1]) ? 0 : -1;
            int64_t b1 = edge_top_left(X[1], Y[1], X[2], Y[2]) ? 0 : -1;
            int64_t b2 = edge_top_left(X[2], Y[2], X[0], Y[0]) ? 0 : -1;
            for (int64_t py = py0; py <= py1; py++) {
                for (int64_t px = px0; px <= px1; px++) {
                    int64_t cx = px * 256 + 128, cy = py * 256 + 128;
                    const int64_t e0 = edge_fn(X[0], Y[0], X[1], Y[1], cx, cy), e1 = edge_fn(X[1], Y[1], X[2], Y[2], cx, cy),
                                  e2 = edge_fn(X[2], Y[2], X[0], Y[0], cx, cy);
                    const int centre = e0 + b0 >= 0 && e1 + b1 >= 0 && e2 + b2 >= 0;
                    uint32_t key = t + 1;
                    if (cons) {
                        /* AOB_RASTER_CONSERVATIVE: the texel's square and the triangle share area iff no edge separates them
                         * (the sides of the square: the loop bounds); a triangle covering the centre beats any that does not */
                        const int touch = e0 + edge_reach(X[0], Y[0], X[1], Y[1]) > 0 && e1 + edge_reach(X[1], Y[1], X[2], Y[2]) > 0 &&
                                          e2 + edge_reach(X[2], Y[2], X[0], Y[0]) > 0;
                        if (!centre && !touch) continue;
                        if (centre) key |= 0x80000000u;
                    } else if (!centre) continue;
                    uint32_t* c = &cover[(size_t)py * W + px];
                    if (peel) {
                        /* the first triangle after the previous layer's */
                        if (key > prev[(size_t)py * W + px] && (*c == 0 || key < *c)) { if (*c == 0) layer_count++; *c = key; }
                        continue;
                    }
                    if (key > *c) *c = key;
                }
            }
        }
        if (peel && layer > 0 && layer_count == 0) break;             /* the peel is through */
        /* pass 2: attributes of the winning fragment, row-major extraction */
        uint32_t local = 0;
        for (int32_t py = 0; py < H; py++) {
            for (int32_t px = 0; px < W; px++) {
                uint32_t c = cover[(size_t)py * W + px];
                if (!c) continue;
                uint32_t t = (c & 0x7fffffffu) - 1;
                if (peel) { t = c - 1; prev[(size_t)py * W + px] = c; }
                const int clamp = opt->raster_mode == AOB_RASTER_CONSERVATIVE && !(c & 0x80000000u);
                RasterVertex rv[3];
                int64_t X[3], Y[3];
                for (int k = 0; k < 3; k++) {
                    uint32_t vi = m->indices[3 * t + k];
                    const float* uv = attr_ptr(m->uvs, m->uv_stride, vi);
                    X[k] = (int64_t)rintf(uv[0] * (float)W * (float)SUBPIXEL);
                    Y[k] = (int64_t)rintf(uv[1] * (float)H * (float)SUBPIXEL);
                    rv[k].wpos = transform_point(m->model, attr_ptr(m->positions, m->position_stride, vi));
                    rv[k].wnrm = transform_normal(m->model, attr_ptr(m->normals, m->normal_stride, vi));
                    rv[k].u = uv[0]; rv[k].v = uv[1];
                }
                int64_t area = edge_fn(X[0], Y[0], X[1], Y[1], X[2], Y[2]);
                if (area < 0) {
                    int64_t tx = X[1], ty = Y[1]; X[1] = X[2]; Y[1] = Y[2]; X[2] = tx; Y[2] = ty;
                    RasterVertex tv = rv[1]; rv[1] = rv[2]; rv[2] = tv;
                    area = -area;
                }
                int64_t cx = (int64_t)px * 256 + 128, cy = (int64_t)py * 256 + 128;
                float fa = (float)area;
                float l0 = (float)edge_fn(X[1], Y[1], X[2], Y[2], cx, cy) / fa;
                float l1 = (float)edge_fn(X[2], Y[2], X[0], Y[0], cx, cy) / fa;
                float l2 = (float)edge_fn(X[0], Y[0], X[1], Y[1], cx, cy) / fa;
                if (clamp) {
                    /* the centre is outside: sample at the point of the triangle nearest to it, which lies on one of the
                     * edges the centre is on the wrong side of (first such edge on ties) */
                    float l[3] = { l0, l1, l2 };
                    double best = INFINITY;
                    for (int e = 0; e < 3; e++) {
                        const int a = e, bb = (e + 1) % 3, cc = (e + 2) % 3;
                        if (edge_fn(X[a], Y[a], X[bb], Y[bb], cx, cy) >= 0) continue;
                        const int64_t dx = X[bb] - X[a], dy = Y[bb] - Y[a], qx = cx - X[a], qy = cy - Y[a];
                        const int64_t num = qx * dx + qy * dy, den = dx * dx + dy * dy;
                        float t; double d2;
                        if (num <= 0) { t = 0.0f; d2 = (double)(qx * qx + qy * qy); }
                        else if (num >= den) { t = 1.0f; const int64_t rx = cx - X[bb], ry = cy - Y[bb]; d2 = (double)(rx * rx + ry * ry); }
                        else { t = (float)num / (float)den; d2 = (double)(qx * qx + qy * qy) - ((double)num * (double)num) / (double)den; }
                        if (d2 < best) { best = d2; l[a] = 1.0f - t; l[bb] = t; l[cc] = 0.0f; }
                    }
                    l0 = l[0]; l1 = l[1]; l2 = l[2];
                }
#define LERP(f) fmaf(l2, rv[2].f, fmaf(l1, rv[1].f, l0 * rv[0].f))
                if (total < capacity) {
                    AobSurfel* s = &surfels[total];
                    s->texel_x = px; s->texel_y = py;
                    s->position[0] = LERP(wpos.x); s->position[1] = LERP(wpos.y); s->position[2] = LERP(wpos.z);
                    /* RasterizePixelShader.hlsl:20-21: U rides with the fp32 position,
                     * normal + V go through the RGBA16F target (BakeEngine.cpp:446,584-585) */
                    s->texcoord[0] = LERP(u);
                    s->texcoord[1] = quantise_fp16(LERP(v), opt->normal_fp16);
                    s->normal[0] = quantise_fp16(LERP(wnrm.x), opt->normal_fp16);
                    s->normal[1] = quantise_fp16(LERP(wnrm.y), opt->normal_fp16);
                    s->normal[2] = quantise_fp16(LERP(wnrm.z), opt->normal_fp16);
                    if (section_index) section_index[total] = local;
                    if (surfel_mesh) surfel_mesh[total] = mi;
                }
#undef LERP
                total++; local++;
            }
        }
        }   /* layers */
    }
    free(cover); free(prev);
    return total;
}

/* ------------------------------------------------------------------------ */
/* 4-ary collapse of the BVH2 (the structure the default GPU kernel walks)    */

/* Every inner node at even depth becomes a 4-ary node whose children are its grandchildren
 * (or a child itself where that child is a leaf); the boxes are the BVH2 boxes, unchanged.
 * Derived from the 64-byte node array alone.  nodes4 needs max(1, n-1) entries (indexed by the
 * BVH2 inner index; entries of collapsed-away levels stay zero), tris needs n entries. */
uint64_t orc_bvh4_build(const AobBvhNode* nodes2, uint64_t node_count2, AobBvh4Node* nodes4, AobLeafTri* tris)
{
    return orc_bvh4_build_ex(nodes2, node_count2, nodes4, tris, 0);
}

/* a BVH2 node whose children are the leaves at Morton positions p and p + 1 (AOB_PAIR_FLAG) */
static inline int is_leaf_pair(const AobBvhNode* nodes2, uint32_t c, uint32_t leaf_base)
{
    return c < leaf_base && nodes2[c].addr_left >= leaf_base && nodes2[c].addr_right == nodes2[c].addr_left + 1u;
}

uint64_t orc_bvh4_build_ex(const AobBvhNode* nodes2, uint64_t node_count2, AobBvh4Node* nodes4, AobLeafTri* tris, int pairs)
{
    if (node_count2 == 0) return 0;
    const uint64_t n = (node_count2 + 1) / 2;
    const uint32_t leaf_base = (uint32_t)(n - 1);
    uint32_t inf_bits = 0x7f800000u, ninf_bits = 0xff800000u;
    float pinf, ninf; memcpy(&pinf, &inf_bits, 4); memcpy(&ninf, &ninf_bits, 4);
    const uint32_t invalid = AOB_INVALID_ADDR;
    for (uint64_t p = 0; p < n; p++) {
        const AobBvhNode* lf = &nodes2[leaf_base + p];
        memcpy(tris[p].v0, lf->lmin_or_v0, 12); memcpy(&tris[p].v0[3], &lf->shape_id, 4);
        memcpy(tris[p].v1, lf->lmax_or_v1, 12); memcpy(&tris[p].v1[3], &lf->prim_id, 4);
        memcpy(tris[p].v2, lf->rmin_or_v2, 12); tris[p].v2[3] = 0.0f;
    }
    const uint64_t count4 = n == 1 ? 1 : n - 1;
    memset(nodes4, 0, sizeof(AobBvh4Node) * count4);
#define EMPTY_SLOT(nd, j) do { (nd)->lo[j][0] = (nd)->lo[j][1] = (nd)->lo[j][2] = pinf; memcpy(&(nd)->lo[j][3], &invalid, 4); \
                               (nd)->hi[j][0] = (nd)->hi[j][1] = (nd)->hi[j][2] = ninf; (nd)->hi[j][3] = 0.0f; } while (0)
    if (n == 1) {
        AobBvh4Node* nd = &nodes4[0];
        const AobBvhNode* lf = &nodes2[0];
        for (int k = 0; k < 3; k++) {
            nd->lo[0][k] = fminf(fminf(lf->lmin_or_v0[k], lf->lmax_or_v1[k]), lf->rmin_or_v2[k]);
            nd->hi[0][k] = fmaxf(fmaxf(lf->lmin_or_v0[k], lf->lmax_or_v1[k]), lf->rmin_or_v2[k]);
        }
        uint32_t ref = AOB_LEAF_FLAG; memcpy(&nd->lo[0][3], &ref, 4); nd->hi[0][3] = 0.0f;
        for (int j = 1; j < 4; j++) EMPTY_SLOT(nd, j);
        return 1;
    }
    /* walk the even levels from the root */
    uint32_t* todo = (uint32_t*)malloc(sizeof(uint32_t) * n);
    uint64_t top = 0;
    todo[top++] = 0;
    while (top) {
        const uint32_t i = todo[--top];
        const AobBvhNode* me = &nodes2[i];
        AobBvh4Node* nd = &nodes4[i];
        int k = 0;
        const uint32_t c2[2] = { me->addr_left, me->addr_right };
        const float* cmin[2] = { me->lmin_or_v0, me->rmin_or_v2 };
        const float* cmax[2] = { me->lmax_or_v1, me->rmax };
        for (int s2 = 0; s2 < 2; s2++) {
            const uint32_t c = c2[s2];
            if (c < leaf_base && !(pairs && is_leaf_pair(nodes2, c, leaf_base))) {
                const AobBvhNode* ch = &nodes2[c];
                const uint32_t g[2] = { ch->addr_left, ch->addr_right };
                const float* gmin[2] = { ch->lmin_or_v0, ch->rmin_or_v2 };
                const float* gmax[2] = { ch->lmax_or_v1, ch->rmax };
                for (int t = 0; t < 2; t++, k++) {
                    uint32_t ref = g[t] < leaf_base ? g[t] : (AOB_LEAF_FLAG | (g[t] - leaf_base));
                    const int pair = pairs && is_leaf_pair(nodes2, g[t], leaf_base);
                    if (pair) ref = AOB_LEAF_FLAG | AOB_PAIR_FLAG | (nodes2[g[t]].addr_left - leaf_base);
                    memcpy(nd->lo[k], gmin[t], 12); memcpy(&nd->lo[k][3], &ref, 4);
                    memcpy(nd->hi[k], gmax[t], 12); nd->hi[k][3] = 0.0f;
                    if (g[t] < leaf_base && !pair) todo[top++] = g[t];
                }
            } else {
                uint32_t ref = c < leaf_base ? (AOB_LEAF_FLAG | AOB_PAIR_FLAG | (nodes2[c].addr_left - leaf_base)) : (AOB_LEAF_FLAG | (c - leaf_base));
                memcpy(nd->lo[k], cmin[s2], 12); memcpy(&nd->lo[k][3], &ref, 4);
                memcpy(nd->hi[k], cmax[s2], 12); nd->hi[k][3] = 0.0f;
                k++;
            }
        }
        for (; k < 4; k++) EMPTY_SLOT(nd, k);
    }
#undef EMPTY_SLOT
    free(todo);
    return count4;
}

#define ORC_STACK 128

/* Section visibility (AOB_MESH_*): the ray mask of intersect_bvh2_lds.cl:163-166 by section.  A ray cast from a
 * surfel of shape `self` ignores the triangles of shape s when s == self and self is hidden from itself, or
 * s != self and s is hidden from others.  Shape ids are mesh index + 1.  NULL: nothing is hidden. */
typedef struct { const uint32_t* mesh_flags; uint32_t mesh_count; uint32_t self_shape; } OrcMask;
static inline int mask_skips(const OrcMask* mask, uint32_t shape)
{
    if (!mask || shape == 0 || shape > mask->mesh_count) return 0;
    const uint32_t flags = mask->mesh_flags[shape - 1];
    return shape == mask->self_shape ? (flags & AOB_MESH_HIDDEN_FROM_SELF) != 0 : (flags & AOB_MESH_HIDDEN_FROM_OTHERS) != 0;
}

/* ---- quantised BVH2 nodes (AobBvh2QNode) ------------------------------------------------- */
/* A 15-bit grid (the 4-ary nodes keep the 16th bit of each plane set, see orc_bvh4q_build).
 * frame: origin = scene min, scale = extent/32767 nudged up until origin + 32767*scale >= max;
 * child box min rounded down, max rounded up on that grid (so the decoded box contains the fp32
 * box).  The scene box is the union of the root's two child boxes. */
#define ORC_QUANT_MAX 32767.0f
static inline uint32_t quant_down(float v, float origin, float scale)
{
    if (!(scale > 0.0f)) return 0u;
    float q = fminf(fmaxf(floorf((v - origin) / scale), 0.0f), ORC_QUANT_MAX);
    while (q > 0.0f && fmaf(q, scale, origin) > v) q -= 1.0f;
    return (uint32_t)q;
}
static inline uint32_t quant_up(float v, float origin, float scale)
{
    if (!(scale > 0.0f)) return 0u;
    float q = fminf(fmaxf(ceilf((v - origin) / scale), 0.0f), ORC_QUANT_MAX);
    while (q < ORC_QUANT_MAX && fmaf(q, scale, origin) < v) q += 1.0f;
    return (uint32_t)q;
}

uint64_t orc_bvh2q_build(const AobBvhNode* nodes2, uint64_t node_count2, AobBvh2QNode* out, AobQuantFrame* frame)
{
    memset(frame, 0, sizeof *frame);
    if (node_count2 == 0) return 0;
    const uint64_t n = (node_count2 + 1) / 2;
    float smin[3], smax[3];
    const AobBvhNode* root = &nodes2[0];
    for (int k = 0; k < 3; k++) {
        if (n == 1) {
            smin[k] = fminf(fminf(root->lmin_or_v0[k], root->lmax_or_v1[k]), root->rmin_or_v2[k]);
            smax[k] = fmaxf(fmaxf(root->lmin_or_v0[k], root->lmax_or_v1[k]), root->rmin_or_v2[k]);
        } else {
            smin[k] = fminf(root->lmin_or_v0[k], root->rmin_or_v2[k]);
            smax[k] = fmaxf(root->lmax_or_v1[k], root->rmax[k]);
        }
        float extent = smax[k] - smin[k];
        float scale = extent > 0.0f ? extent / ORC_QUANT_MAX : 0.0f;
        if (scale > 0.0f) while (fmaf(ORC_QUANT_MAX, scale, smin[k]) < smax[k]) { uint32_t u; memcpy(&u, &scale, 4); u++; memcpy(&scale, &u, 4); }
        frame->origin[k] = smin[k]; frame->scale[k] = scale;
    }
    for (uint64_t i = 0; i + 1 < n; i++) {
        const AobBvhNode* nd = &nodes2[i];
        uint32_t q[12];
        for (int k = 0; k < 3; k++) {
            const float o = frame->origin[k], sc = frame->scale[k];
            q[k] = quant_down(nd->lmin_or_v0[k], o, sc); q[3 + k] = quant_up(nd->lmax_or_v1[k], o, sc);
            q[6 + k] = quant_down(nd->rmin_or_v2[k], o, sc); q[9 + k] = quant_up(nd->rmax[k], o, sc);
        }
        uint32_t* w = out[i].w;
        /* one word per axis and box: min | max << 16 */
        w[0] = q[0] | q[3] << 16; w[1] = q[1] | q[4] << 16; w[2] = q[2] | q[5] << 16; w[3] = q[6] | q[9] << 16;
        w[4] = q[7] | q[10] << 16; w[5] = q[8] | q[11] << 16; w[6] = nd->addr_left; w[7] = nd->addr_right;
    }
    return n - 1;
}

/* intersect_bvh2_lds.cl's walk on the quantised nodes: the de-quantisation is folded into the ray,
 * (origin + q*scale - o)*invdir = q*(scale*invdir) + (origin*invdir + oxinvdir), so the slab test of
 * :54-63 runs on the raw grid values. */
static inline int traverse2q(const AobBvh2QNode* nq, uint64_t n, const AobQuantFrame* fr, const AobLeafTri* tris,
                             f3 o, f3 d, float t_max, int any_hit, float* t_out, OrcTraceStats* st, int* error, const OrcMask* mask)
{
    if (n == 0) { *t_out = t_max; return 0; }
    f3 invdir = safe_invdir(d);
    f3 oxinvdir = f3_make(-o.x * invdir.x, -o.y * invdir.y, -o.z * invdir.z);
    f3 fb = f3_make(fmaf(fr->origin[0], invdir.x, oxinvdir.x), fmaf(fr->origin[1], invdir.y, oxinvdir.y), fmaf(fr->origin[2], invdir.z, oxinvdir.z));
    f3 fa = f3_make(fr->scale[0] * invdir.x, fr->scale[1] * invdir.y, fr->scale[2] * invdir.z);
    const uint32_t leaf_base = (uint32_t)(n - 1);
    float closest_t = t_max;
    int hit = 0;
    uint32_t stack[ORC_STACK];
    int sp = 0;
    stack[sp++] = AOB_INVALID_ADDR;
    uint32_t addr = 0;
    while (addr != AOB_INVALID_ADDR) {
        if (addr < leaf_base) {
            const uint32_t* w = nq[addr].w;
            st->inner_visits++;
            const float lmin[3] = { (float)(w[0] & 0xffffu), (float)(w[1] & 0xffffu), (float)(w[2] & 0xffffu) };
            const float lmax[3] = { (float)(w[0] >> 16), (float)(w[1] >> 16), (float)(w[2] >> 16) };
            const float rmin[3] = { (float)(w[3] & 0xffffu), (float)(w[4] & 0xffffu), (float)(w[5] & 0xffffu) };
            const float rmax[3] = { (float)(w[3] >> 16), (float)(w[4] >> 16), (float)(w[5] >> 16) };
            float s0x, s0y, s1x, s1y;
            intersect_bbox(lmin, lmax, fa, fb, closest_t, &s0x, &s0y);
            intersect_bbox(rmin, rmax, fa, fb, closest_t, &s1x, &s1y);
            int c0 = s0x <= s0y, c1 = s1x <= s1y;
            int c1first = c1 && (s0x > s1x);
            if (c0 || c1) {
                uint32_t deferred;
                if (c1first || !c0) { addr = w[7]; deferred = w[6]; }
                else { addr = w[6]; deferred = w[7]; }
                if (c0 && c1) {
                    if (sp >= ORC_STACK) { *error = AOB_ERR_BVH_TOO_DEEP; *t_out = t_max; return 0; }
                    stack[sp++] = deferred;
                }
                continue;
            }
        } else {
            const AobLeafTri* tr = &tris[addr - leaf_base];
            st->leaf_visits++;
            float t = intersect_triangle(o, d, f3_make(tr->v0[0], tr->v0[1], tr->v0[2]), f3_make(tr->v1[0], tr->v1[1], tr->v1[2]),
                                         f3_make(tr->v2[0], tr->v2[1], tr->v2[2]), closest_t);
            if (mask) { uint32_t shape; memcpy(&shape, &tr->v0[3], 4); if (mask_skips(mask, shape)) t = closest_t; }
            if (t < closest_t) {
                closest_t = t; hit = 1;
                if (any_hit) break;
            }
        }
        addr = stack[--sp];
    }
    *t_out = closest_t;
    return hit;
}

/* Quantised 4-ary nodes (AobBvh4QNode): the boxes of orc_bvh4_build on the grid of orc_bvh2q_build.  A plane q is
 * stored as 0x8000 | q: with 0x3f above and 0x00 below, those two bytes are the float 1 + q/32768 (plane_float). */
uint64_t orc_bvh4q_build(const AobBvh4Node* nodes4, uint64_t count4, const AobQuantFrame* fr, AobBvh4QNode* out)
{
    /* packed: the live records (slot 0 holds a ref; collapsed-away levels are all zero) in index order, refs renumbered */
    uint32_t* packed = (uint32_t*)malloc(sizeof(uint32_t) * (count4 ? count4 : 1));
    uint64_t live = 0;
    for (uint64_t i = 0; i < count4; i++) {
        uint32_t ref0; memcpy(&ref0, &nodes4[i].lo[0][3], 4);
        packed[i] = (uint32_t)live;
        live += ref0 != 0u;
    }
    for (uint64_t i = 0; i < count4; i++) {
        uint32_t ref0; memcpy(&ref0, &nodes4[i].lo[0][3], 4);
        if (ref0 == 0u) continue;
        AobBvh4QNode* o = &out[packed[i]];
        for (int j = 0; j < 4; j++) {
            const float* lo = nodes4[i].lo[j]; const float* hi = nodes4[i].hi[j];
            uint32_t ref; memcpy(&ref, &lo[3], 4);
            uint32_t* w = &o->w[4 * j];
            /* an empty slot: the inverted box (min plane 0x7fff, max plane 0) no ray enters — and, for the scenes whose planes
             * arithmetic cannot tell apart (no extent at all, or less than the rounding of a ray's distance to it), a ref that
             * is harmless to follow: leaf triangle 0 (one more test of a triangle of the scene changes no answer) */
            w[0] = w[1] = w[2] = 0x8000ffffu; w[3] = AOB_LEAF_FLAG;
            if (ref == AOB_INVALID_ADDR) continue;
            w[3] = (ref & AOB_LEAF_FLAG) ? ref : packed[ref];
            for (int k = 0; k < 3; k++)
                w[k] = 0x80008000u | quant_down(lo[k], fr->origin[k], fr->scale[k]) | quant_up(hi[k], fr->origin[k], fr->scale[k]) << 16;
        }
    }
    free(packed);
    return live;
}

#define ORC_STACK4 512
/* measurement only (tools/measure_closest_cuts.py): closest-hit rays follow their hit children in SLOT order, as any-hit rays do */
static int g_measure_slot_order = 0;
void orc_measure_slot_order(int on) { g_measure_slot_order = on; }

/* bytes {0x3f, hi, lo, 0x00}: what one v_perm_b32 makes of a stored plane */
static inline float plane_float(uint32_t half) { uint32_t u = 0x3f000000u | (half & 0xffffu) << 8; float f; memcpy(&f, &u, 4); return f; }
/* v_max3_f32 / v_min3_f32: NaN operands are ignored */
static inline float max3n(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
static inline float min3n(float a, float b, float c) { return fminf(fminf(a, b), c); }

/* traverse4 on the quantised boxes, with the arithmetic of the HIP kernel (trace.hip, QUANT 2):
 *   - a stored plane becomes f = 1 + q/32768, so the frame folded into the ray is origin - 32768*scale and 32768*scale:
 *     t = f*fa + fb with fa = scale2*invdir, fb = origin2*invdir + oxinvdir;
 *   - the near plane of an axis is min for rays that travel up it and max for rays that travel down it (the sign of 1/dir);
 *   - any hit: closest_t never moves off t_max, so fa and fb are scaled by 1/t_max and the near distance is clamped to
 *     [0, 1] (NaN -> 0, the clamp modifier of v_max3_f32); the far distance is not clamped — a box wholly beyond t_max
 *     passes, which is harmless because the triangle test has the last word;
 *   - closest hit: intersect_bvh2_lds.cl:54-63 as before, near clamped at 0 and far at closest_t. */
/* (start / n_start: measurement only — the walk begins with these refs on its stack instead of at the root, start[0] first;
 * NULL: the root) */
static inline int traverse4q_from(const AobBvh4QNode* nq, uint64_t count4, const AobQuantFrame* fr, const AobLeafTri* tris,
                                  f3 o, f3 d, float t_max, int any_hit, float* t_out, OrcTraceStats* st, int* error, const OrcMask* mask,
                                  const uint32_t* start, int n_start)
{
    if (count4 == 0) { *t_out = t_max; return 0; }
    f3 invdir = safe_invdir(d);
    f3 oxinvdir = f3_make(-o.x * invdir.x, -o.y * invdir.y, -o.z * invdir.z);
    const float inv[3] = { invdir.x, invdir.y, invdir.z };
    const float oxi[3] = { oxinvdir.x, oxinvdir.y, oxinvdir.z };
    const float inv_tmax = 1.0f / t_max;
    float fa[3], fb[3];
    int down[3];
    for (int k = 0; k < 3; k++) {
        const float origin2 = fmaf(-32768.0f, fr->scale[k], fr->origin[k]);
        const float scale2 = fr->scale[k] * 32768.0f;
        fb[k] = fmaf(origin2, inv[k], oxi[k]);
        fa[k] = scale2 * inv[k];
        if (any_hit) { fa[k] = fa[k] * inv_tmax; fb[k] = fb[k] * inv_tmax; }
        uint32_t bits; memcpy(&bits, &inv[k], 4);
        down[k] = (int)(bits >> 31);
    }
    float closest_t = t_max;
    int hit = 0;
    uint32_t stack[ORC_STACK4];
    int sp = 0;
    stack[sp++] = AOB_INVALID_ADDR;
    uint32_t node = 0;
    if (start) {
        if (n_start < 1) { *t_out = t_max; return 0; }
        for (int k = n_start - 1; k >= 1; k--) stack[sp++] = start[k];
        node = start[0];
    }
    while (node != AOB_INVALID_ADDR) {
        if (!(node & AOB_LEAF_FLAG)) {
            const uint32_t* w = nq[node].w;
            st->inner_visits++;
            float key[4]; uint32_t ref[4]; int chit[4], nhit = 0;
            for (int j = 0; j < 4; j++) {
                const uint32_t* c = &w[4 * j];
                float tn[3], tf[3];
                for (int k = 0; k < 3; k++) {
                    const uint32_t nearh = down[k] ? c[k] >> 16 : c[k] & 0xffffu, farh = down[k] ? c[k] & 0xffffu : c[k] >> 16;
                    tn[k] = fmaf(plane_float(nearh), fa[k], fb[k]);
                    tf[k] = fmaf(plane_float(farh), fa[k], fb[k]);
                }
                float t0 = max3n(tn[0], tn[1], tn[2]), t1 = min3n(tf[0], tf[1], tf[2]);
                if (any_hit) t0 = t0 != t0 ? 0.0f : fminf(fmaxf(t0, 0.0f), 1.0f);
                else { t0 = fmaxf(t0, 0.0f); t1 = fminf(t1, closest_t); }
                ref[j] = c[3];
                chit[j] = t0 <= t1;                      /* an empty slot's inverted box fails this by itself (or leads to triangle 0) */
                key[j] = chit[j] ? t0 : INFINITY;
                nhit += chit[j];
            }
            if (nhit == 0) { node = stack[--sp]; continue; }
            if (sp + nhit - 1 > ORC_STACK4) { *error = AOB_ERR_BVH_TOO_DEEP; *t_out = t_max; return 0; }
            uint32_t first = AOB_INVALID_ADDR;
            /* any hit: the answer does not depend on the order — the kernel follows the first hit child and defers the rest
             * in index order; closest hit: nearest first, the others deferred farthest first, ties by child index */
            if (any_hit || g_measure_slot_order) for (int j = 0; j < 4; j++) key[j] = chit[j] ? (float)j : INFINITY;
            for (int j = 0; j < 4; j++) {
                if (!chit[j]) continue;
                int rank = 0;
                for (int k = 0; k < 4; k++) rank += (key[k] < key[j]) || (key[k] == key[j] && k < j);
                if (rank == 0) first = ref[j];
                else stack[sp + (nhit - 1) - rank] = ref[j];
            }
            sp += nhit - 1;
            node = first;
        } else {
            /* a leaf is one triangle or — AOB_PAIR_FLAG — the two at p and p + 1, both tested against the same bound */
            const AobLeafTri* tr = &tris[node & ~(AOB_LEAF_FLAG | AOB_PAIR_FLAG)];
            st->leaf_visits++;
            float t = intersect_triangle(o, d, f3_make(tr->v0[0], tr->v0[1], tr->v0[2]), f3_make(tr->v1[0], tr->v1[1], tr->v1[2]),
                                         f3_make(tr->v2[0], tr->v2[1], tr->v2[2]), closest_t);
            if (mask) { uint32_t shape; memcpy(&shape, &tr->v0[3], 4); if (mask_skips(mask, shape)) t = closest_t; }
            if (node & AOB_PAIR_FLAG) {
                const AobLeafTri* tr2 = tr + 1;
                st->leaf_visits++;
                float t2 = intersect_triangle(o, d, f3_make(tr2->v0[0], tr2->v0[1], tr2->v0[2]), f3_make(tr2->v1[0], tr2->v1[1], tr2->v1[2]),
                                              f3_make(tr2->v2[0], tr2->v2[1], tr2->v2[2]), closest_t);
                if (mask) { uint32_t shape; memcpy(&shape, &tr2->v0[3], 4); if (mask_skips(mask, shape)) t2 = closest_t; }
                t = fminf(t, t2);
            }
            if (t < closest_t) {
                closest_t = t; hit = 1;
                if (any_hit) break;
            }
            node = stack[--sp];
        }
    }
    *t_out = closest_t;
    return hit;
}

static inline int traverse4q(const AobBvh4QNode* nq, uint64_t count4, const AobQuantFrame* fr, const AobLeafTri* tris,
                             f3 o, f3 d, float t_max, int any_hit, float* t_out, OrcTraceStats* st, int* error, const OrcMask* mask)
{
    return traverse4q_from(nq, count4, fr, tris, o, d, t_max, any_hit, t_out, st, error, mask, NULL, 0);
}


/* The 4-ary walk of the GPU's quad-per-ray kernel, one ray at a time: slab-test the four child
 * boxes (intersect_bvh2_lds.cl:54-63), follow the nearest hit child and defer the others
 * farthest first (the 4-ary form of the near-child-first order of :123-153); leaves are single
 * triangles (common.cl:177-218). */
static inline int traverse4(const AobBvh4Node* nodes4, uint64_t count4, const AobLeafTri* tris, f3 o, f3 d, float t_max,
                            int any_hit, float* t_out, OrcTraceStats* st, int* error, const OrcMask* mask)
{
    if (count4 == 0) { *t_out = t_max; return 0; }
    f3 invdir = safe_invdir(d);
    f3 oxinvdir = f3_make(-o.x * invdir.x, -o.y * invdir.y, -o.z * invdir.z);
    float closest_t = t_max;
    int hit = 0;
    uint32_t stack[ORC_STACK4];
    int sp = 0;
    stack[sp++] = AOB_INVALID_ADDR;
    uint32_t node = 0;
    while (node != AOB_INVALID_ADDR) {
        if (!(node & AOB_LEAF_FLAG)) {
            const AobBvh4Node* nd = &nodes4[node];
            st->inner_visits++;
            float key[4]; uint32_t ref[4]; int chit[4], nhit = 0;
            for (int j = 0; j < 4; j++) {
                float t0, t1;
                intersect_bbox(nd->lo[j], nd->hi[j], invdir, oxinvdir, closest_t, &t0, &t1);
                memcpy(&ref[j], &nd->lo[j][3], 4);
                chit[j] = t0 <= t1 && ref[j] != AOB_INVALID_ADDR;
                key[j] = chit[j] ? t0 : INFINITY;
                nhit += chit[j];
            }
            if (nhit == 0) { node = stack[--sp]; continue; }
            if (sp + nhit - 1 > ORC_STACK4) { *error = AOB_ERR_BVH_TOO_DEEP; *t_out = t_max; return 0; }
            uint32_t first = AOB_INVALID_ADDR;
            for (int j = 0; j < 4; j++) {
                if (!chit[j]) continue;
                int rank = 0;
                for (int k = 0; k < 4; k++) rank += (key[k] < key[j]) || (key[k] == key[j] && k < j);
                if (rank == 0) first = ref[j];
                else stack[sp + (nhit - 1) - rank] = ref[j];
            }
            sp += nhit - 1;
            node = first;
        } else {
            const AobLeafTri* tr = &tris[node & ~AOB_LEAF_FLAG];
            st->leaf_visits++;
            float t = intersect_triangle(o, d, f3_make(tr->v0[0], tr->v0[1], tr->v0[2]), f3_make(tr->v1[0], tr->v1[1], tr->v1[2]),
                                         f3_make(tr->v2[0], tr->v2[1], tr->v2[2]), closest_t);
            if (mask) { uint32_t shape; memcpy(&shape, &tr->v0[3], 4); if (mask_skips(mask, shape)) t = closest_t; }
            if (t < closest_t) {
                closest_t = t; hit = 1;
                if (any_hit) break;
            }
            node = stack[--sp];
        }
    }
    *t_out = closest_t;
    return hit;
}

/* ------------------------------------------------------------------------ */
/* trace                                                                     */


typedef struct {
    const AobBvhNode* nodes; uint64_t node_count;
    const AobBvh4Node* nodes4; uint64_t node4_count; const AobLeafTri* tris;   /* bvh_width != 2 */
    const AobBvh2QNode* nodes2q; AobQuantFrame frame; uint64_t tri_count;        /* bvh_width 2, bvh_quantised */
    const AobBvh4QNode* nodes4q;                                                 /* bvh_width 4, bvh_quantised */
    const AobSurfel* surfels; const uint32_t* section_index;
    const uint32_t* surfel_mesh; const uint32_t* mesh_flags; uint32_t mesh_count;   /* section visibility, or NULL */
    float ao_max_range;                                                             /* material range, or 0 */
    const AobOptions* opt; const float* hemisphere;
    int32_t width, height; AobResult* results;
    uint64_t begin, end;
    uint64_t next;            /* dynamic work distribution */
    pthread_mutex_t lock;
    OrcTraceStats stats;
    int error;
} TraceJob;

/* intersect_bvh2_lds.cl:65-226 (closest) / :228-363 (any hit): near child first,
 * deferred child on a stack, leaf = inlined triangle. Returns hit flag, *t_out. */
static inline int traverse_addr(const AobBvhNode* nodes, uint64_t node_count, f3 o, f3 d, float t_max,
                                int any_hit, float* t_out, OrcTraceStats* st, int* error, const OrcMask* mask, uint32_t* addr_out)
{
    if (addr_out) *addr_out = AOB_INVALID_ADDR;
    if (node_count == 0) { *t_out = t_max; return 0; }
    f3 invdir = safe_invdir(d);
    f3 oxinvdir = f3_make(-o.x * invdir.x, -o.y * invdir.y, -o.z * invdir.z);
    float closest_t = t_max;
    int hit = 0;
    uint32_t stack[ORC_STACK];
    int sp = 0;
    stack[sp++] = AOB_INVALID_ADDR;
    uint32_t addr = 0;
    while (addr != AOB_INVALID_ADDR) {
        const AobBvhNode* n = &nodes[addr];
        if (n->addr_left != AOB_INVALID_ADDR) {
            st->inner_visits++;
            float s0x, s0y, s1x, s1y;
            intersect_bbox(n->lmin_or_v0, n->lmax_or_v1, invdir, oxinvdir, closest_t, &s0x, &s0y);
            intersect_bbox(n->rmin_or_v2, n->rmax, invdir, oxinvdir, closest_t, &s1x, &s1y);
            int c0 = s0x <= s0y, c1 = s1x <= s1y;
            int c1first = c1 && (s0x > s1x);
            if (c0 || c1) {
                uint32_t deferred;
                if (c1first || !c0) { addr = n->addr_right; deferred = n->addr_left; }
                else { addr = n->addr_left; deferred = n->addr_right; }
                if (c0 && c1) {
                    if (sp >= ORC_STACK) { *error = AOB_ERR_BVH_TOO_DEEP; *t_out = t_max; return 0; }
                    stack[sp++] = deferred;
                }
                continue;
            }
        } else {
            st->leaf_visits++;
            float t = intersect_triangle(o, d,
                                         f3_make(n->lmin_or_v0[0], n->lmin_or_v0[1], n->lmin_or_v0[2]),
                                         f3_make(n->lmax_or_v1[0], n->lmax_or_v1[1], n->lmax_or_v1[2]),
                                         f3_make(n->rmin_or_v2[0], n->rmin_or_v2[1], n->rmin_or_v2[2]),
                                         closest_t);
            if (mask && mask_skips(mask, n->shape_id)) t = closest_t;
            if (t < closest_t) {
                closest_t = t; hit = 1;
                if (addr_out) *addr_out = addr;          /* closest_addr, intersect_bvh2_lds.cl:176 */
                if (any_hit) break;
            }
        }
        addr = stack[--sp];
    }
    *t_out = closest_t;
    return hit;
}

static inline int traverse(const AobBvhNode* nodes, uint64_t node_count, f3 o, f3 d, float t_max,
                           int any_hit, float* t_out, OrcTraceStats* st, int* error, const OrcMask* mask)
{
    return traverse_addr(nodes, node_count, o, d, t_max, any_hit, t_out, st, error, mask, NULL);
}

/* FLOAT accumulation is order independent by construction: every sample's r*w and w are
 * converted (exactly: a float times 2^32, truncated) to Q32.32 fixed point, the 64-bit integer
 * sums are exact, and each sum is rounded to float once.  The reference has no float mode
 * (it sums (int)(x*255), ProcessIntersectionsComputeShader.hlsl:87-90); this is the build's
 * higher-precision variant of the same accumulation. */
static inline int64_t f2fix(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) v = 2147483520.0f;
    if (v <= -2147483648.0f) v = -2147483520.0f;
    return (int64_t)(v * 4294967296.0f);
}

static void trace_surfel(const TraceJob* job, uint64_t k, OrcTraceStats* st, int* error,
                         const CosineGrid* grid, uint64_t chunk_max)
{
    const AobOptions* opt = job->opt;
    const AobSurfel* s = &job->surfels[k];
    const int32_t spp = opt->sample_count;
    f3 P = f3_make(s->position[0], s->position[1], s->position[2]);
    f3 N = f3_make(s->normal[0], s->normal[1], s->normal[2]);
    if (s->texel_x < 0 || s->texel_y < 0 || s->texel_x >= job->width || s->texel_y >= job->height) return;
    AobResult* cell = &job->results[(size_t)s->texel_y * job->width + s->texel_x];
    /* GenerateRaysComputeShader.hlsl:34-41 */
    f3 X = f3_make(1.0f, 0.0f, 0.0f), Z = f3_make(0.0f, 0.0f, 1.0f);
    if (N.y < 0.9999f && N.y > -0.9999f) {
        X = f3_normalize(f3_make(-N.z, 0.0f, N.x));      /* cross(N, (0,1,0)) */
        Z = f3_normalize(f3_cross(X, N));
    }
    /* a non-finite frame makes every weight NaN, and (int)NaN == 0 in HLSL: the
     * surfel contributes nothing in the reference, so it contributes nothing here */
    if (!f3_finite(P) || !f3_finite(N) || !f3_finite(X) || !f3_finite(Z)) return;

    const int compat = opt->sampler == AOB_SAMPLER_REFERENCE_COMPAT;
    const int any_hit = opt->use_any_hit != 0;
    float t_max = opt->ray_tmax;
    float F = opt->falloff_distance;
    if (opt->material_ranges && job->ao_max_range > 0.0f) {       /* MaterialObject::AOMaxRange */
        if (job->ao_max_range < t_max) t_max = job->ao_max_range;
        F = job->ao_max_range;
    }
    if (!any_hit && opt->clamp_tmax_to_falloff && F < t_max) t_max = F;
    OrcMask mask_store = { job->mesh_flags, job->mesh_count, job->surfel_mesh ? job->surfel_mesh[k] + 1u : 0u };
    const OrcMask* mask = job->mesh_flags && job->surfel_mesh ? &mask_store : NULL;
    uint32_t texel = (uint32_t)s->texel_y * (uint32_t)job->width + (uint32_t)s->texel_x;
    uint32_t base = pcg_hash(texel + pcg_hash(opt->seed));
    /* GenerateRaysComputeShader.hlsl:45 Hemisphere[DTid.x * SampleCount + i], DTid.x chunk-local */
    uint64_t local = job->section_index ? job->section_index[k] : k;
    const float* table = compat ? job->hemisphere + 3 * ((local % chunk_max) * (uint64_t)spp) : NULL;

    int32_t qT = 0, qW = 0;
    int64_t xT = 0, xW = 0;     /* FLOAT mode: exact Q32.32 sums, rounded to float once at the end */
    for (int32_t i = 0; i < spp; i++) {
        f3 h = compat ? f3_make(table[3 * i], table[3 * i + 1], table[3 * i + 2]) : cosine_sample(base, i, grid);
        /* :46 */
        f3 dir = f3_make(fmaf(h.z, Z.x, fmaf(h.y, N.x, h.x * X.x)),
                         fmaf(h.z, Z.y, fmaf(h.y, N.y, h.x * X.y)),
                         fmaf(h.z, Z.z, fmaf(h.y, N.z, h.x * X.z)));
        /* :47 origin offset along the ray */
        f3 org = f3_make(fmaf(dir.x, opt->ray_bias, P.x), fmaf(dir.y, opt->ray_bias, P.y), fmaf(dir.z, opt->ray_bias, P.z));
        float t;
        st->rays++;
        int hit = job->nodes4q ? traverse4q(job->nodes4q, job->node4_count, &job->frame, job->tris, org, dir, t_max, any_hit, &t, st, error, mask)
                : job->nodes2q ? traverse2q(job->nodes2q, job->tri_count, &job->frame, job->tris, org, dir, t_max, any_hit, &t, st, error, mask)
                : job->nodes4 ? traverse4(job->nodes4, job->node4_count, job->tris, org, dir, t_max, any_hit, &t, st, error, mask)
                              : traverse(job->nodes, job->node_count, org, dir, t_max, any_hit, &t, st, error, mask);
        st->hits += (uint64_t)hit;
        /* ProcessIntersectionsComputeShader.hlsl:52-85 */
        float w = compat ? f3_dot(N, dir) : 1.0f;
        float r;
        if (any_hit) r = hit ? 0.0f : 1.0f;
        else r = (hit && t < F) ? t / F : 1.0f;
        if (opt->accumulate == AOB_ACCUMULATE_QUANTISED) {
            qT += orc_ftoi(r * w * 255.0f);          /* :87 */
            qW += orc_ftoi(w * 255.0f);              /* :88 */
        } else {
            xT += f2fix(r * w); xW += f2fix(w);
        }
    }
    /* :89-90 InterlockedAdd into Results[y*W+x] (one surfel per cell per section) */
    if (opt->accumulate == AOB_ACCUMULATE_QUANTISED) {
        cell->total_bits = (uint32_t)((int32_t)cell->total_bits + qT);
        cell->weight_bits = (uint32_t)((int32_t)cell->weight_bits + qW);
    } else {
        float a, b;
        memcpy(&a, &cell->total_bits, 4); memcpy(&b, &cell->weight_bits, 4);
        a += (float)xT * 2.3283064365386962890625e-10f; b += (float)xW * 2.3283064365386962890625e-10f;
        memcpy(&cell->total_bits, &a, 4); memcpy(&cell->weight_bits, &b, 4);
    }
}

static void* trace_worker(void* arg)
{
    TraceJob* job = (TraceJob*)arg;
    OrcTraceStats st; memset(&st, 0, sizeof st);
    int error = 0;
    CosineGrid grid;
    cosine_grid(job->opt->sample_count, &grid.sx, &grid.sy);
    grid.inv_sx = 1.0f / (float)grid.sx; grid.inv_sy = 1.0f / (float)grid.sy;
    uint64_t chunk_max = chunk_texel_max(job->opt);
    const uint64_t grain = 64;
    for (;;) {
        pthread_mutex_lock(&job->lock);
        uint64_t b = job->next;
        job->next = b + grain;
        pthread_mutex_unlock(&job->lock);
        if (b >= job->end) break;
        uint64_t e = b + grain < job->end ? b + grain : job->end;
        for (uint64_t k = b; k < e; k++) trace_surfel(job, k, &st, &error, &grid, chunk_max);
    }
    pthread_mutex_lock(&job->lock);
    job->stats.rays += st.rays; job->stats.inner_visits += st.inner_visits;
    job->stats.leaf_visits += st.leaf_visits; job->stats.hits += st.hits;
    if (error) job->error = error;
    pthread_mutex_unlock(&job->lock);
    return NULL;
}

int orc_trace(const AobBvhNode* nodes, uint64_t node_count,
              const AobSurfel* surfels, const uint32_t* section_index,
              uint64_t begin, uint64_t end, const AobOptions* opt, const float* hemisphere,
              int32_t width, int32_t height, AobResult* results, int32_t threads, OrcTraceStats* stats)
{
    return orc_trace_ex(nodes, node_count, surfels, section_index, NULL, NULL, 0, 0.0f, begin, end, opt, hemisphere, width, height,
                        results, threads, stats);
}

int orc_trace_ex(const AobBvhNode* nodes, uint64_t node_count,
                 const AobSurfel* surfels, const uint32_t* section_index,
                 const uint32_t* surfel_mesh, const uint32_t* mesh_flags, uint32_t mesh_count, float ao_max_range,
                 uint64_t begin, uint64_t end, const AobOptions* opt, const float* hemisphere,
                 int32_t width, int32_t height, AobResult* results, int32_t threads, OrcTraceStats* stats)
{
    if (opt->sampler == AOB_SAMPLER_REFERENCE_COMPAT && !hemisphere && end > begin) return AOB_ERR_INVALID_ARGUMENT;
    TraceJob job;
    memset(&job, 0, sizeof job);
    job.surfel_mesh = surfel_mesh; job.mesh_flags = mesh_flags; job.mesh_count = mesh_count; job.ao_max_range = ao_max_range;
    job.nodes = nodes; job.node_count = node_count; job.surfels = surfels; job.section_index = section_index;
    job.opt = opt; job.hemisphere = hemisphere; job.width = width; job.height = height; job.results = results;
    job.begin = begin; job.end = end; job.next = begin;
    AobBvh4Node* nodes4 = NULL; AobLeafTri* tris = NULL;
    if (opt->bvh_width == 4 && node_count) {
        const uint64_t n = (node_count + 1) / 2;
        nodes4 = (AobBvh4Node*)malloc(sizeof(AobBvh4Node) * (n > 1 ? n - 1 : 1));
        tris = (AobLeafTri*)malloc(sizeof(AobLeafTri) * n);
        job.node4_count = orc_bvh4_build_ex(nodes, node_count, nodes4, tris, opt->bvh_quantised != 0);
        job.nodes4 = nodes4; job.tris = tris;
    }
    AobBvh4QNode* nodes4q = NULL;
    if (opt->bvh_width == 4 && opt->bvh_quantised && node_count) {
        AobBvh2QNode* scratch2q = (AobBvh2QNode*)malloc(sizeof(AobBvh2QNode) * (node_count / 2 + 1));
        orc_bvh2q_build(nodes, node_count, scratch2q, &job.frame);         /* for the frame */
        free(scratch2q);
        nodes4q = (AobBvh4QNode*)malloc(sizeof(AobBvh4QNode) * job.node4_count);
        orc_bvh4q_build(nodes4, job.node4_count, &job.frame, nodes4q);
        job.nodes4q = nodes4q;
    }
    AobBvh2QNode* nodes2q = NULL;
    if (opt->bvh_width != 4 && opt->bvh_quantised && node_count) {
        const uint64_t n = (node_count + 1) / 2;
        AobBvh4Node* scratch4 = (AobBvh4Node*)malloc(sizeof(AobBvh4Node) * (n > 1 ? n - 1 : 1));
        tris = (AobLeafTri*)malloc(sizeof(AobLeafTri) * n);
        orc_bvh4_build(nodes, node_count, scratch4, tris);          /* for the Morton-ordered triangle array */
        free(scratch4);
        nodes2q = (AobBvh2QNode*)malloc(sizeof(AobBvh2QNode) * (n > 1 ? n - 1 : 1));
        orc_bvh2q_build(nodes, node_count, nodes2q, &job.frame);
        job.nodes2q = nodes2q; job.tris = tris; job.tri_count = n;
    }
    pthread_mutex_init(&job.lock, NULL);
    if (threads > 256) threads = 256;
    /* One section at a time, in scene order: two sections of a material may cover the same texel (BakeEngine.cpp:473,659),
     * within a section every surfel has its own cell.  The reference's integer sums do not care about the order; the
     * float sums are rounded per add, so the order is part of the result — sections in order, as the library's one
     * launch per section.  A section starts where the surfel's index within its section is 0. */
    for (uint64_t sb = begin; sb < end; ) {
        uint64_t se = sb + 1;
        while (se < end && !(section_index && section_index[se] == 0)) se++;
        job.begin = sb; job.end = se; job.next = sb;
        if (threads <= 1) trace_worker(&job);
        else {
            pthread_t th[256];
            for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, trace_worker, &job);
            for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
        }
        sb = se;
    }
    pthread_mutex_destroy(&job.lock);
    free(nodes4); free(tris); free(nodes2q); free(nodes4q);
    if (stats) {
        stats->rays += job.stats.rays; stats->inner_visits += job.stats.inner_visits;
        stats->leaf_visits += job.stats.leaf_visits; stats->hits += job.stats.hits;
    }
    return job.error;
}

/* ------------------------------------------------------------------------ */
/* the reference's own records: rays in, intersections out                   */

/* The tangent frame of a surfel (GenerateRaysComputeShader.hlsl:34-41); 0 when the frame is not finite. */
static inline int surfel_frame(const AobSurfel* s, f3* P, f3* N, f3* X, f3* Z)
{
    *P = f3_make(s->position[0], s->position[1], s->position[2]);
    *N = f3_make(s->normal[0], s->normal[1], s->normal[2]);
    *X = f3_make(1.0f, 0.0f, 0.0f); *Z = f3_make(0.0f, 0.0f, 1.0f);
    if (N->y < 0.9999f && N->y > -0.9999f) {
        *X = f3_normalize(f3_make(-N->z, 0.0f, N->x));
        *Z = f3_normalize(f3_cross(*X, *N));
    }
    return f3_finite(*P) && f3_finite(*N) && f3_finite(*X) && f3_finite(*Z);
}

/* GenerateRaysComputeShader.hlsl:43-54: the 48-byte RadeonRays ray records of surfels [begin, end), sample_count
 * records per surfel, exactly the rays orc_trace casts.  `mask` goes to extra.x (the reference writes 0, :51).  A
 * surfel without a finite frame gets inactive records (extra.y = 0): the kernels skip them. */
uint64_t orc_generate_rays(const AobSurfel* surfels, const uint32_t* section_index, uint64_t begin, uint64_t end,
                           const AobOptions* opt, const float* hemisphere, int32_t width, float t_max, int32_t mask,
                           OrcRay* out)
{
    const int32_t spp = opt->sample_count;
    const int compat = opt->sampler == AOB_SAMPLER_REFERENCE_COMPAT;
    if (compat && !hemisphere && end > begin) return 0;
    CosineGrid grid;
    cosine_grid(spp, &grid.sx, &grid.sy);
    grid.inv_sx = 1.0f / (float)grid.sx; grid.inv_sy = 1.0f / (float)grid.sy;
    const uint64_t chunk_max = chunk_texel_max(opt);
    uint64_t n = 0;
    for (uint64_t k = begin; k < end; k++) {
        const AobSurfel* s = &surfels[k];
        f3 P, N, X, Z;
        const int ok = surfel_frame(s, &P, &N, &X, &Z);
        const uint32_t texel = (uint32_t)s->texel_y * (uint32_t)width + (uint32_t)s->texel_x;
        const uint32_t base = pcg_hash(texel + pcg_hash(opt->seed));
        const uint64_t local = section_index ? section_index[k] : k;
        const float* table = compat ? hemisphere + 3 * ((local % chunk_max) * (uint64_t)spp) : NULL;
        for (int32_t i = 0; i < spp; i++, n++) {
            OrcRay* r = &out[n];
            memset(r, 0, sizeof *r);
            if (!ok) continue;
            f3 h = compat ? f3_make(table[3 * i], table[3 * i + 1], table[3 * i + 2]) : cosine_sample(base, i, &grid);
            f3 dir = f3_make(fmaf(h.z, Z.x, fmaf(h.y, N.x, h.x * X.x)),
                             fmaf(h.z, Z.y, fmaf(h.y, N.y, h.x * X.y)),
                             fmaf(h.z, Z.z, fmaf(h.y, N.z, h.x * X.z)));                                   /* :46 */
            r->o[0] = fmaf(dir.x, opt->ray_bias, P.x); r->o[1] = fmaf(dir.y, opt->ray_bias, P.y);
            r->o[2] = fmaf(dir.z, opt->ray_bias, P.z); r->o[3] = t_max;                                    /* :47,49 */
            r->d[0] = dir.x; r->d[1] = dir.y; r->d[2] = dir.z; r->d[3] = 0.0f;                             /* :50 */
            r->extra[0] = mask; r->extra[1] = (int32_t)0xCDCDCDCDu;                                        /* :51 */
        }
    }
    return n;
}

/* common.cl:249-277, USE_SAFE_MATH branch */
static inline void triangle_barycentrics(f3 p, f3 v1, f3 v2, f3 v3, float* b1, float* b2)
{
    f3 e1 = f3_sub(v2, v1), e2 = f3_sub(v3, v1), e = f3_sub(p, v1);
    float d00 = f3_dot(e1, e1), d01 = f3_dot(e1, e2), d11 = f3_dot(e2, e2), d20 = f3_dot(e, e1), d21 = f3_dot(e, e2);
    float denom = fmaf(d00, d11, -(d01 * d01));
    if (denom == 0.0f) { *b1 = 0.0f; *b2 = 0.0f; return; }
    float invdenom = 1.0f / denom;
    *b1 = fmaf(d11, d20, -(d01 * d21)) * invdenom;
    *b2 = fmaf(d00, d21, -(d01 * d20)) * invdenom;
}

/* intersect_main (intersect_bvh2_lds.cl:65-226) over ray records: the 32-byte Intersection of every active ray
 * (:201-223); inactive rays leave their record untouched.  use_mask: the RR_RAY_MASK build (:163-166) — a ray skips
 * the triangles of the mesh whose id equals extra.x. */
void orc_intersect_rays(const AobBvhNode* nodes, uint64_t node_count, const OrcRay* rays, uint64_t n, int32_t use_mask,
                        OrcIntersection* hits)
{
    OrcTraceStats st; memset(&st, 0, sizeof st);
    for (uint64_t i = 0; i < n; i++) {
        const OrcRay* r = &rays[i];
        if (r->extra[1] == 0) continue;                                  /* ray_is_active, :88 */
        f3 o = f3_make(r->o[0], r->o[1], r->o[2]), d = f3_make(r->d[0], r->d[1], r->d[2]);
        uint32_t flags_self = AOB_MESH_HIDDEN_FROM_SELF;
        OrcMask m = { &flags_self, 1u, 1u };
        float t; uint32_t addr; int error = 0;
        int hit;
        if (use_mask && r->extra[0] > 0) {
            /* one hidden shape: present it to mask_skips as "shape extra.x hidden from itself" */
            const uint32_t shape = (uint32_t)r->extra[0];
            uint32_t* flags = (uint32_t*)calloc(shape, sizeof(uint32_t));
            flags[shape - 1] = AOB_MESH_HIDDEN_FROM_SELF;
            m.mesh_flags = flags; m.mesh_count = shape; m.self_shape = shape;
            hit = traverse_addr(nodes, node_count, o, d, r->o[3], 0, &t, &st, &error, &m, &addr);
            free(flags);
        } else {
            hit = traverse_addr(nodes, node_count, o, d, r->o[3], 0, &t, &st, &error, NULL, &addr);
        }
        OrcIntersection* h = &hits[i];
        if (hit) {
            const AobBvhNode* nd = &nodes[addr];
            f3 p = f3_make(fmaf(t, d.x, o.x), fmaf(t, d.y, o.y), fmaf(t, d.z, o.z));        /* :207 o + t*d */
            float b1, b2;
            triangle_barycentrics(p, f3_make(nd->lmin_or_v0[0], nd->lmin_or_v0[1], nd->lmin_or_v0[2]),
                                  f3_make(nd->lmax_or_v1[0], nd->lmax_or_v1[1], nd->lmax_or_v1[2]),
                                  f3_make(nd->rmin_or_v2[0], nd->rmin_or_v2[1], nd->rmin_or_v2[2]), &b1, &b2);
            h->prim_id = (int32_t)nd->prim_id; h->shape_id = (int32_t)nd->shape_id;
            h->uvwt[0] = b1; h->uvwt[1] = b2; h->uvwt[2] = 0.0f; h->uvwt[3] = t;
        } else {
            h->prim_id = -1; h->shape_id = -1;                            /* MISS_MARKER, :221-222 */
        }
    }
}

/* occluded_main (intersect_bvh2_lds.cl:228-363): 1 (HIT_MARKER) or -1 (MISS_MARKER) per active ray. */
void orc_occluded_rays(const AobBvhNode* nodes, uint64_t node_count, const OrcRay* rays, uint64_t n, int32_t* hits)
{
    OrcTraceStats st; memset(&st, 0, sizeof st);
    for (uint64_t i = 0; i < n; i++) {
        const OrcRay* r = &rays[i];
        if (r->extra[1] == 0) continue;
        float t; int error = 0;
        int hit = traverse(nodes, node_count, f3_make(r->o[0], r->o[1], r->o[2]), f3_make(r->d[0], r->d[1], r->d[2]),
                           r->o[3], 1, &t, &st, &error, NULL);
        hits[i] = hit ? 1 : -1;
    }
}

/* The deferred-entry count the reference's walk reaches for every active ray (closest hit, or any hit): the kernel keeps
 * 15 entries + the sentinel per work item in LDS and, when a 16th is pushed, moves the 15 to its global stack
 * (intersect_bvh2_lds.cl:144-153; brought back at :186-196), whose STACK_SIZE 32 words per ray hold two such moves —
 * so a ray of depth >= 16 exercises the spill and one of depth > 45 would overrun the reference's own stack.  The walk
 * is traverse()'s; only the high-water mark is kept.  Inactive rays get 0. */
void orc_reference_stack_depths(const AobBvhNode* nodes, uint64_t node_count, const OrcRay* rays, uint64_t n, int32_t any_hit,
                                uint32_t* depth_out)
{
    for (uint64_t i = 0; i < n; i++) {
        const OrcRay* r = &rays[i];
        depth_out[i] = 0;
        if (r->extra[1] == 0 || node_count == 0) continue;
        const f3 o = f3_make(r->o[0], r->o[1], r->o[2]), d = f3_make(r->d[0], r->d[1], r->d[2]);
        const f3 invdir = safe_invdir(d);
        const f3 oxinvdir = f3_make(-o.x * invdir.x, -o.y * invdir.y, -o.z * invdir.z);
        float closest_t = r->o[3];
        uint32_t stack[ORC_STACK];
        int sp = 0, high = 0;
        stack[sp++] = AOB_INVALID_ADDR;
        uint32_t addr = 0;
        while (addr != AOB_INVALID_ADDR) {
            const AobBvhNode* nd = &nodes[addr];
            if (nd->addr_left != AOB_INVALID_ADDR) {
                float s0x, s0y, s1x, s1y;
                intersect_bbox(nd->lmin_or_v0, nd->lmax_or_v1, invdir, oxinvdir, closest_t, &s0x, &s0y);
                intersect_bbox(nd->rmin_or_v2, nd->rmax, invdir, oxinvdir, closest_t, &s1x, &s1y);
                const int c0 = s0x <= s0y, c1 = s1x <= s1y;
                const int c1first = c1 && (s0x > s1x);
                if (c0 || c1) {
                    uint32_t deferred;
                    if (c1first || !c0) { addr = nd->addr_right; deferred = nd->addr_left; }
                    else { addr = nd->addr_left; deferred = nd->addr_right; }
                    if (c0 && c1) {
                        if (sp >= ORC_STACK) break;
                        stack[sp++] = deferred;
                        if (sp - 1 > high) high = sp - 1;
                    }
                    continue;
                }
            } else {
                const float t = intersect_triangle(o, d, f3_make(nd->lmin_or_v0[0], nd->lmin_or_v0[1], nd->lmin_or_v0[2]),
                                                   f3_make(nd->lmax_or_v1[0], nd->lmax_or_v1[1], nd->lmax_or_v1[2]),
                                                   f3_make(nd->rmin_or_v2[0], nd->rmin_or_v2[1], nd->rmin_or_v2[2]), closest_t);
                if (t < closest_t) { closest_t = t; if (any_hit) break; }
            }
            addr = stack[--sp];
        }
        depth_out[i] = (uint32_t)high;
    }
}

/* ProcessIntersectionsComputeShader.hlsl:52-92 over the reference's own records: scores the rays of surfels
 * [begin, end) from their Intersection records (shape_id == -1: miss; t in uvwt.w) and accumulates into `results`
 * exactly as orc_trace does — so cells made from the REFERENCE kernel's hits can be compared with the library's. */
void orc_process_intersections(const AobSurfel* surfels, uint64_t begin, uint64_t end, const OrcRay* rays,
                               const OrcIntersection* hits, const AobOptions* opt, float falloff,
                               int32_t width, int32_t height, AobResult* results)
{
    const int32_t spp = opt->sample_count;
    const int compat = opt->sampler == AOB_SAMPLER_REFERENCE_COMPAT;
    const int any_hit = opt->use_any_hit != 0;
    for (uint64_t k = begin; k < end; k++) {
        const AobSurfel* s = &surfels[k];
        if (s->texel_x < 0 || s->texel_y < 0 || s->texel_x >= width || s->texel_y >= height) continue;
        AobResult* cell = &results[(size_t)s->texel_y * width + s->texel_x];
        f3 N = f3_make(s->normal[0], s->normal[1], s->normal[2]);
        int32_t qT = 0, qW = 0; int64_t xT = 0, xW = 0;
        int any = 0;
        for (int32_t i = 0; i < spp; i++) {
            const uint64_t ri = (k - begin) * (uint64_t)spp + (uint64_t)i;
            const OrcRay* r = &rays[ri];
            if (r->extra[1] == 0) continue;
            any = 1;
            const int hit = hits[ri].shape_id != -1;
            const float t = hits[ri].uvwt[3];
            float w = compat ? f3_dot(N, f3_make(r->d[0], r->d[1], r->d[2])) : 1.0f;
            float rr;
            if (any_hit) rr = hit ? 0.0f : 1.0f;
            else rr = (hit && t < falloff) ? t / falloff : 1.0f;
            if (opt->accumulate == AOB_ACCUMULATE_QUANTISED) { qT += orc_ftoi(rr * w * 255.0f); qW += orc_ftoi(w * 255.0f); }
            else { xT += f2fix(rr * w); xW += f2fix(w); }
        }
        if (!any) continue;
        if (opt->accumulate == AOB_ACCUMULATE_QUANTISED) {
            cell->total_bits = (uint32_t)((int32_t)cell->total_bits + qT);
            cell->weight_bits = (uint32_t)((int32_t)cell->weight_bits + qW);
        } else {
            float a, b;
            memcpy(&a, &cell->total_bits, 4); memcpy(&b, &cell->weight_bits, 4);
            a += (float)xT * 2.3283064365386962890625e-10f; b += (float)xW * 2.3283064365386962890625e-10f;
            memcpy(&cell->total_bits, &a, 4); memcpy(&cell->weight_bits, &b, 4);
        }
    }
}

/* The 63-bit Morton codes orc_bvh_build_ex sorts by (primitive order, before the sort) and the primitive and scene
 * boxes they are made from — build_hlbvh.cl:101-125 makes the top 30 bits of the non-adaptive code. */
void orc_morton_codes(const OrcTriangle* tris, uint64_t n, int adaptive, uint64_t* codes, float* prim_boxes8, float* scene_box8)
{
    Box scene;
    for (int k = 0; k < 3; k++) { scene.pmin[k] = INFINITY; scene.pmax[k] = -INFINITY; }
    Box* pb = (Box*)malloc(sizeof(Box) * (n ? n : 1));
    for (uint64_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            pb[i].pmin[k] = fminf(fminf(tris[i].v[0][k], tris[i].v[1][k]), tris[i].v[2][k]);
            pb[i].pmax[k] = fmaxf(fmaxf(tris[i].v[0][k], tris[i].v[1][k]), tris[i].v[2][k]);
            scene.pmin[k] = fminf(scene.pmin[k], pb[i].pmin[k]);
            scene.pmax[k] = fmaxf(scene.pmax[k], pb[i].pmax[k]);
        }
    float ext[3];
    for (int k = 0; k < 3; k++) ext[k] = scene.pmax[k] - scene.pmin[k];
    for (uint64_t i = 0; i < n; i++) {
        uint32_t q[3];
        for (int k = 0; k < 3; k++) q[k] = morton_axis((pb[i].pmax[k] + pb[i].pmin[k]) * 0.5f, scene.pmin[k], ext[k]);
        if (adaptive) {
            float e[3] = { ext[0], ext[1], ext[2] };
            int used[3] = { 0, 0, 0 };
            uint64_t code = 0;
            for (int bit = 0; bit < 63; bit++) {
                int ax = -1; float best = -1.0f;
                for (int k = 0; k < 3; k++) if (used[k] < 21 && e[k] > best) { best = e[k]; ax = k; }
                code = code << 1 | ((q[ax] >> (20 - used[ax])) & 1u);
                used[ax]++; e[ax] *= 0.5f;
            }
            codes[i] = code;
        } else codes[i] = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
        if (prim_boxes8) {       /* the reference's bbox record: float4 pmin, float4 pmax (common.cl:46-50) */
            float* b = prim_boxes8 + 8 * i;
            b[0] = pb[i].pmin[0]; b[1] = pb[i].pmin[1]; b[2] = pb[i].pmin[2]; b[3] = 0.0f;
            b[4] = pb[i].pmax[0]; b[5] = pb[i].pmax[1]; b[6] = pb[i].pmax[2]; b[7] = 0.0f;
        }
    }
    if (scene_box8) {
        scene_box8[0] = scene.pmin[0]; scene_box8[1] = scene.pmin[1]; scene_box8[2] = scene.pmin[2]; scene_box8[3] = 0.0f;
        scene_box8[4] = scene.pmax[0]; scene_box8[5] = scene.pmax[1]; scene_box8[6] = scene.pmax[2]; scene_box8[7] = 0.0f;
    }
    free(pb);
}

/* ------------------------------------------------------------------------ */
/* two-level walk (instancing): top-level tree over instances, shared bottom-level trees   */

/* The structure of kernels/CL/intersect_bvh2level_skiplinks.cl:134-307 — a top-level hierarchy whose leaves are shapes
 * with a world-to-object transform and a bottom-level tree of their own (Shape :75-92), the ray transformed on entry
 * (transform_ray :123-131, :246-253: point and vector by the inverse matrix rows, NOT renormalised, so t keeps its
 * meaning) and restored on return (:278-288) — walked here with the SAME traversal discipline as the flat tree
 * (traverse(): child boxes in the parent, near child first, stack), so that visit counts of the two structures compare
 * like for like.  The top-level tree is an AobBvhNode[] whose "triangles" are the instances' world boxes
 * (v0 = box min, v1 = box max; prim_id = instance index); a leaf there means "enter the instance".
 * st_top counts top-level inner visits in inner_visits and instance entries in leaf_visits; st_bottom counts the
 * bottom-level trees' inner visits and triangle tests. */
int orc_two_level_intersect(const AobBvhNode* tlas, uint64_t tlas_count, const OrcInstance* instances,
                            const AobBvhNode* const* blas_nodes, const uint64_t* blas_counts,
                            const OrcRay* rays, uint64_t n, int32_t any_hit, OrcIntersection* hits,
                            OrcTraceStats* st_top, OrcTraceStats* st_bottom)
{
    int error = 0;
    for (uint64_t i = 0; i < n; i++) {
        const OrcRay* r = &rays[i];
        if (r->extra[1] == 0) continue;
        st_top->rays++;
        const f3 o = f3_make(r->o[0], r->o[1], r->o[2]), d = f3_make(r->d[0], r->d[1], r->d[2]);
        float closest_t = r->o[3];
        int hit = 0; uint32_t hit_shape = 0, hit_prim = 0;
        if (tlas_count) {
            const f3 invdir = safe_invdir(d);
            const f3 oxinvdir = f3_make(-o.x * invdir.x, -o.y * invdir.y, -o.z * invdir.z);
            uint32_t stack[ORC_STACK]; int sp = 0;
            stack[sp++] = AOB_INVALID_ADDR;
            uint32_t addr = 0;
            while (addr != AOB_INVALID_ADDR) {
                const AobBvhNode* nd = &tlas[addr];
                if (nd->addr_left != AOB_INVALID_ADDR) {
                    st_top->inner_visits++;
                    float s0x, s0y, s1x, s1y;
                    intersect_bbox(nd->lmin_or_v0, nd->lmax_or_v1, invdir, oxinvdir, closest_t, &s0x, &s0y);
                    intersect_bbox(nd->rmin_or_v2, nd->rmax, invdir, oxinvdir, closest_t, &s1x, &s1y);
                    const int c0 = s0x <= s0y, c1 = s1x <= s1y, c1first = c1 && (s0x > s1x);
                    if (c0 || c1) {
                        uint32_t deferred;
                        if (c1first || !c0) { addr = nd->addr_right; deferred = nd->addr_left; }
                        else { addr = nd->addr_left; deferred = nd->addr_right; }
                        if (c0 && c1) { if (sp >= ORC_STACK) return AOB_ERR_BVH_TOO_DEEP; stack[sp++] = deferred; }
                        continue;
                    }
                } else {
                    /* a top-level leaf: enter the instance (intersect_bvh2level_skiplinks.cl:225-256) */
                    st_top->leaf_visits++;
                    const OrcInstance* in = &instances[nd->prim_id];
                    const float* m = in->world_to_object;
                    /* transform_point / transform_vector (:105-121) are written `m.s0*p.x + m.s1*p.y + m.s2*p.z (+ m.s3)`; OpenCL C
                     * contracts that to fma(m.s2, p.z, fma(m.s0, p.x, m.s1*p.y)) (+ m.s3) — the SECOND product is the rounded one
                     * (the ROCm clang's code for the reference's source; the GPU test pins it) */
                    const f3 oo = f3_make(fmaf(m[2], o.z, fmaf(m[0], o.x, m[1] * o.y)) + m[3],
                                          fmaf(m[6], o.z, fmaf(m[4], o.x, m[5] * o.y)) + m[7],
                                          fmaf(m[10], o.z, fmaf(m[8], o.x, m[9] * o.y)) + m[11]);
                    const f3 od = f3_make(fmaf(m[2], d.z, fmaf(m[0], d.x, m[1] * d.y)),
                                          fmaf(m[6], d.z, fmaf(m[4], d.x, m[5] * d.y)),
                                          fmaf(m[10], d.z, fmaf(m[8], d.x, m[9] * d.y)));
                    float t; uint32_t leaf;
                    const int h = traverse_addr(blas_nodes[in->blas], blas_counts[in->blas], oo, od, closest_t, any_hit, &t, st_bottom,
                                                &error, NULL, &leaf);
                    if (h && t < closest_t) {
                        closest_t = t; hit = 1; hit_shape = in->shape_id; hit_prim = blas_nodes[in->blas][leaf].prim_id;
                        if (any_hit) break;
                    }
                }
                addr = stack[--sp];
            }
        }
        st_top->hits += (uint64_t)hit;
        if (hits) {
            OrcIntersection* hrec = &hits[i];
            if (hit) { hrec->shape_id = (int32_t)hit_shape; hrec->prim_id = (int32_t)hit_prim; hrec->uvwt[0] = hrec->uvwt[1] = hrec->uvwt[2] = 0.0f; hrec->uvwt[3] = closest_t; }
            else { hrec->shape_id = -1; hrec->prim_id = -1; }
        }
    }
    return error;
}

/* traverse() over ray records with visit counts (the flat tree's side of the comparison). */
int orc_flat_intersect(const AobBvhNode* nodes, uint64_t node_count, const OrcRay* rays, uint64_t n, int32_t any_hit,
                       OrcIntersection* hits, OrcTraceStats* st)
{
    int error = 0;
    for (uint64_t i = 0; i < n; i++) {
        const OrcRay* r = &rays[i];
        if (r->extra[1] == 0) continue;
        st->rays++;
        float t; uint32_t addr;
        const int hit = traverse_addr(nodes, node_count, f3_make(r->o[0], r->o[1], r->o[2]), f3_make(r->d[0], r->d[1], r->d[2]),
                                      r->o[3], any_hit, &t, st, &error, NULL, &addr);
        st->hits += (uint64_t)hit;
        if (hits) {
            if (hit) { hits[i].shape_id = (int32_t)nodes[addr].shape_id; hits[i].prim_id = (int32_t)nodes[addr].prim_id; hits[i].uvwt[3] = t; }
            else { hits[i].shape_id = -1; hits[i].prim_id = -1; }
        }
    }
    return error;
}

/* ------------------------------------------------------------------------ */
/* measurement helpers (tests/measure_tree_variants.py): what a change of tree or node width would buy, counted on   */
/* the CPU before anything is built for the GPU                                                                       */

/* traverse4q over ray records on GIVEN arrays (so that a caller can permute child slots, or pass another tree's
 * arrays); stats of all rays, and of the rays that hit alone. */
int orc_walk4q(const AobBvh4QNode* nq, uint64_t count4, const AobQuantFrame* fr, const AobLeafTri* tris,
               const OrcRay* rays, uint64_t n, int32_t any_hit, OrcTraceStats* all, OrcTraceStats* of_hit_rays)
{
    int error = 0;
    for (uint64_t i = 0; i < n; i++) {
        const OrcRay* r = &rays[i];
        if (r->extra[1] == 0) continue;
        OrcTraceStats st; memset(&st, 0, sizeof st);
        float t;
        const int hit = traverse4q(nq, count4, fr, tris, f3_make(r->o[0], r->o[1], r->o[2]), f3_make(r->d[0], r->d[1], r->d[2]),
                                   r->o[3], any_hit, &t, &st, &error, NULL);
        all->rays++; all->inner_visits += st.inner_visits; all->leaf_visits += st.leaf_visits; all->hits += (uint64_t)hit;
        if (hit && of_hit_rays) { of_hit_rays->rays++; of_hit_rays->inner_visits += st.inner_visits; of_hit_rays->leaf_visits += st.leaf_visits; of_hit_rays->hits++; }
    }
    return error;
}

/* ... and every ray's closest distance (t_max where it hit nothing): what a closest-hit walk from a tile's cut must reproduce */
int orc_walk4q_t(const AobBvh4QNode* nq, uint64_t count4, const AobQuantFrame* fr, const AobLeafTri* tris,
                 const OrcRay* rays, uint64_t n, int32_t any_hit, OrcTraceStats* all, float* t_out)
{
    int error = 0;
    for (uint64_t i = 0; i < n; i++) {
        const OrcRay* r = &rays[i];
        t_out[i] = r->o[3];
        if (r->extra[1] == 0) continue;
        OrcTraceStats st; memset(&st, 0, sizeof st);
        float t;
        const int hit = traverse4q(nq, count4, fr, tris, f3_make(r->o[0], r->o[1], r->o[2]), f3_make(r->d[0], r->d[1], r->d[2]),
                                   r->o[3], any_hit, &t, &st, &error, NULL);
        all->rays++; all->inner_visits += st.inner_visits; all->leaf_visits += st.leaf_visits; all->hits += (uint64_t)hit;
        t_out[i] = t;
    }
    return error;
}

/* The same walk started from a list of refs instead of the root (tests/measure_region_cut.py: what starting every ray of a tile
 * below the nodes its rays cannot leave would save); t_sum: the hit distances added up in double, to compare with the plain walk. */
int orc_walk4q_from(const AobBvh4QNode* nq, uint64_t count4, const AobQuantFrame* fr, const AobLeafTri* tris,
                    const OrcRay* rays, uint64_t n, int32_t any_hit, const uint32_t* start, int32_t n_start, OrcTraceStats* all, double* t_sum)
{
    int error = 0;
    for (uint64_t i = 0; i < n; i++) {
        const OrcRay* r = &rays[i];
        if (r->extra[1] == 0) continue;
        OrcTraceStats st; memset(&st, 0, sizeof st);
        float t;
        const int hit = traverse4q_from(nq, count4, fr, tris, f3_make(r->o[0], r->o[1], r->o[2]), f3_make(r->d[0], r->d[1], r->d[2]),
                                        r->o[3], any_hit, &t, &st, &error, NULL, start, n_start);
        all->rays++; all->inner_visits += st.inner_visits; all->leaf_visits += st.leaf_visits; all->hits += (uint64_t)hit;
        if (hit && t_sum) *t_sum += (double)t;
    }
    return error;
}

/* Node visits of a 2^levels-ary collapse of a BVH2 (every `levels` levels become one node; leaf pairs one slot), walked
 * any-hit in slot order or closest-hit nearest first, on the fp32 boxes.  inner_visits = wide nodes fetched,
 * leaf_visits = triangle tests, and *box_tests = child boxes tested (a wide node tests all its children). */
typedef struct { uint32_t ref; float pmin[3], pmax[3]; } KChild;
static int kary_children(const AobBvhNode* nodes2, uint32_t leaf_base, uint32_t root, int levels, int pairs, KChild* out)
{
    /* breadth-wise expansion of `root` by `levels` levels; a leaf or a pair stops early */
    int k = 0;
    const AobBvhNode* me = &nodes2[root];
    out[k].ref = me->addr_left; memcpy(out[k].pmin, me->lmin_or_v0, 12); memcpy(out[k].pmax, me->lmax_or_v1, 12); k++;
    out[k].ref = me->addr_right; memcpy(out[k].pmin, me->rmin_or_v2, 12); memcpy(out[k].pmax, me->rmax, 12); k++;
    for (int l = 1; l < levels; l++) {
        KChild next[64]; int m = 0;
        for (int j = 0; j < k; j++) {
            const uint32_t c = out[j].ref;
            if (c >= leaf_base || (pairs && is_leaf_pair(nodes2, c, leaf_base))) { next[m++] = out[j]; continue; }
            const AobBvhNode* ch = &nodes2[c];
            next[m].ref = ch->addr_left; memcpy(next[m].pmin, ch->lmin_or_v0, 12); memcpy(next[m].pmax, ch->lmax_or_v1, 12); m++;
            next[m].ref = ch->addr_right; memcpy(next[m].pmin, ch->rmin_or_v2, 12); memcpy(next[m].pmax, ch->rmax, 12); m++;
        }
        memcpy(out, next, sizeof(KChild) * (size_t)m); k = m;
    }
    return k;
}

int orc_kary_visits(const AobBvhNode* nodes2, uint64_t node_count2, const OrcRay* rays, uint64_t n, int32_t levels, int32_t pairs,
                    int32_t any_hit, OrcTraceStats* st, uint64_t* box_tests)
{
    if (levels < 1 || levels > 5 || node_count2 < 3) return AOB_ERR_INVALID_ARGUMENT;
    const uint32_t leaf_base = (uint32_t)((node_count2 + 1) / 2 - 1);
    for (uint64_t i = 0; i < n; i++) {
        const OrcRay* r = &rays[i];
        if (r->extra[1] == 0) continue;
        st->rays++;
        const f3 o = f3_make(r->o[0], r->o[1], r->o[2]), d = f3_make(r->d[0], r->d[1], r->d[2]);
        const f3 invdir = safe_invdir(d);
        const f3 oxinvdir = f3_make(-o.x * invdir.x, -o.y * invdir.y, -o.z * invdir.z);
        float closest_t = r->o[3];
        int hit = 0;
        uint32_t stack[1024]; int sp = 0;
        stack[sp++] = 0;
        while (sp) {
            const uint32_t node = stack[--sp];
            if (node >= leaf_base || (pairs && is_leaf_pair(nodes2, node, leaf_base))) {
                const uint32_t first = node >= leaf_base ? node : nodes2[node].addr_left;
                const int cnt = node >= leaf_base ? 1 : 2;
                float best = closest_t;
                for (int q = 0; q < cnt; q++) {
                    const AobBvhNode* lf = &nodes2[first + (uint32_t)q];
                    st->leaf_visits++;
                    const float t = intersect_triangle(o, d, f3_make(lf->lmin_or_v0[0], lf->lmin_or_v0[1], lf->lmin_or_v0[2]),
                                                       f3_make(lf->lmax_or_v1[0], lf->lmax_or_v1[1], lf->lmax_or_v1[2]),
                                                       f3_make(lf->rmin_or_v2[0], lf->rmin_or_v2[1], lf->rmin_or_v2[2]), closest_t);
                    if (t < best) best = t;
                }
                if (best < closest_t) { closest_t = best; hit = 1; if (any_hit) break; }
                continue;
            }
            st->inner_visits++;
            KChild ch[64];
            const int k = kary_children(nodes2, leaf_base, node, levels, pairs, ch);
            *box_tests += (uint64_t)k;
            float key[64]; int idx[64]; int m = 0;
            for (int j = 0; j < k; j++) {
                float t0, t1;
                intersect_bbox(ch[j].pmin, ch[j].pmax, invdir, oxinvdir, closest_t, &t0, &t1);
                if (t0 <= t1) { key[m] = any_hit ? (float)j : t0; idx[m] = j; m++; }
            }
            /* defer farthest (or last) first so that the nearest (or first) is popped next */
            for (int a = 1; a < m; a++) { const float kk = key[a]; const int ii = idx[a]; int b = a - 1; while (b >= 0 && key[b] > kk) { key[b + 1] = key[b]; idx[b + 1] = idx[b]; b--; } key[b + 1] = kk; idx[b + 1] = ii; }
            if (sp + m > 1024) return AOB_ERR_BVH_TOO_DEEP;
            for (int a = m - 1; a >= 0; a--) stack[sp++] = ch[idx[a]].ref;
        }
        st->hits += (uint64_t)hit;
    }
    return 0;
}

/* The four slab tests of one quantised 4-ary node for a ray prepared as traverse4q prepares it (same arithmetic). */
typedef struct { float fa[3], fb[3]; int down[3]; int any_hit; } Ray4q;
static inline void ray4q_prepare(Ray4q* r, const AobQuantFrame* fr, f3 o, f3 d, float t_max, int any_hit)
{
    f3 invdir = safe_invdir(d);
    const float inv[3] = { invdir.x, invdir.y, invdir.z };
    const float oxi[3] = { -o.x * invdir.x, -o.y * invdir.y, -o.z * invdir.z };
    const float inv_tmax = 1.0f / t_max;
    for (int k = 0; k < 3; k++) {
        const float origin2 = fmaf(-32768.0f, fr->scale[k], fr->origin[k]);
        const float scale2 = fr->scale[k] * 32768.0f;
        r->fb[k] = fmaf(origin2, inv[k], oxi[k]);
        r->fa[k] = scale2 * inv[k];
        if (any_hit) { r->fa[k] = r->fa[k] * inv_tmax; r->fb[k] = r->fb[k] * inv_tmax; }
        uint32_t bits; memcpy(&bits, &inv[k], 4);
        r->down[k] = (int)(bits >> 31);
    }
    r->any_hit = any_hit;
}
static inline int ray4q_box(const Ray4q* r, const uint32_t* c, float closest_t)
{
    float tn[3], tf[3];
    for (int k = 0; k < 3; k++) {
        const uint32_t nearh = r->down[k] ? c[k] >> 16 : c[k] & 0xffffu, farh = r->down[k] ? c[k] & 0xffffu : c[k] >> 16;
        tn[k] = fmaf(plane_float(nearh), r->fa[k], r->fb[k]);
        tf[k] = fmaf(plane_float(farh), r->fa[k], r->fb[k]);
    }
    float t0 = max3n(tn[0], tn[1], tn[2]), t1 = min3n(tf[0], tf[1], tf[2]);
    if (r->any_hit) t0 = t0 != t0 ? 0.0f : fminf(fmaxf(t0, 0.0f), 1.0f);
    else { t0 = fmaxf(t0, 0.0f); t1 = fminf(t1, closest_t); }
    return t0 <= t1;
}

/* Measurement (tests/measure_entry_lists.py): the any-hit walk of k_trace's ENTRY mode, one ray at a time.  `entries` is a
 * tile's entry list (trace.hip k_tile_entries): word e = node | (parent entry * 4 + parent slot) << 26, entry 0 the root.
 * Phase 1 tests the four children of every entry whose own box the ray entered (bit 4e + j of the mask: child j of entry e
 * was hit); phase 2 walks the hit children that are not entries themselves, last bit first, in traverse4q's order, and stops at
 * the first triangle hit.  Counts what each phase costs; `hits` must equal traverse4q's (the visited set is the same). */
int orc_entry_walk(const AobBvh4QNode* nq, uint64_t count4, const AobQuantFrame* fr, const AobLeafTri* tris,
                   const OrcRay* rays, uint64_t n, const uint32_t* entries, int32_t n_entries, OrcEntryStats* out)
{
    if (count4 == 0 || n_entries < 1 || n_entries > 8) return AOB_ERR_INVALID_ARGUMENT;
    uint32_t expanded = 0;
    for (int e = 1; e < n_entries; e++) expanded |= 1u << ((entries[e] >> 26) & 31u);
    for (uint64_t i = 0; i < n; i++) {
        const OrcRay* rr = &rays[i];
        if (rr->extra[1] == 0) continue;
        const f3 o = f3_make(rr->o[0], rr->o[1], rr->o[2]), d = f3_make(rr->d[0], rr->d[1], rr->d[2]);
        const float t_max = rr->o[3];
        Ray4q r; ray4q_prepare(&r, fr, o, d, t_max, 1);
        uint32_t mask = 0;
        for (int e = 0; e < n_entries; e++) {
            const uint32_t node = entries[e] & 0x3ffffffu;
            if (e > 0 && !((mask >> ((entries[e] >> 26) & 31u)) & 1u)) continue;
            out->entries_tested++;
            for (int j = 0; j < 4; j++) if (ray4q_box(&r, &nq[node].w[4 * j], t_max)) mask |= 1u << (4 * e + j);
        }
        mask &= ~expanded;
        out->rays++;
        out->mask_bits += (uint64_t)__builtin_popcount(mask);
        int hit = 0;
        uint32_t stack[ORC_STACK4]; int sp = 0;
        for (int b = 0; b < 32; b++) if ((mask >> b) & 1u) stack[sp++] = nq[entries[b >> 2] & 0x3ffffffu].w[4 * (b & 3) + 3];
        while (sp > 0 && !hit) {
            const uint32_t node = stack[--sp];
            if (!(node & AOB_LEAF_FLAG)) {
                out->inner_visits++;
                int chit[4], nhit = 0;
                for (int j = 0; j < 4; j++) { chit[j] = ray4q_box(&r, &nq[node].w[4 * j], t_max); nhit += chit[j]; }
                if (sp + nhit > ORC_STACK4) return AOB_ERR_BVH_TOO_DEEP;
                for (int j = 3; j >= 0; j--) if (chit[j]) stack[sp++] = nq[node].w[4 * j + 3];
            } else {
                const AobLeafTri* tr = &tris[node & ~(AOB_LEAF_FLAG | AOB_PAIR_FLAG)];
                const int count = (node & AOB_PAIR_FLAG) ? 2 : 1;
                for (int k = 0; k < count; k++) {
                    out->leaf_visits++;
                    const float t = intersect_triangle(o, d, f3_make(tr[k].v0[0], tr[k].v0[1], tr[k].v0[2]), f3_make(tr[k].v1[0], tr[k].v1[1], tr[k].v1[2]),
                                                       f3_make(tr[k].v2[0], tr[k].v2[1], tr[k].v2[2]), t_max);
                    if (t < t_max) hit = 1;
                }
            }
        }
        out->hits += (uint64_t)hit;
    }
    return 0;
}

/* ---- tile cuts (trace.hip, CUT mode: k_tile_cuts + the entry phase of k_trace) ------------------------------------------
 * The rays of one 8x8-texel tile all start inside the same chain of nodes.  A CUT of the tree for a tile is a list of child
 * slots {ref, plane words x, y, z} such that every root-to-leaf path crosses exactly one of them: starting from the root's four
 * slots, the first slot in list order that is an inner node whose box meets (rule 1) or contains (rule 0) the tile's box — the
 * surfels' positions, a margin wider — is replaced by its own slots (appended), while the list fits `max_entries`.  A ray tests
 * the boxes of ALL entries (same arithmetic as a visit) and walks the subtrees of those it enters.  Child boxes are nested in
 * their parent's on the quantisation grid and fma is monotone in the plane, so a slot whose ancestor's test fails fails
 * itself: the triangles a ray reaches are EXACTLY those of the walk from the root (any hit: the answer, too).
 * orc_tile_cut returns the number of entries (4 words each). */
int32_t orc_tile_cut(const AobBvh4QNode* nq, uint64_t count4, const AobQuantFrame* fr, const float lo[3], const float hi[3],
                     int32_t max_entries, int32_t rule, uint32_t* out)
{
    if (count4 == 0 || max_entries < 4) return 0;
    float glo[3], ghi[3];
    for (int k = 0; k < 3; k++) {
        glo[k] = fr->scale[k] > 0.0f ? (lo[k] - fr->origin[k]) / fr->scale[k] : -INFINITY;
        ghi[k] = fr->scale[k] > 0.0f ? (hi[k] - fr->origin[k]) / fr->scale[k] : INFINITY;
    }
    uint32_t list[4 * 128]; uint8_t dead[128]; int n = 0, live = 0;
    /* append the non-empty slots of `node` */
    #define CUT_APPEND(node) do { for (int j_ = 0; j_ < 4; j_++) { const uint32_t* c_ = &nq[node].w[4 * j_]; \
        if (c_[0] == 0x8000ffffu && c_[1] == 0x8000ffffu && c_[2] == 0x8000ffffu) continue; \
        list[4 * n] = c_[3]; list[4 * n + 1] = c_[0]; list[4 * n + 2] = c_[1]; list[4 * n + 3] = c_[2]; dead[n] = 0; n++; live++; } } while (0)
    CUT_APPEND(0);
    for (int e = 0; e < n; e++) {
        const uint32_t ref = list[4 * e];
        if (ref & AOB_LEAF_FLAG) continue;
        int ok = 1;
        for (int k = 0; k < 3; k++) {
            const float qmin = (float)(list[4 * e + 1 + k] & 0x7fffu), qmax = (float)((list[4 * e + 1 + k] >> 16) & 0x7fffu);
            if (rule == 0) ok &= qmin <= glo[k] && qmax >= ghi[k];
            else ok &= qmin <= ghi[k] && qmax >= glo[k];
        }
        if (!ok) continue;
        int kids = 0;
        for (int j = 0; j < 4; j++) { const uint32_t* c = &nq[ref].w[4 * j]; kids += !(c[0] == 0x8000ffffu && c[1] == 0x8000ffffu && c[2] == 0x8000ffffu); }
        if (live - 1 + kids > max_entries || n + kids > 128) continue;
        dead[e] = 1; live--;
        CUT_APPEND(ref);
    }
    #undef CUT_APPEND
    int m = 0;
    for (int e = 0; e < n; e++) if (!dead[e]) { memcpy(&out[4 * m], &list[4 * e], 16); m++; }
    return m;
}

/* The any-hit walk of a batch of rays from a tile's cut, counted: entries tested (= n_cut per ray), entries entered, inner and leaf
 * visits below them.  `hits` must equal the plain walk's. */
int orc_cut_walk(const AobBvh4QNode* nq, uint64_t count4, const AobQuantFrame* fr, const AobLeafTri* tris,
                 const OrcRay* rays, uint64_t n, const uint32_t* cut, int32_t n_cut, OrcEntryStats* out, int32_t* hit_out, uint64_t* mask_out)
{
    if (count4 == 0 || n_cut < 1 || n_cut > 64) return AOB_ERR_INVALID_ARGUMENT;
    for (uint64_t i = 0; i < n; i++) {
        const OrcRay* rr = &rays[i];
        if (hit_out) hit_out[i] = 0;
        if (mask_out) mask_out[i] = 0;
        if (rr->extra[1] == 0) continue;
        const f3 o = f3_make(rr->o[0], rr->o[1], rr->o[2]), d = f3_make(rr->d[0], rr->d[1], rr->d[2]);
        const float t_max = rr->o[3];
        Ray4q r; ray4q_prepare(&r, fr, o, d, t_max, 1);
        uint32_t stack[ORC_STACK4]; int sp = 0;
        out->rays++;
        uint64_t entered = 0;
        /* (the kernel walks the entries a ray entered shallowest first — the list's first entry first: pushed last, popped first — so that the
         * visits counted here are the kernel's; the answer does not depend on it) */
        for (int e = n_cut - 1; e >= 0; e--) {
            out->entries_tested++;
            if (ray4q_box(&r, &cut[4 * e + 1], t_max)) { stack[sp++] = cut[4 * e]; out->mask_bits++; entered |= 1ull << e; }
        }
        if (mask_out) mask_out[i] = entered;
        int hit = 0;
        while (sp > 0 && !hit) {
            const uint32_t node = stack[--sp];
            if (!(node & AOB_LEAF_FLAG)) {
                out->inner_visits++;
                int chit[4], nhit = 0;
                for (int j = 0; j < 4; j++) { chit[j] = ray4q_box(&r, &nq[node].w[4 * j], t_max); nhit += chit[j]; }
                if (sp + nhit > ORC_STACK4) return AOB_ERR_BVH_TOO_DEEP;
                for (int j = 3; j >= 0; j--) if (chit[j]) stack[sp++] = nq[node].w[4 * j + 3];
            } else {
                const AobLeafTri* tr = &tris[node & ~(AOB_LEAF_FLAG | AOB_PAIR_FLAG)];
                const int count = (node & AOB_PAIR_FLAG) ? 2 : 1;
                for (int k = 0; k < count; k++) {
                    out->leaf_visits++;
                    const float t = intersect_triangle(o, d, f3_make(tr[k].v0[0], tr[k].v0[1], tr[k].v0[2]), f3_make(tr[k].v1[0], tr[k].v1[1], tr[k].v1[2]),
                                                       f3_make(tr[k].v2[0], tr[k].v2[1], tr[k].v2[2]), t_max);
                    if (t < t_max) hit = 1;
                }
            }
        }
        out->hits += (uint64_t)hit;
        if (hit_out) hit_out[i] = hit;
    }
    return 0;
}

/* Round 5: a tile's cut for rays of bounded REACH (closest-hit scoring with tmax clamped to the falloff distance, a material range):
 * the twin of trace.hip k_tile_cuts with `reach` > 0.  The tile's rays start at the surfels' positions — inside the FULL box
 * [flo, fhi] — and end within `reach` of them, so a slot whose box lies farther than that from the full box holds no triangle any of
 * them can hit: such a slot is never appended (not when the root's slots start the list, not when an opened entry's slots join it), and
 * an opened entry counts only the slots it appends against the budget.  Everything else is orc_tile_cut's rule 1: breadth first, a level
 * at a time in list order, an inner slot whose box meets the opening box [lo, hi] (the tile's box pulled in) is replaced by its slots
 * while the list fits max_entries.  `grow2` >= 0 adds a second sweep when the first has ended with room to spare: the live inner entries, in
 * list order again, whose box meets the full box grown by grow2 * reach are opened too, level by level as before (with bounded rays most of
 * the first sweep's siblings are dropped, and the budget they free is spent on the tree around the tile, where its rays end).
 * The distance is computed in float exactly as below (both sides; no contraction). */
static inline int cut_within_reach(const AobQuantFrame* fr, const uint32_t* planes, const float flo[3], const float fhi[3], float reach2)
{
    float d2 = 0.0f;
    for (int k = 0; k < 3; k++) {
        const float bmin = fmaf((float)(planes[k] & 0x7fffu), fr->scale[k], fr->origin[k]);
        const float bmax = fmaf((float)((planes[k] >> 16) & 0x7fffu), fr->scale[k], fr->origin[k]);
        const float gap = fmaxf(fmaxf(bmin - fhi[k], flo[k] - bmax), 0.0f);
        d2 = fmaf(gap, gap, d2);
    }
    return d2 <= reach2;
}
int32_t orc_tile_cut_reach(const AobBvh4QNode* nq, uint64_t count4, const AobQuantFrame* fr, const float lo[3], const float hi[3],
                           const float flo[3], const float fhi[3], float reach, int32_t max_entries, float grow2, uint32_t* out)
{
    if (count4 == 0 || max_entries < 4) return 0;
    const int bounded = reach > 0.0f && reach < INFINITY;
    const float reach2 = reach * reach;
    uint32_t list[4 * 128]; uint8_t dead[128]; int n = 0, live = 0;
    #define SLOT_EMPTY(c_) ((c_)[0] == 0x8000ffffu && (c_)[1] == 0x8000ffffu && (c_)[2] == 0x8000ffffu)
    #define SLOT_KEPT(c_) (!SLOT_EMPTY(c_) && (!bounded || cut_within_reach(fr, (c_), flo, fhi, reach2)))
    #define CUT_APPEND(node) do { for (int j_ = 0; j_ < 4; j_++) { const uint32_t* c_ = &nq[node].w[4 * j_]; \
        if (!SLOT_KEPT(c_)) continue; \
        list[4 * n] = c_[3]; list[4 * n + 1] = c_[0]; list[4 * n + 2] = c_[1]; list[4 * n + 3] = c_[2]; dead[n] = 0; n++; live++; } } while (0)
    CUT_APPEND(0);
    for (int sweep = 0; sweep < (grow2 >= 0.0f && bounded ? 2 : 1); sweep++) {
        float glo[3], ghi[3];
        for (int k = 0; k < 3; k++) {
            const float a = sweep ? flo[k] - grow2 * reach : lo[k], b = sweep ? fhi[k] + grow2 * reach : hi[k];
            glo[k] = fr->scale[k] > 0.0f ? (a - fr->origin[k]) / fr->scale[k] : -INFINITY;
            ghi[k] = fr->scale[k] > 0.0f ? (b - fr->origin[k]) / fr->scale[k] : INFINITY;
        }
        for (int e = 0; e < n; e++) {
            const uint32_t ref = list[4 * e];
            if (dead[e] || (ref & AOB_LEAF_FLAG)) continue;
            int ok = 1;
            for (int k = 0; k < 3; k++) {
                const float qmin = (float)(list[4 * e + 1 + k] & 0x7fffu), qmax = (float)((list[4 * e + 1 + k] >> 16) & 0x7fffu);
                ok &= qmin <= ghi[k] && qmax >= glo[k];
            }
            if (!ok) continue;
            int kids = 0;
            for (int j = 0; j < 4; j++) { const uint32_t* c = &nq[ref].w[4 * j]; kids += SLOT_KEPT(c); }
            if (live - 1 + kids > max_entries || n + kids > 128) continue;
            dead[e] = 1; live--;
            CUT_APPEND(ref);
        }
    }
    #undef CUT_APPEND
    #undef SLOT_KEPT
    #undef SLOT_EMPTY
    int m = 0;
    for (int e = 0; e < n; e++) if (!dead[e]) { memcpy(&out[4 * m], &list[4 * e], 16); m++; }
    return m;
}

/* The closest-hit walk of a batch of rays from a tile's cut (trace.hip, CUT mode with closest-hit scoring), counted.  A ray tests the
 * boxes of all entries against its full span (closest_t = t_max: it is generated then) and walks the subtrees of those it entered
 * one after the other, each in traverse4q's near-first order, carrying its closest hit from one to the next — `order` 1: the LAST entry
 * of the list first (the deepest: the small boxes around the tile, where a short ray's nearest hit is), 0: the first first.
 * t_out[i] = the ray's closest hit distance or t_max.  The cut's subtrees partition the tree, so the triangles a ray can reach are
 * those of the walk from the root; which of them it TESTS depends on the order, the minimum over those within reach does not — except
 * where a hit lies within rounding of a box face, which is why the product and this oracle walk the same cut in the same order. */
int orc_cut_walk_closest(const AobBvh4QNode* nq, uint64_t count4, const AobQuantFrame* fr, const AobLeafTri* tris,
                         const OrcRay* rays, uint64_t n, const uint32_t* cut, int32_t n_cut, int32_t order, OrcEntryStats* out, float* t_out)
{
    if (count4 == 0 || n_cut < 0 || n_cut > 64) return AOB_ERR_INVALID_ARGUMENT;
    for (uint64_t i = 0; i < n; i++) {
        const OrcRay* rr = &rays[i];
        if (t_out) t_out[i] = rr->o[3];
        if (rr->extra[1] == 0) continue;
        const f3 o = f3_make(rr->o[0], rr->o[1], rr->o[2]), d = f3_make(rr->d[0], rr->d[1], rr->d[2]);
        const float t_max = rr->o[3];
        Ray4q r; ray4q_prepare(&r, fr, o, d, t_max, 0);
        out->rays++;
        uint64_t entered = 0;
        float near_t[64]; int by_near[64], n_in = 0;
        for (int e = 0; e < n_cut; e++) {
            out->entries_tested++;
            if (ray4q_box(&r, &cut[4 * e + 1], t_max)) {
                out->mask_bits++; entered |= 1ull << e;
                /* (order 2, measurement only: the entry's near distance, as the slab test computes it) */
                float tn[3];
                for (int k = 0; k < 3; k++) { const uint32_t w = cut[4 * e + 1 + k]; tn[k] = fmaf(plane_float(r.down[k] ? w >> 16 : w & 0xffffu), r.fa[k], r.fb[k]); }
                near_t[e] = fmaxf(max3n(tn[0], tn[1], tn[2]), 0.0f);
                int at = n_in++;
                while (at > 0 && near_t[by_near[at - 1]] > near_t[e]) { by_near[at] = by_near[at - 1]; at--; }
                by_near[at] = e;
            }
        }
        float closest_t = t_max;
        int hit = 0;
        for (int q = 0; q < (order == 2 ? n_in : n_cut); q++) {
            const int e = order == 2 ? by_near[q] : (order ? n_cut - 1 - q : q);
            if (!((entered >> e) & 1ull)) continue;
            if (order == 3 && !ray4q_box(&r, &cut[4 * e + 1], closest_t)) continue;      /* (measurement: the entry's box again, against the hit so far) */
            if (order == 3) out->entries_tested++;
            OrcTraceStats st; memset(&st, 0, sizeof st);
            int error = 0;
            float t = closest_t;
            const uint32_t ref = cut[4 * e];
            hit |= traverse4q_from(nq, count4, fr, tris, o, d, closest_t, 0, &t, &st, &error, NULL, &ref, 1);
            if (error) return error;
            closest_t = t;
            out->inner_visits += st.inner_visits; out->leaf_visits += st.leaf_visits;
        }
        out->hits += (uint64_t)hit;
        if (t_out) t_out[i] = closest_t;
    }
    return 0;
}

/* A top-down binned-SAH BVH2 (16 bins per axis, one triangle per leaf) in the AobBvhNode layout — the tree quality a
 * treelet-restructuring pass over the PLOC tree could approach at best.  CPU only; never built on the GPU. */
typedef struct { uint32_t begin, end, index; } SahTask;
uint64_t orc_bvh_build_sah(const OrcTriangle* tris, uint64_t n, AobBvhNode* nodes, uint32_t* depth_out)
{
    if (depth_out) *depth_out = 0;
    if (n == 0) return 0;
    Box* pb = (Box*)malloc(sizeof(Box) * n);
    float (*cen)[3] = malloc(sizeof(float[3]) * n);
    uint32_t* order = (uint32_t*)malloc(4 * n);
    for (uint64_t i = 0; i < n; i++) {
        order[i] = (uint32_t)i;
        for (int k = 0; k < 3; k++) {
            pb[i].pmin[k] = fminf(fminf(tris[i].v[0][k], tris[i].v[1][k]), tris[i].v[2][k]);
            pb[i].pmax[k] = fmaxf(fmaxf(tris[i].v[0][k], tris[i].v[1][k]), tris[i].v[2][k]);
            cen[i][k] = 0.5f * (pb[i].pmin[k] + pb[i].pmax[k]);
        }
    }
    const uint64_t node_count = 2 * n - 1;
    memset(nodes, 0, sizeof(AobBvhNode) * node_count);
    if (n == 1) {
        memcpy(nodes[0].lmin_or_v0, tris[0].v[0], 12); memcpy(nodes[0].lmax_or_v1, tris[0].v[1], 12); memcpy(nodes[0].rmin_or_v2, tris[0].v[2], 12);
        nodes[0].addr_left = AOB_INVALID_ADDR; nodes[0].shape_id = tris[0].shape_id; nodes[0].prim_id = tris[0].prim_id;
        free(pb); free(cen); free(order); return 1;
    }
    /* a node over order[begin, end): inner index handed out in creation order; a range of one triangle is leaf n-1+begin */
    SahTask* todo = (SahTask*)malloc(sizeof(SahTask) * n);
    uint32_t* depth_of = (uint32_t*)calloc(n, 4);
    uint64_t top = 0; uint32_t next_inner = 1, max_depth = 0;
    todo[top].begin = 0; todo[top].end = (uint32_t)n; todo[top].index = 0; top++;
    const uint32_t leaf_base = (uint32_t)(n - 1);
    enum { BINS = 16 };
    while (top) {
        const SahTask t = todo[--top];
        const uint32_t b = t.begin, e = t.end, cnt = e - b;
        float cmin[3] = { INFINITY, INFINITY, INFINITY }, cmax[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (uint32_t i = b; i < e; i++) for (int k = 0; k < 3; k++) { cmin[k] = fminf(cmin[k], cen[order[i]][k]); cmax[k] = fmaxf(cmax[k], cen[order[i]][k]); }
        int best_axis = -1, best_split = 0; float best_cost = INFINITY;
        for (int ax = 0; ax < 3; ax++) {
            const float ext = cmax[ax] - cmin[ax];
            if (!(ext > 0.0f)) continue;
            Box bb[BINS]; uint32_t bc[BINS];
            for (int q = 0; q < BINS; q++) { bc[q] = 0; for (int k = 0; k < 3; k++) { bb[q].pmin[k] = INFINITY; bb[q].pmax[k] = -INFINITY; } }
            for (uint32_t i = b; i < e; i++) {
                int q = (int)((cen[order[i]][ax] - cmin[ax]) / ext * (float)BINS); if (q >= BINS) q = BINS - 1; if (q < 0) q = 0;
                bc[q]++; box_union(&bb[q], &bb[q], &pb[order[i]]);
            }
            float right_area[BINS]; uint32_t right_cnt[BINS];
            Box acc; for (int k = 0; k < 3; k++) { acc.pmin[k] = INFINITY; acc.pmax[k] = -INFINITY; }
            uint32_t c = 0;
            for (int q = BINS - 1; q > 0; q--) {
                c += bc[q]; if (bc[q]) box_union(&acc, &acc, &bb[q]);
                const float dx = acc.pmax[0] - acc.pmin[0], dy = acc.pmax[1] - acc.pmin[1], dz = acc.pmax[2] - acc.pmin[2];
                right_area[q] = c ? dx * dy + dy * dz + dz * dx : 0.0f; right_cnt[q] = c;
            }
            for (int k = 0; k < 3; k++) { acc.pmin[k] = INFINITY; acc.pmax[k] = -INFINITY; }
            c = 0;
            for (int q = 0; q < BINS - 1; q++) {
                c += bc[q]; if (bc[q]) box_union(&acc, &acc, &bb[q]);
                if (c == 0 || right_cnt[q + 1] == 0) continue;
                const float dx = acc.pmax[0] - acc.pmin[0], dy = acc.pmax[1] - acc.pmin[1], dz = acc.pmax[2] - acc.pmin[2];
                const float cost = (dx * dy + dy * dz + dz * dx) * (float)c + right_area[q + 1] * (float)right_cnt[q + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = ax; best_split = q + 1; }
            }
        }
        uint32_t mid;
        if (best_axis < 0) mid = b + cnt / 2;       /* all centres coincide: split the list in half */
        else {
            const float ext = cmax[best_axis] - cmin[best_axis];
            uint32_t i = b, j = e;
            while (i < j) {
                int q = (int)((cen[order[i]][best_axis] - cmin[best_axis]) / ext * (float)BINS); if (q >= BINS) q = BINS - 1; if (q < 0) q = 0;
                if (q < best_split) i++; else { j--; const uint32_t tmp = order[i]; order[i] = order[j]; order[j] = tmp; }
            }
            mid = i;
            if (mid == b || mid == e) mid = b + cnt / 2;
        }
        const uint32_t ranges[2][2] = { { b, mid }, { mid, e } };
        uint32_t child[2];
        for (int s2 = 0; s2 < 2; s2++) {
            const uint32_t cb = ranges[s2][0], ce = ranges[s2][1];
            if (ce - cb == 1) { child[s2] = leaf_base + cb; if (depth_of[t.index] + 1 > max_depth) max_depth = depth_of[t.index] + 1; }
            else {
                child[s2] = next_inner++;
                depth_of[child[s2]] = depth_of[t.index] + 1;
                todo[top].begin = cb; todo[top].end = ce; todo[top].index = child[s2]; top++;
            }
            Box cbx; for (int k = 0; k < 3; k++) { cbx.pmin[k] = INFINITY; cbx.pmax[k] = -INFINITY; }
            for (uint32_t i = cb; i < ce; i++) box_union(&cbx, &cbx, &pb[order[i]]);
            AobBvhNode* nd = &nodes[t.index];
            if (s2 == 0) { memcpy(nd->lmin_or_v0, cbx.pmin, 12); memcpy(nd->lmax_or_v1, cbx.pmax, 12); nd->addr_left = child[0]; }
            else { memcpy(nd->rmin_or_v2, cbx.pmin, 12); memcpy(nd->rmax, cbx.pmax, 12); nd->addr_right = child[1]; }
        }
    }
    for (uint64_t p = 0; p < n; p++) {
        AobBvhNode* nd = &nodes[leaf_base + p];
        const OrcTriangle* tr = &tris[order[p]];
        memcpy(nd->lmin_or_v0, tr->v[0], 12); memcpy(nd->lmax_or_v1, tr->v[1], 12); memcpy(nd->rmin_or_v2, tr->v[2], 12);
        nd->addr_left = AOB_INVALID_ADDR; nd->addr_right = 0; nd->shape_id = tr->shape_id; nd->prim_id = tr->prim_id;
    }
    if (depth_out) *depth_out = max_depth;
    free(depth_of); free(todo); free(order); free(cen); free(pb);
    return node_count;
}

/* A top-down binned-SAH BVH2 WITH SPATIAL SPLITS (Stich et al. 2009, "Spatial Splits in Bounding Volume Hierarchies" — the reference
 * stack's `bvh.sah.use_splits`, RadeonRays/radeon_rays.h:288-294) in the AobBvhNode layout: a triangle that straddles a spatial split
 * plane goes to BOTH sides as two references with clipped boxes, so the tree has more leaves than the scene has triangles (a leaf holds
 * the whole triangle inline: testing it twice changes no answer).  Measurement only (tools/measure_spatial_splits.py): what splits would
 * buy on long thin triangles.  `alpha`: a spatial split is tried where the object split's children overlap by more than alpha x the
 * root's area (1e-5 in the paper); `max_refs`: the reference budget (splits stop when it is used up; 0: no splits at all = plain SAH).
 * `nodes` holds node_capacity records (0 is returned if the tree needs more: the budget is checked before a node is split, and one split may
 * overshoot it).  Returns the node count (2 * leaves - 1), *leaves_out = references. */
typedef struct { uint32_t tri; Box box; } SRef;
typedef struct { int32_t left, right; Box lbox, rbox; uint32_t tri; } STmp;     /* left < 0: a leaf of triangle `tri` */
typedef struct {
    const OrcTriangle* tris; STmp* tmp; uint64_t tmp_count, tmp_cap; uint64_t refs_total, max_refs; float alpha_area; uint32_t max_depth;
} SBuild;

static inline float box_area(const Box* b)
{
    const float dx = b->pmax[0] - b->pmin[0], dy = b->pmax[1] - b->pmin[1], dz = b->pmax[2] - b->pmin[2];
    return dx * dy + dy * dz + dz * dx;
}
static inline void box_empty(Box* b) { for (int k = 0; k < 3; k++) { b->pmin[k] = INFINITY; b->pmax[k] = -INFINITY; } }

/* the box of triangle t clipped to lo <= x[axis] <= hi, inside `within` */
static void clip_triangle_box(const OrcTriangle* t, int axis, float lo, float hi, const Box* within, Box* out)
{
    float poly[2][10][3]; int np = 3, cur = 0;
    for (int i = 0; i < 3; i++) memcpy(poly[0][i], t->v[i], 12);
    for (int pass = 0; pass < 2; pass++) {
        const float plane = pass ? hi : lo, sgn = pass ? -1.0f : 1.0f;       /* keep sgn * (x - plane) >= 0 */
        int nn = 0;
        for (int i = 0; i < np; i++) {
            const float* a = poly[cur][i]; const float* b = poly[cur][(i + 1) % np];
            const float da = sgn * (a[axis] - plane), db = sgn * (b[axis] - plane);
            if (da >= 0.0f) memcpy(poly[cur ^ 1][nn++], a, 12);
            if ((da > 0.0f && db < 0.0f) || (da < 0.0f && db > 0.0f)) {
                const float u = da / (da - db);
                float* o = poly[cur ^ 1][nn++];
                for (int k = 0; k < 3; k++) o[k] = a[k] + u * (b[k] - a[k]);
                o[axis] = plane;
            }
        }
        np = nn; cur ^= 1;
        if (np == 0) break;
    }
    box_empty(out);
    for (int i = 0; i < np; i++) for (int k = 0; k < 3; k++) { out->pmin[k] = fminf(out->pmin[k], poly[cur][i][k]); out->pmax[k] = fmaxf(out->pmax[k], poly[cur][i][k]); }
    for (int k = 0; k < 3; k++) { out->pmin[k] = fmaxf(out->pmin[k], within->pmin[k]); out->pmax[k] = fminf(out->pmax[k], within->pmax[k]); }
}

static int32_t sbvh_node(SBuild* B, SRef* refs, uint32_t cnt, const Box* bounds, uint32_t depth, Box* box_out)
{
    enum { BINS = 16 };
    *box_out = *bounds;
    if (depth > B->max_depth) B->max_depth = depth;
    if (B->tmp_count == B->tmp_cap) { B->tmp_cap *= 2; B->tmp = (STmp*)realloc(B->tmp, sizeof(STmp) * B->tmp_cap); }
    const int32_t me = (int32_t)B->tmp_count++;
    STmp* T = &B->tmp[me];
    if (cnt == 1) { T->left = -1; T->right = -1; T->tri = refs[0].tri; *box_out = refs[0].box; return me; }
    /* ---- object split: bins over the references' box centres ---- */
    float cmin[3] = { INFINITY, INFINITY, INFINITY }, cmax[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (uint32_t i = 0; i < cnt; i++) for (int k = 0; k < 3; k++) {
        const float c = 0.5f * (refs[i].box.pmin[k] + refs[i].box.pmax[k]);
        cmin[k] = fminf(cmin[k], c); cmax[k] = fmaxf(cmax[k], c);
    }
    int o_axis = -1, o_split = 0; float o_cost = INFINITY; Box o_l, o_r; box_empty(&o_l); box_empty(&o_r);
    for (int ax = 0; ax < 3; ax++) {
        const float ext = cmax[ax] - cmin[ax];
        if (!(ext > 0.0f)) continue;
        Box bb[BINS]; uint32_t bc[BINS];
        for (int q = 0; q < BINS; q++) { bc[q] = 0; box_empty(&bb[q]); }
        for (uint32_t i = 0; i < cnt; i++) {
            const float c = 0.5f * (refs[i].box.pmin[ax] + refs[i].box.pmax[ax]);
            int q = (int)((c - cmin[ax]) / ext * (float)BINS); if (q >= BINS) q = BINS - 1; if (q < 0) q = 0;
            bc[q]++; box_union(&bb[q], &bb[q], &refs[i].box);
        }
        Box rb[BINS]; uint32_t rc[BINS]; Box acc; box_empty(&acc); uint32_t c = 0;
        for (int q = BINS - 1; q > 0; q--) { c += bc[q]; if (bc[q]) box_union(&acc, &acc, &bb[q]); rb[q] = acc; rc[q] = c; }
        box_empty(&acc); c = 0;
        for (int q = 0; q < BINS - 1; q++) {
            c += bc[q]; if (bc[q]) box_union(&acc, &acc, &bb[q]);
            if (c == 0 || rc[q + 1] == 0) continue;
            const float cost = box_area(&acc) * (float)c + box_area(&rb[q + 1]) * (float)rc[q + 1];
            if (cost < o_cost) { o_cost = cost; o_axis = ax; o_split = q + 1; o_l = acc; o_r = rb[q + 1]; }
        }
    }
    /* ---- spatial split: bins over the node's box, straddlers clipped into every bin they touch ---- */
    int s_axis = -1; float s_cost = INFINITY, s_plane = 0.0f;
    int try_spatial = B->refs_total < B->max_refs;
    if (try_spatial && o_axis >= 0) {
        Box ov;                                           /* overlap of the object split's children */
        float a = 1.0f;
        for (int k = 0; k < 3; k++) { ov.pmin[k] = fmaxf(o_l.pmin[k], o_r.pmin[k]); ov.pmax[k] = fminf(o_l.pmax[k], o_r.pmax[k]); if (!(ov.pmax[k] > ov.pmin[k])) a = 0.0f; }
        if (a == 0.0f || !(box_area(&ov) > B->alpha_area)) try_spatial = 0;
    }
    if (try_spatial) {
        for (int ax = 0; ax < 3; ax++) {
            const float lo = bounds->pmin[ax], ext = bounds->pmax[ax] - lo;
            if (!(ext > 0.0f)) continue;
            Box bb[BINS]; uint32_t en[BINS], ex[BINS];
            for (int q = 0; q < BINS; q++) { en[q] = ex[q] = 0; box_empty(&bb[q]); }
            for (uint32_t i = 0; i < cnt; i++) {
                int q0 = (int)((refs[i].box.pmin[ax] - lo) / ext * (float)BINS), q1 = (int)((refs[i].box.pmax[ax] - lo) / ext * (float)BINS);
                if (q0 < 0) q0 = 0; if (q0 >= BINS) q0 = BINS - 1; if (q1 < q0) q1 = q0; if (q1 >= BINS) q1 = BINS - 1;
                en[q0]++; ex[q1]++;
                if (q0 == q1) { box_union(&bb[q0], &bb[q0], &refs[i].box); continue; }
                for (int q = q0; q <= q1; q++) {
                    Box cb; clip_triangle_box(&B->tris[refs[i].tri], ax, lo + ext * (float)q / (float)BINS, lo + ext * (float)(q + 1) / (float)BINS, &refs[i].box, &cb);
                    if (cb.pmax[0] >= cb.pmin[0] && cb.pmax[1] >= cb.pmin[1] && cb.pmax[2] >= cb.pmin[2]) box_union(&bb[q], &bb[q], &cb);
                }
            }
            Box rb[BINS]; uint32_t rc[BINS]; Box acc; box_empty(&acc); uint32_t c = 0;
            for (int q = BINS - 1; q > 0; q--) { c += ex[q]; box_union(&acc, &acc, &bb[q]); rb[q] = acc; rc[q] = c; }
            box_empty(&acc); c = 0;
            for (int q = 0; q < BINS - 1; q++) {
                c += en[q]; box_union(&acc, &acc, &bb[q]);
                if (c == 0 || rc[q + 1] == 0 || (c == cnt && rc[q + 1] == cnt)) continue;
                const float cost = box_area(&acc) * (float)c + box_area(&rb[q + 1]) * (float)rc[q + 1];
                if (cost < s_cost) { s_cost = cost; s_axis = ax; s_plane = lo + ext * (float)(q + 1) / (float)BINS; }
            }
        }
    }
    SRef* L = NULL; SRef* R = NULL; uint32_t nl = 0, nr = 0; Box lb, rbx; box_empty(&lb); box_empty(&rbx);
    if (s_axis >= 0 && s_cost < o_cost) {
        L = (SRef*)malloc(sizeof(SRef) * cnt); R = (SRef*)malloc(sizeof(SRef) * cnt);
        for (uint32_t i = 0; i < cnt; i++) {
            const SRef* r = &refs[i];
            if (r->box.pmax[s_axis] <= s_plane) L[nl++] = *r;
            else if (r->box.pmin[s_axis] >= s_plane) R[nr++] = *r;
            else {
                Box a, b2;
                clip_triangle_box(&B->tris[r->tri], s_axis, -INFINITY, s_plane, &r->box, &a);
                clip_triangle_box(&B->tris[r->tri], s_axis, s_plane, INFINITY, &r->box, &b2);
                const int va = a.pmax[0] >= a.pmin[0] && a.pmax[1] >= a.pmin[1] && a.pmax[2] >= a.pmin[2];
                const int vb = b2.pmax[0] >= b2.pmin[0] && b2.pmax[1] >= b2.pmin[1] && b2.pmax[2] >= b2.pmin[2];
                if (va) { L[nl].tri = r->tri; L[nl].box = a; nl++; }
                if (vb) { R[nr].tri = r->tri; R[nr].box = b2; nr++; }
                if (va && vb) B->refs_total++;
                if (!va && !vb) L[nl++] = *r;
            }
        }
        if (nl == 0 || nr == 0 || (nl == cnt && nr == cnt)) { free(L); free(R); L = R = NULL; nl = nr = 0; }      /* the split separated nothing */
    }
    if (!L) {
        L = (SRef*)malloc(sizeof(SRef) * cnt); R = (SRef*)malloc(sizeof(SRef) * cnt);
        if (o_axis >= 0) {
            const float ext = cmax[o_axis] - cmin[o_axis];
            for (uint32_t i = 0; i < cnt; i++) {
                const float c = 0.5f * (refs[i].box.pmin[o_axis] + refs[i].box.pmax[o_axis]);
                int q = (int)((c - cmin[o_axis]) / ext * (float)BINS); if (q >= BINS) q = BINS - 1; if (q < 0) q = 0;
                if (q < o_split) L[nl++] = refs[i]; else R[nr++] = refs[i];
            }
        }
        if (nl == 0 || nr == 0) { nl = cnt / 2; nr = cnt - nl; memcpy(L, refs, sizeof(SRef) * nl); memcpy(R, refs + nl, sizeof(SRef) * nr); }
    }
    for (uint32_t i = 0; i < nl; i++) box_union(&lb, &lb, &L[i].box);
    for (uint32_t i = 0; i < nr; i++) box_union(&rbx, &rbx, &R[i].box);
    Box got;
    const int32_t li = sbvh_node(B, L, nl, &lb, depth + 1, &got); T = &B->tmp[me]; T->lbox = got;
    free(L);
    const int32_t ri = sbvh_node(B, R, nr, &rbx, depth + 1, &got); T = &B->tmp[me]; T->rbox = got;
    free(R);
    T->left = li; T->right = ri; T->tri = 0;
    box_union(box_out, &T->lbox, &T->rbox);
    return me;
}

uint64_t orc_bvh_build_sbvh(const OrcTriangle* tris, uint64_t n, AobBvhNode* nodes, uint64_t node_capacity, uint64_t max_refs, float alpha, uint32_t* depth_out, uint64_t* leaves_out)
{
    if (depth_out) *depth_out = 0;
    if (leaves_out) *leaves_out = 0;
    if (n == 0) return 0;
    SBuild B; memset(&B, 0, sizeof B);
    B.tris = tris; B.refs_total = n; B.max_refs = max_refs > n ? max_refs : n;
    const uint64_t cap = 2 * B.max_refs + 16;
    B.tmp = (STmp*)malloc(sizeof(STmp) * cap); B.tmp_cap = cap;
    SRef* refs = (SRef*)malloc(sizeof(SRef) * n);
    Box root; box_empty(&root);
    for (uint64_t i = 0; i < n; i++) {
        refs[i].tri = (uint32_t)i;
        for (int k = 0; k < 3; k++) {
            refs[i].box.pmin[k] = fminf(fminf(tris[i].v[0][k], tris[i].v[1][k]), tris[i].v[2][k]);
            refs[i].box.pmax[k] = fmaxf(fmaxf(tris[i].v[0][k], tris[i].v[1][k]), tris[i].v[2][k]);
        }
        box_union(&root, &root, &refs[i].box);
    }
    B.alpha_area = alpha * box_area(&root);
    Box got;
    sbvh_node(&B, refs, (uint32_t)n, &root, 0, &got);
    free(refs);
    /* emit: inner nodes in creation order (root = 0), leaves in depth-first order */
    const uint64_t total = B.tmp_count, leaves = (total + 1) / 2, leaf_base = leaves - 1;
    if (total > node_capacity) { free(B.tmp); return 0; }
    uint32_t* index = (uint32_t*)malloc(4 * total);
    uint64_t ni = 0, nl2 = 0;
    /* creation order is depth-first preorder: a leaf's rank among the leaves in that order is its depth-first position */
    for (uint64_t i = 0; i < total; i++) index[i] = B.tmp[i].left < 0 ? (uint32_t)(leaf_base + nl2++) : (uint32_t)ni++;
    memset(nodes, 0, sizeof(AobBvhNode) * total);
    for (uint64_t i = 0; i < total; i++) {
        const STmp* t = &B.tmp[i];
        AobBvhNode* nd = &nodes[index[i]];
        if (t->left < 0) {
            const OrcTriangle* tr = &tris[t->tri];
            memcpy(nd->lmin_or_v0, tr->v[0], 12); memcpy(nd->lmax_or_v1, tr->v[1], 12); memcpy(nd->rmin_or_v2, tr->v[2], 12);
            nd->addr_left = AOB_INVALID_ADDR; nd->addr_right = 0; nd->shape_id = tr->shape_id; nd->prim_id = tr->prim_id;
        } else {
            memcpy(nd->lmin_or_v0, t->lbox.pmin, 12); memcpy(nd->lmax_or_v1, t->lbox.pmax, 12); nd->addr_left = index[t->left];
            memcpy(nd->rmin_or_v2, t->rbox.pmin, 12); memcpy(nd->rmax, t->rbox.pmax, 12); nd->addr_right = index[t->right];
        }
    }
    free(index); free(B.tmp);
    if (depth_out) *depth_out = B.max_depth;
    if (leaves_out) *leaves_out = leaves;
    return total;
}

/* ------------------------------------------------------------------------ */
/* resolve + bleed + format conversion                                       */

static inline int32_t as_i32(uint32_t u) { return (int32_t)u; }
static inline float as_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline int cell_covered(const AobResult* r, int quantised)
{
    /* PostprocessPixelShader.hlsl:43 (== 0) and :66 (> 0) */
    return quantised ? as_i32(r->weight_bits) != 0 : as_f32(r->weight_bits) != 0.0f;
}
static inline int cell_positive(const AobResult* r, int quantised)
{
    return quantised ? as_i32(r->weight_bits) > 0 : as_f32(r->weight_bits) > 0.0f;
}
/* PostprocessPixelShader.hlsl:31-35,81: ((T*255)/Wt)/255.0f with integer division */
static inline float cell_value(const AobResult* r, int quantised)
{
    if (quantised) {
        int32_t T = as_i32(r->total_bits), Wt = as_i32(r->weight_bits);
        /* 32-bit wrap-around multiply as in HLSL; division truncates toward zero */
        int32_t num = (int32_t)((uint32_t)T * 255u);
        int32_t q = (Wt == -1 && num == INT32_MIN) ? INT32_MIN : num / Wt;
        return (float)q / 255.0f;
    }
    return as_f32(r->total_bits) / as_f32(r->weight_bits);
}

static inline uint8_t to_unorm8(float v)
{
    /* D3D11 float -> UNORM: NaN -> 0, clamp to [0,1], scale, add 0.5, truncate */
    if (!(v == v)) return 0;
    float c = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    return (uint8_t)(c * 255.0f + 0.5f);
}

void orc_resolve(const AobResult* results, int32_t W, int32_t H, const AobOptions* opt,
                 int32_t channels, int32_t bpc, uint8_t* out)
{
    const int quantised = opt->accumulate == AOB_ACCUMULATE_QUANTISED;
    const int32_t B = opt->bleed_distance;
    const int64_t cells = (int64_t)W * H;
    const int f16mode = opt->normal_fp16 == AOB_FP16_RTZ ? AOB_FP16_RTZ : AOB_FP16_RNE;
    for (int32_t py = 0; py < H; py++) {
        for (int32_t px = 0; px < W; px++) {
            const AobResult* me = &results[(size_t)py * W + px];
            float result = 0.0f;
            if (!cell_covered(me, quantised) && opt->bleed_mode == AOB_BLEED_NEAREST) {
                /* the nearest covered texel within B (inclusive), inside the texture only; ties by scan order */
                int32_t closest = B * B + 1;
                for (int32_t y = py - B < 0 ? 0 : py - B; y <= (py + B > H - 1 ? H - 1 : py + B); y++) {
                    for (int32_t x = px - B < 0 ? 0 : px - B; x <= (px + B > W - 1 ? W - 1 : px + B); x++) {
                        const int32_t d2 = (x - px) * (x - px) + (y - py) * (y - py);
                        if (d2 >= closest) continue;
                        const AobResult* nb = &results[(size_t)y * W + x];
                        if (cell_positive(nb, quantised)) { closest = d2; result = cell_value(nb, quantised); }
                    }
                }
            } else if (!cell_covered(me, quantised)) {
                /* :43-78: scan x outer / y inner; the read uses the RAW (x, y), so a negative x
                 * aliases into the previous row and an index outside the buffer reads zeros */
                int32_t closest = (B + 1) * (B + 1);
                for (int32_t x = px - B; x <= px + B; x++) {
                    for (int32_t y = py - B; y <= py + B; y++) {
                        int64_t idx = (int64_t)y * W + x;
                        if (idx < 0 || idx >= cells) continue;
                        const AobResult* nb = &results[idx];
                        if (cell_positive(nb, quantised)) {
                            int32_t d2 = (x - px) * (x - px) + (y - py) * (y - py);
                            if (d2 < closest && d2 < B * B) {
                                closest = d2;
                                result = cell_value(nb, quantised);
                            }
                        }
                    }
                }
            } else {
                result = cell_value(me, quantised);
            }
            /* :85 float4(result, 0, 0, 1) into the AO texture's format (BakeEngine.cpp:482) */
            const float px4[4] = { result, 0.0f, 0.0f, 1.0f };
            uint8_t* dst = out + ((size_t)py * W + px) * (size_t)channels * (size_t)bpc;
            for (int c = 0; c < channels; c++) {
                float v = px4[c < 4 ? c : 3];
                if (bpc == 1) dst[c] = to_unorm8(v);
                else if (bpc == 2) { uint16_t h = orc_f32_to_f16(v, f16mode); memcpy(dst + 2 * c, &h, 2); }
                else memcpy(dst + 4 * c, &v, 4);
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* end to end                                                                */

/* AobOptions::bvh_builder -1 (include/aobaker.h aob_resolve_bvh_builder): PLOC under the binned-SAH top for an any-hit bake of at least
 * 2^28 rays (texels x samples) over more than 131 072 triangles, else PLOC. */
int32_t orc_resolve_bvh_builder(const AobOptions* opt, uint64_t texels, uint64_t triangles)
{
    if (opt->bvh_builder >= 0) return opt->bvh_builder;
    const uint64_t rays = texels * (uint64_t)(opt->sample_count > 0 ? opt->sample_count : 0);
    return (opt->use_any_hit && rays >= (1ull << 28) && triangles > 131072ull) ? 2 : 1;
}

int orc_bake(const AobSceneDesc* scene, const AobOptions* opt_in, int32_t threads, OrcTraceStats* stats)
{
    if (!scene || !opt_in || opt_in->sample_count <= 0) return AOB_ERR_INVALID_ARGUMENT;
    uint64_t ntri = orc_scene_triangle_count(scene);
    AobOptions opt_v = *opt_in;
    {
        uint64_t texels = 0;
        for (uint32_t mi = 0; mi < scene->material_count; mi++)
            if (scene->materials[mi].width > 0 && scene->materials[mi].height > 0) texels += (uint64_t)scene->materials[mi].width * (uint64_t)scene->materials[mi].height;
        opt_v.bvh_builder = orc_resolve_bvh_builder(opt_in, texels, ntri);
    }
    const AobOptions* opt = &opt_v;
    OrcTriangle* tris = (OrcTriangle*)malloc(sizeof(OrcTriangle) * (ntri ? ntri : 1));
    orc_scene_triangles(scene, tris);
    AobBvhNode* nodes = (AobBvhNode*)malloc(sizeof(AobBvhNode) * (ntri ? 2 * ntri : 1));
    uint32_t depth = 0;
    uint64_t node_count = orc_bvh_build_ex(tris, ntri, nodes, &depth, opt->bvh_builder, opt->morton_adaptive);
    float* hemi = NULL;
    if (opt->sampler == AOB_SAMPLER_REFERENCE_COMPAT) {
        uint64_t n = orc_compat_table_entries(opt);
        hemi = (float*)malloc(sizeof(float) * 3 * (n ? n : 1));
        orc_hemisphere_table(hemi, n);
    }
    int rc = 0;
    for (uint32_t mi = 0; mi < scene->material_count && rc == 0; mi++) {
        const AobMaterialDesc* mat = &scene->materials[mi];
        if (mat->width <= 0 || mat->height <= 0) continue;
        size_t cells = (size_t)mat->width * mat->height;
        uint64_t cnt = orc_rasterize(scene, (int32_t)mi, opt, NULL, NULL, 0);
        AobSurfel* surfels = (AobSurfel*)malloc(sizeof(AobSurfel) * (cnt ? cnt : 1));
        uint32_t* sidx = (uint32_t*)malloc(sizeof(uint32_t) * (cnt ? cnt : 1));
        uint32_t* smesh = (uint32_t*)malloc(sizeof(uint32_t) * (cnt ? cnt : 1));
        orc_rasterize_ex(scene, (int32_t)mi, opt, surfels, sidx, smesh, cnt);
        AobResult* results = (AobResult*)calloc(cells, sizeof(AobResult));
        uint32_t* flags = (uint32_t*)malloc(sizeof(uint32_t) * (scene->mesh_count ? scene->mesh_count : 1));
        int any_flags = 0;
        for (uint32_t k = 0; k < scene->mesh_count; k++) { flags[k] = scene->meshes[k].flags; any_flags |= flags[k] != 0; }
        rc = orc_trace_ex(nodes, node_count, surfels, sidx, any_flags ? smesh : NULL, any_flags ? flags : NULL, scene->mesh_count,
                          mat->ao_max_range, 0, cnt, opt, hemi, mat->width, mat->height, results, threads, stats);
        free(flags); free(smesh);
        if (rc == 0 && mat->data) orc_resolve(results, mat->width, mat->height, opt, mat->channels, mat->bytes_per_channel, mat->data);
        free(results); free(sidx); free(surfels);
    }
    free(hemi); free(nodes); free(tris);
    return rc;
}

// bvh_build.hip — world-space triangles + LBVH build on the GPU (gfx950).
//
// Replaces what RadeonRays does inside CreateMesh/SetTransform/AttachShape/
// Commit (reference BakeEngine.cpp:370-400).  The algorithm is the one of the
// reference stack's own GPU builder, kernels/CL/build_hlbvh.cl (Morton codes
// :69-125, Karras span/split with index tie-break :131-261, bottom-up refit
// with atomic flags :264-314), with 63-bit instead of 30-bit codes, emitted
// into the 64-byte node layout of kernels/CL/intersect_bvh2_lds.cl:40-52
// (inner: two child boxes + child addresses; leaf: the triangle, inlined).
//
// All kernels are HBM-streaming: one thread per primitive / node, 16-byte
// coalesced loads and stores where the record allows it.
#include "aob_internal.h"
#include "aob_math.h"

#include <cstring>
#include <rocprim/rocprim.hpp>

namespace aob {

static constexpr int kBlock = 256;
static inline uint32_t blocks_for(uint64_t n) { return (uint32_t)((n + kBlock - 1) / kBlock); }

// ---- order-preserving float <-> uint encoding for atomicMin/Max on bounds ----
__device__ __forceinline__ uint32_t enc_float(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_float(uint32_t e)
{
    uint32_t u = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
    return __uint_as_float(u);
}

__global__ void k_init_scene_bounds(uint32_t* enc)
{
    if (threadIdx.x < 3) enc[threadIdx.x] = 0xffffffffu;        // min slots
    else if (threadIdx.x < 6) enc[threadIdx.x] = 0u;            // max slots
}

struct MeshXform { float m[16]; };

// RasterizeVertexShader.hlsl:32 worldpos = mul(float4(pos,1), model), row vector.
__device__ __forceinline__ f3 transform_point(const MeshXform& x, const float* p)
{
    f3 r;
    float t;
    t = p[0] * x.m[0]; t = fmaf(p[1], x.m[4], t); t = fmaf(p[2], x.m[8], t);  r.x = t + x.m[12];
    t = p[0] * x.m[1]; t = fmaf(p[1], x.m[5], t); t = fmaf(p[2], x.m[9], t);  r.y = t + x.m[13];
    t = p[0] * x.m[2]; t = fmaf(p[1], x.m[6], t); t = fmaf(p[2], x.m[10], t); r.z = t + x.m[14];
    return r;
}

// One thread per triangle of one mesh: world-space vertices (what RadeonRays holds
// after SetTransform), primitive box, and a block-reduced scene box.
__global__ __launch_bounds__(kBlock)
void k_world_triangles(const float* __restrict__ positions, const uint32_t* __restrict__ indices,
                       uint32_t ntri, MeshXform xf, uint32_t shape_id,
                       float4* __restrict__ tri_verts, Box* __restrict__ prim_bounds,
                       uint32_t* __restrict__ scene_enc)
{
    __shared__ float red[6][kBlock / 64];
    uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    if (t < ntri) {
        f3 v[3];
#pragma unroll
        for (int c = 0; c < 3; c++) v[c] = transform_point(xf, positions + 3ull * indices[3ull * t + c]);
        tri_verts[3ull * t + 0] = make_float4(v[0].x, v[0].y, v[0].z, __uint_as_float(shape_id));
        tri_verts[3ull * t + 1] = make_float4(v[1].x, v[1].y, v[1].z, __uint_as_float(t));
        tri_verts[3ull * t + 2] = make_float4(v[2].x, v[2].y, v[2].z, 0.0f);
        mn[0] = fminf(fminf(v[0].x, v[1].x), v[2].x); mx[0] = fmaxf(fmaxf(v[0].x, v[1].x), v[2].x);
        mn[1] = fminf(fminf(v[0].y, v[1].y), v[2].y); mx[1] = fmaxf(fmaxf(v[0].y, v[1].y), v[2].y);
        mn[2] = fminf(fminf(v[0].z, v[1].z), v[2].z); mx[2] = fmaxf(fmaxf(v[0].z, v[1].z), v[2].z);
        Box b;
#pragma unroll
        for (int k = 0; k < 3; k++) { b.pmin[k] = mn[k]; b.pmax[k] = mx[k]; }
        prim_bounds[t] = b;
    }
    // wave reduce, then one set of atomics per block
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], off));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off));
        }
    }
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { red[k][wave] = mn[k]; red[3 + k][wave] = mx[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        int k = threadIdx.x;
        float v = red[k][0];
        for (int w = 1; w < kBlock / 64; w++) v = k < 3 ? fminf(v, red[k][w]) : fmaxf(v, red[k][w]);
        if (k < 3) atomicMin(&scene_enc[k], enc_float(v));
        else atomicMax(&scene_enc[k], enc_float(v));
    }
}

__device__ __forceinline__ uint64_t expand21(uint32_t v)
{
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__device__ __forceinline__ uint32_t morton_axis(float c, float smin, float extent)
{
    // build_hlbvh.cl:122 then :71-73, at 21 bits per axis
    float n = extent > 0.0f ? (c - smin) / extent : 0.0f;
    float q = fminf(fmaxf(n * 2097152.0f, 0.0f), 2097151.0f);
    return (uint32_t)q;
}

// ADAPTIVE: the 63 bits are dealt to the axes in the order a spatial-median builder would split them — always the axis
// whose cell is currently longest, halving that cell (the rule of the reference library's default builder,
// radeon_rays.h:283-287, applied to the scene box) — instead of strictly x, y, z in turn.  On a cube-shaped scene the
// two orders coincide; on a flat one (a terrain: 1 x 0.15 x 1) the strict order spends every third split on a plane
// that separates nothing: 8 % fewer node visits per ray on the 1 M-triangle terrain, 3 - 15 % on the instanced rocks.
template <int ADAPTIVE>
__global__ __launch_bounds__(kBlock)
void k_morton(const Box* __restrict__ prim_bounds, uint32_t n, const uint32_t* __restrict__ scene_enc,
              uint64_t* __restrict__ codes, uint32_t* __restrict__ idx)
{
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Box b = prim_bounds[i];
    uint32_t q[3];
    float ext[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float smin = dec_float(scene_enc[k]), smax = dec_float(scene_enc[3 + k]);
        float c = (b.pmax[k] + b.pmin[k]) * 0.5f;                 // build_hlbvh.cl:118
        ext[k] = smax - smin;
        q[k] = morton_axis(c, smin, ext[k]);
    }
    if (ADAPTIVE) {
        uint64_t code = 0;
        int used0 = 0, used1 = 0, used2 = 0;                      // bits of q[k] consumed, from the top
        float e0 = ext[0], e1 = ext[1], e2 = ext[2];
        for (int bit = 0; bit < 63; bit++) {
            // the longest cell among the axes with bits left; ties go to the lower axis
            int ax = -1; float best = -1.0f;
            if (used0 < 21 && e0 > best) { best = e0; ax = 0; }
            if (used1 < 21 && e1 > best) { best = e1; ax = 1; }
            if (used2 < 21 && e2 > best) { best = e2; ax = 2; }
            if (ax == 0) { code = code << 1 | ((q[0] >> (20 - used0)) & 1u); used0++; e0 *= 0.5f; }
            else if (ax == 1) { code = code << 1 | ((q[1] >> (20 - used1)) & 1u); used1++; e1 *= 0.5f; }
            else { code = code << 1 | ((q[2] >> (20 - used2)) & 1u); used2++; e2 *= 0.5f; }
        }
        codes[i] = code;
    } else {
        codes[i] = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
    }
    idx[i] = i;
}

// build_hlbvh.cl:131-150 with 64-bit codes
__device__ __forceinline__ int lbvh_delta(const uint64_t* __restrict__ codes, int64_t n, int64_t i1, int64_t i2)
{
    int64_t left = i1 < i2 ? i1 : i2, right = i1 < i2 ? i2 : i1;
    if (left < 0 || right >= n) return -1;
    uint64_t lc = codes[left], rc = codes[right];
    if (lc != rc) return __clzll((long long)(lc ^ rc));
    return 64 + __clz((int)((uint32_t)left ^ (uint32_t)right));
}

// build_hlbvh.cl:152-261: one thread per internal node
__global__ __launch_bounds__(kBlock)
void k_hierarchy(const uint64_t* __restrict__ codes, uint32_t n,
                 uint32_t* __restrict__ child_left, uint32_t* __restrict__ child_right,
                 uint32_t* __restrict__ parent)
{
    int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int64_t N = n;
    if (idx >= N - 1) return;
    if (idx == 0) parent[0] = 0xffffffffu;
    int dnext = lbvh_delta(codes, N, idx, idx + 1), dprev = lbvh_delta(codes, N, idx, idx - 1);
    int d = dnext >= dprev ? 1 : -1;
    int delta_min = lbvh_delta(codes, N, idx, idx - d);
    int64_t lmax = 2;
    while (lbvh_delta(codes, N, idx, idx + lmax * d) > delta_min) lmax *= 2;
    int64_t l = 0, t = lmax;
    do {
        t /= 2;
        if (lbvh_delta(codes, N, idx, idx + (l + t) * d) > delta_min) l += t;
    } while (t > 1);
    int64_t other = idx + l * d;
    int64_t sx = idx < other ? idx : other, sy = idx < other ? other : idx;
    int64_t lo = sx, hi = sy;
    int num_identical = lbvh_delta(codes, N, lo, hi);
    do {
        int64_t ns = (hi + lo) / 2;
        if (lbvh_delta(codes, N, lo, ns) > num_identical) lo = ns; else hi = ns;
    } while (hi > lo + 1);
    int64_t split = lo;
    uint32_t c1 = (split == sx) ? (uint32_t)(N - 1 + split) : (uint32_t)split;
    uint32_t c2 = (split + 1 == sy) ? (uint32_t)(N - 1 + split + 1) : (uint32_t)(split + 1);
    child_left[idx] = c1; child_right[idx] = c2;
    parent[c1] = (uint32_t)idx; parent[c2] = (uint32_t)idx;
}

__global__ __launch_bounds__(kBlock)
void k_leaf_bounds(const Box* __restrict__ prim_bounds, const uint32_t* __restrict__ sorted_idx, uint32_t n,
                   Box* __restrict__ node_bounds)
{
    uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p < n) node_bounds[(uint64_t)n - 1 + p] = prim_bounds[sorted_idx[p]];
}

// build_hlbvh.cl:264-314.  The second thread to arrive at a node reads both child boxes, one of
// which another CU wrote in this same launch.  Hand-off without fences, in the form
// MI355X_MICROARCH.md (inter-workgroup visibility, "valid forms") measures as safe: every store
// of a box is an agent-scope (sc1, write-through) store, the storing lane drains them with
// s_waitcnt vmcnt(0) BEFORE its atomic add on the parent's flag, and the reader loads the boxes
// with agent-scope (sc1, L1-bypassing) loads only after its own add has returned.  (The previous
// version bracketed the add with two __threadfence() per level: 6.3 ms for 1 M triangles.)
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(kBlock)
void k_refit(uint32_t n, const uint32_t* __restrict__ parent, const uint32_t* __restrict__ child_left,
             const uint32_t* __restrict__ child_right, Box* node_bounds, uint32_t* flags,
             uint32_t* __restrict__ depth_max)
{
    uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    uint32_t depth = 0;
    if (p < n) {
        uint32_t idx = n - 1 + p;
        // depth of this leaf (root = 0)
        for (uint32_t a = idx; parent[a] != 0xffffffffu; a = parent[a]) depth++;
        do {
            idx = parent[idx];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // my box stores have left the CU
            if (__hip_atomic_fetch_add(&flags[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) break;
            const float* bl = node_bounds[child_left[idx]].pmin;     // Box = pmin[3], pmax[3]: six floats
            const float* br = node_bounds[child_right[idx]].pmin;
            float* out = node_bounds[idx].pmin;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                st_agent(out + k, fminf(ld_agent(bl + k), ld_agent(br + k)));
                st_agent(out + 3 + k, fmaxf(ld_agent(bl + 3 + k), ld_agent(br + 3 + k)));
            }
        } while (idx != 0);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) depth = max(depth, (uint32_t)__shfl_xor((int)depth, off));
    if ((threadIdx.x & 63) == 0 && depth) atomicMax(depth_max, depth);
}

// One thread per node: inner nodes [0, n-1), leaves [n-1, 2n-1).  64-byte records,
// written as four float4 stores.
__global__ __launch_bounds__(kBlock)
void k_emit_nodes(uint32_t n, const uint32_t* __restrict__ child_left, const uint32_t* __restrict__ child_right,
                  const Box* __restrict__ node_bounds, const uint32_t* __restrict__ sorted_idx,
                  const float4* __restrict__ tri_verts, float4* __restrict__ nodes)
{
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    uint32_t total = n == 1 ? 1 : 2 * n - 1;
    if (i >= total) return;
    float4* out = nodes + 4ull * i;
    if (n > 1 && i < n - 1) {
        uint32_t l = child_left[i], r = child_right[i];
        Box bl = node_bounds[l], br = node_bounds[r];
        out[0] = make_float4(bl.pmin[0], bl.pmin[1], bl.pmin[2], __uint_as_float(l));
        out[1] = make_float4(bl.pmax[0], bl.pmax[1], bl.pmax[2], __uint_as_float(0u));
        out[2] = make_float4(br.pmin[0], br.pmin[1], br.pmin[2], __uint_as_float(r));
        out[3] = make_float4(br.pmax[0], br.pmax[1], br.pmax[2], __uint_as_float(0u));
    } else {
        uint32_t p = n == 1 ? 0 : i - (n - 1);
        uint32_t t = sorted_idx[p];
        float4 v0 = tri_verts[3ull * t], v1 = tri_verts[3ull * t + 1], v2 = tri_verts[3ull * t + 2];
        out[0] = make_float4(v0.x, v0.y, v0.z, __uint_as_float(0xffffffffu));   // addr_left = INVALID
        out[1] = make_float4(v1.x, v1.y, v1.z, v0.w);                            // shape id
        out[2] = make_float4(v2.x, v2.y, v2.z, __uint_as_float(0u));
        out[3] = make_float4(0.0f, 0.0f, 0.0f, v1.w);                            // prim id
    }
}

// ---- 4-ary collapse -------------------------------------------------------------------------
// Every inner node at EVEN depth becomes a 4-ary node whose children are its grandchildren (or
// a child itself where that child is a leaf); odd levels disappear.  The child boxes are the
// BVH2 boxes, unchanged, so a ray tests a subset of the boxes the BVH2 walk would test.
// Layout (AobBvh4Node): lo[j] = {min, ref_j}, hi[j] = {max, 0}; the quad of lanes that traces a
// ray fetches lo[0..3] with one 64-byte access and hi[0..3] with another.
// PAIRS (the lane-per-ray quantised walk): a BVH2 node whose children are the leaves at Morton positions p and p + 1
// is not a node of its own but ONE child slot, ref = AOB_LEAF_FLAG | AOB_PAIR_FLAG | p — a leaf of two triangles that
// lie next to each other in AobLeafTri[].  96 % of the terrain's triangles, 87 % of the sphere's and 82 % of the
// rocks' come in such pairs (the two halves of a grid cell): 39 % fewer leaf visits and 3 % fewer node visits per ray
// for 4 % more triangle tests.
__device__ __forceinline__ bool is_leaf_pair(uint32_t c, uint32_t leaf_base, const uint32_t* __restrict__ child_left,
                                             const uint32_t* __restrict__ child_right)
{
    if (c >= leaf_base) return false;
    const uint32_t l = child_left[c], r = child_right[c];
    return l >= leaf_base && r == l + 1u;
}

template <int PAIRS>
__global__ __launch_bounds__(kBlock)
void k_collapse4(uint32_t n, const uint32_t* __restrict__ parent, const uint32_t* __restrict__ child_left,
                 const uint32_t* __restrict__ child_right, const Box* __restrict__ node_bounds,
                 float4* __restrict__ nodes4)
{
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n - 1) return;
    uint32_t depth = 0;
    for (uint32_t a = i; parent[a] != 0xffffffffu; a = parent[a]) depth++;
    if (depth & 1u) return;
    const uint32_t leaf_base = n - 1;
    if (PAIRS && i != 0u && is_leaf_pair(i, leaf_base, child_left, child_right)) return;     // a child slot of its grandparent
    uint32_t kids[4]; int k = 0;
    const uint32_t c2[2] = { child_left[i], child_right[i] };
#pragma unroll
    for (int s = 0; s < 2; s++) {
        uint32_t c = c2[s];
        if (c < leaf_base && !(PAIRS && is_leaf_pair(c, leaf_base, child_left, child_right))) { kids[k++] = child_left[c]; kids[k++] = child_right[c]; }
        else kids[k++] = c;
    }
    float4* out = nodes4 + 8ull * i;
    for (int j = 0; j < 4; j++) {
        if (j < k) {
            const uint32_t c = kids[j];
            const Box b = node_bounds[c];
            uint32_t ref = c < leaf_base ? c : (0x80000000u | (c - leaf_base));
            if (PAIRS && is_leaf_pair(c, leaf_base, child_left, child_right)) ref = 0xC0000000u | (child_left[c] - leaf_base);
            out[j] = make_float4(b.pmin[0], b.pmin[1], b.pmin[2], __uint_as_float(ref));
            out[4 + j] = make_float4(b.pmax[0], b.pmax[1], b.pmax[2], 0.0f);
        } else {
            out[j] = make_float4(INFINITY, INFINITY, INFINITY, __uint_as_float(0xffffffffu));
            out[4 + j] = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
        }
    }
}

// ---- quantised nodes ---------------------------------------------------------------------------
// Child boxes live on a 15-bit grid over the scene box (cell = extent / 32767; on the 1 M-triangle terrain
// 0.3 % more node visits than a 16-bit grid), rounded outward.  15 bits because the 4-ary nodes store a
// plane as 0x8000 | q: the traversal turns those two bytes into the float 1 + q/32768 with ONE v_perm_b32.
// frame: origin = scene min, scale = extent / 32767 nudged up until origin + 32767*scale >= scene max
static constexpr float kQuantMax = 32767.0f;
__global__ void k_quant_frame(const uint32_t* __restrict__ scene_enc, float* __restrict__ frame)
{
    const int k = threadIdx.x;
    if (k >= 3) return;
    const float smin = dec_float(scene_enc[k]), smax = dec_float(scene_enc[3 + k]);
    const float extent = smax - smin;
    float scale = extent > 0.0f ? extent / kQuantMax : 0.0f;
    if (scale > 0.0f) while (fmaf(kQuantMax, scale, smin) < smax) scale = __uint_as_float(__float_as_uint(scale) + 1u);
    frame[k] = smin; frame[3 + k] = scale;
}

// largest q with origin + q*scale <= v  /  smallest q with origin + q*scale >= v
__device__ __forceinline__ uint32_t quant_down(float v, float origin, float scale)
{
    if (!(scale > 0.0f)) return 0u;
    float q = fminf(fmaxf(floorf((v - origin) / scale), 0.0f), kQuantMax);
    while (q > 0.0f && fmaf(q, scale, origin) > v) q -= 1.0f;
    return (uint32_t)q;
}
__device__ __forceinline__ uint32_t quant_up(float v, float origin, float scale)
{
    if (!(scale > 0.0f)) return 0u;
    float q = fminf(fmaxf(ceilf((v - origin) / scale), 0.0f), kQuantMax);
    while (q < kQuantMax && fmaf(q, scale, origin) < v) q += 1.0f;
    return (uint32_t)q;
}

__global__ __launch_bounds__(kBlock)
void k_emit_nodes_q(uint32_t n, const uint32_t* __restrict__ child_left, const uint32_t* __restrict__ child_right,
                    const Box* __restrict__ node_bounds, const float* __restrict__ frame, uint4* __restrict__ nodesq)
{
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n - 1) return;
    const uint32_t l = child_left[i], r = child_right[i];
    const Box bl = node_bounds[l], br = node_bounds[r];
    uint32_t q[12];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float o = frame[k], sc = frame[3 + k];
        q[k] = quant_down(bl.pmin[k], o, sc); q[3 + k] = quant_up(bl.pmax[k], o, sc);
        q[6 + k] = quant_down(br.pmin[k], o, sc); q[9 + k] = quant_up(br.pmax[k], o, sc);
    }
    // one word per axis and box: min | max << 16 (the trace kernel rotates the word by 16 for rays that
    // travel down that axis, so that the low half is always the near plane)
    nodesq[2ull * i + 0] = make_uint4(q[0] | q[3] << 16, q[1] | q[4] << 16, q[2] | q[5] << 16, q[6] | q[9] << 16);
    nodesq[2ull * i + 1] = make_uint4(q[7] | q[10] << 16, q[8] | q[11] << 16, l, r);
}

// The 4-ary nodes sit at the index of the BVH2 node they were made from, so every second record is a hole (a
// collapsed-away level).  The quantised array the traversal walks is packed: live records in index order — for both
// builders that is the tree top-down, neighbours in space next to each other — and refs renumbered.  Half the
// bytes, and both halves of every 128-byte cache line are nodes.
__global__ __launch_bounds__(kBlock)
void k_live4(uint32_t count4, const float4* __restrict__ nodes4, uint32_t* __restrict__ live)
{
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= count4) return;
    live[i] = __float_as_uint(nodes4[8ull * i].w) != 0u ? 1u : 0u;     // slot 0 of a node is never empty, and never node 0
}

// quantised form of the 4-ary nodes: child j = {x, y, z words ((0x8000 | min) | (0x8000 | max) << 16), ref}
__global__ __launch_bounds__(kBlock)
void k_emit_nodes4q(uint32_t count4, const float4* __restrict__ nodes4, const float* __restrict__ frame,
                    const uint32_t* __restrict__ live, const uint32_t* __restrict__ packed_index, uint4* __restrict__ nodes4q,
                    uint32_t* __restrict__ packed_count)
{
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= count4) return;
    if (i == count4 - 1) *packed_count = packed_index[i] + live[i];
    if (!live[i]) return;
    const uint64_t at = packed_index[i];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float4 lo = nodes4[8ull * i + j], hi = nodes4[8ull * i + 4 + j];
        uint32_t ref = __float_as_uint(lo.w);
        // an empty slot is an INVERTED box — min plane 0x7fff, max plane 0 on every axis: its near distance exceeds its far
        // distance for every ray, so the walk needs no test of the ref (trace.hip).  Where arithmetic cannot tell the two
        // planes apart (a scene without extent; an extent below the rounding of the ray's distance to it) the slot may pass
        // all the same, so its ref is one that is harmless to follow: leaf triangle 0, a triangle of the scene — testing it
        // once more changes no ray's answer.
        uint4 o = make_uint4(0x8000ffffu, 0x8000ffffu, 0x8000ffffu, 0x80000000u);
        if (ref != 0xffffffffu) {
            o.w = ref;
            if (ref < 0x80000000u) o.w = packed_index[ref];
            o.x = 0x80008000u | quant_down(lo.x, frame[0], frame[3]) | quant_up(hi.x, frame[0], frame[3]) << 16;
            o.y = 0x80008000u | quant_down(lo.y, frame[1], frame[4]) | quant_up(hi.y, frame[1], frame[4]) << 16;
            o.z = 0x80008000u | quant_down(lo.z, frame[2], frame[5]) | quant_up(hi.z, frame[2], frame[5]) << 16;
        }
        nodes4q[4ull * at + j] = o;
    }
}

// n == 1: the tree is a single leaf; give it a 4-ary root with that one child
__global__ void k_single_leaf4(const Box* __restrict__ prim_bounds, float4* __restrict__ nodes4)
{
    if (threadIdx.x != 0) return;
    const Box b = prim_bounds[0];
    nodes4[0] = make_float4(b.pmin[0], b.pmin[1], b.pmin[2], __uint_as_float(0x80000000u));
    nodes4[4] = make_float4(b.pmax[0], b.pmax[1], b.pmax[2], 0.0f);
    for (int j = 1; j < 4; j++) {
        nodes4[j] = make_float4(INFINITY, INFINITY, INFINITY, __uint_as_float(0xffffffffu));
        nodes4[4 + j] = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
    }
}

__global__ __launch_bounds__(kBlock)
void k_leaf_tris(uint32_t n, const uint32_t* __restrict__ sorted_idx, const float4* __restrict__ tri_verts,
                 float4* __restrict__ leaf_tris)
{
    uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const uint32_t t = sorted_idx[p];
    leaf_tris[3ull * p + 0] = tri_verts[3ull * t + 0];
    leaf_tris[3ull * p + 1] = tri_verts[3ull * t + 1];
    leaf_tris[3ull * p + 2] = tri_verts[3ull * t + 2];
}


// ---- PLOC builder (bvh_builder = 1) -----------------------------------------------------------
// Parallel locally-ordered clustering (Meister & Bittner 2018) over the Morton-sorted triangles:
// every cluster looks r places left and right for the neighbour whose merged box has the smallest
// surface area (ties: the lower position), mutual nearest neighbours merge, the list is compacted,
// repeat.  Trees are ~10 % (terrain) to ~20 % (instanced rocks) cheaper to walk than the LBVH of
// build_hlbvh.cl.  The reference's RadeonRays picks its builder inside the missing library
// ("median" by default, "sah" optional — RadeonRays/radeon_rays.h:283-292); both builders here
// feed the same node layouts.  Node ids are final from the start: leaf at Morton position p is
// n-1+p, the k-th inner node created is n-2-k, so the last one (the root) is 0.
static constexpr int kPlocRadius = 2;             // with the adaptive Morton order a SHORT search makes the better tree: node visits per ray on
                                                        // the terrain 14.3 / 14.0 / 14.2 / 14.4 / 14.8 / 15.2 at radius 1 / 2 / 3 / 4 / 8 / 16
static constexpr int kPlocItems = 4;                    // clusters per thread in the scan kernels
static constexpr int kPlocTile = kBlock * kPlocItems;

__device__ __forceinline__ float merged_area(const Box& a, const Box& b)
{
    const float dx = fmaxf(a.pmax[0], b.pmax[0]) - fminf(a.pmin[0], b.pmin[0]);
    const float dy = fmaxf(a.pmax[1], b.pmax[1]) - fminf(a.pmin[1], b.pmin[1]);
    const float dz = fmaxf(a.pmax[2], b.pmax[2]) - fminf(a.pmin[2], b.pmin[2]);
    return fmaf(dz, dx, fmaf(dy, dz, dx * dy));
}

__global__ __launch_bounds__(kBlock)
void k_ploc_init(uint32_t n, uint32_t* __restrict__ cl, uint32_t* __restrict__ state)
{
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) cl[i] = n - 1 + i;
    if (i == 0) { state[0] = n; state[1] = 0; state[2] = 0; state[3] = 0; }     // list length, inner nodes made, no-progress flag
}

// the position of cluster i's nearest neighbour within `radius` places (merged surface area; ties: the lower position)
__device__ __forceinline__ uint32_t ploc_nearest(const uint32_t* cl, uint32_t m, const Box* bounds, int radius, int forced, uint32_t i)
{
    if (forced) return (i ^ 1u) < m ? (i ^ 1u) : i;                   // safety net: plain pairing halves the list
    const Box me = bounds[cl[i]];
    const uint32_t lo = i > (uint32_t)radius ? i - (uint32_t)radius : 0u;
    const uint32_t hi = i + (uint32_t)radius < m - 1 ? i + (uint32_t)radius : m - 1;
    float best = INFINITY; uint32_t bj = i;
    for (uint32_t j = lo; j <= hi; j++) {
        if (j == i) continue;
        const float a = merged_area(me, bounds[cl[j]]);
        if (a < best || bj == i) { best = a; bj = j; }
    }
    return bj;
}

// The round kernels take the list length (state[0]) and the number of inner nodes made so far (state[1]) from device
// memory: the host launches a round with the length of two rounds ago as the grid's upper bound and reads the new length
// back while the next round runs, instead of draining the stream after every round.
__global__ __launch_bounds__(kBlock)
void k_ploc_nn(const uint32_t* __restrict__ cl, const uint32_t* __restrict__ state, const Box* __restrict__ bounds, int radius, int forced,
               uint32_t stop, uint32_t* __restrict__ nn)
{
    const uint32_t m = state[0];
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    // (a list of `stop` clusters or fewer is left as it is — bvh_builder 2 builds the rest of the tree over it top-down — and a round
    // launched on it because the host had not seen that length yet copies it: everybody its own neighbour, nobody merges)
    if (i < m) nn[i] = m <= stop ? i : ploc_nearest(cl, m, bounds, radius, forced, i);
}

// alive: the cluster survives (possibly as the new merged node); merge: it is the lower half of a mutual pair
__device__ __forceinline__ void ploc_flags(const uint32_t* __restrict__ nn, uint32_t m, uint32_t i, uint32_t& alive, uint32_t& merge)
{
    alive = 0; merge = 0;
    if (i >= m) return;
    const uint32_t j = nn[i];
    const bool mutual = j != i && nn[j] == i;
    merge = mutual && i < j;
    alive = !(mutual && i > j);
}

__device__ __forceinline__ uint2 block_exclusive_scan2(uint2 v, uint2* total_out)
{
    __shared__ uint2 wave_sums[kBlock / 64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint2 inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t x = __shfl_up(inc.x, off), y = __shfl_up(inc.y, off);
        if (lane >= (uint32_t)off) { inc.x += x; inc.y += y; }
    }
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    uint2 base = make_uint2(0, 0), total = make_uint2(0, 0);
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) {
        if ((uint32_t)w < wave) { base.x += wave_sums[w].x; base.y += wave_sums[w].y; }
        total.x += wave_sums[w].x; total.y += wave_sums[w].y;
    }
    __syncthreads();
    *total_out = total;
    return make_uint2(base.x + inc.x - v.x, base.y + inc.y - v.y);
}

__global__ __launch_bounds__(kBlock)
void k_ploc_count(const uint32_t* __restrict__ nn, const uint32_t* __restrict__ state, uint2* __restrict__ sums)
{
    const uint32_t m = state[0];                 // tiles beyond the list count nothing
    const uint32_t base = blockIdx.x * kPlocTile + threadIdx.x * kPlocItems;
    uint2 c = make_uint2(0, 0);
#pragma unroll
    for (int k = 0; k < kPlocItems; k++) { uint32_t a, g; ploc_flags(nn, m, base + k, a, g); c.x += a; c.y += g; }
    uint2 total;
    block_exclusive_scan2(c, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// one tile of the compaction: survivors keep their order, the lower half of a mutual pair becomes the merged node
__device__ __forceinline__ void ploc_emit_tile(const uint32_t* cl, const uint32_t* nn, uint32_t m, uint32_t base, uint2 off,
                                               uint32_t created, uint32_t n, uint32_t* nxt, uint32_t* child_left,
                                               uint32_t* child_right, uint32_t* parent, Box* bounds, uint32_t* height, uint2* total_out)
{
    uint32_t al[kPlocItems], mg[kPlocItems];
    uint2 c = make_uint2(0, 0);
#pragma unroll
    for (int k = 0; k < kPlocItems; k++) { ploc_flags(nn, m, base + k, al[k], mg[k]); c.x += al[k]; c.y += mg[k]; }
    const uint2 ex = block_exclusive_scan2(c, total_out);
    uint32_t pos = off.x + ex.x, mk = off.y + ex.y;
#pragma unroll
    for (int k = 0; k < kPlocItems; k++) {
        const uint32_t i = base + k;
        if (!al[k]) continue;
        uint32_t id = cl[i];
        if (mg[k]) {
            const uint32_t a = cl[i], b2 = cl[nn[i]];
            id = n - 2u - (created + mk);                              // the (created + mk)-th inner node
            mk++;
            const Box ba = bounds[a], bb = bounds[b2];
            Box u;
#pragma unroll
            for (int q = 0; q < 3; q++) { u.pmin[q] = fminf(ba.pmin[q], bb.pmin[q]); u.pmax[q] = fmaxf(ba.pmax[q], bb.pmax[q]); }
            bounds[id] = u;
            child_left[id] = a; child_right[id] = b2;
            parent[a] = id; parent[b2] = id;
            // leaves (ids from n - 1) have height 0; the root's height is the depth of the deepest leaf
            height[id] = 1u + max(a < n - 1u ? height[a] : 0u, b2 < n - 1u ? height[b2] : 0u);
        }
        nxt[pos++] = id;
    }
}

// `sums` holds every tile's (alive, merged) counts (k_ploc_count): a tile's offsets are the sums of the tiles before it — at
// most a thousand values, added up by the tile itself rather than by a scan launch of its own; tile 0 also leaves the totals
// for the host.
__global__ __launch_bounds__(kBlock)
void k_ploc_emit(const uint32_t* __restrict__ cl, const uint32_t* __restrict__ nn, const uint32_t* __restrict__ state, const uint2* __restrict__ sums,
                 uint32_t tiles, uint32_t stop, uint32_t* __restrict__ state_out,
                 uint32_t n, uint32_t* __restrict__ nxt, uint32_t* __restrict__ child_left,
                 uint32_t* __restrict__ child_right, uint32_t* __restrict__ parent, Box* __restrict__ bounds, uint32_t* height)
{
    const uint32_t m = state[0], created = state[1];
    uint2 before = make_uint2(0, 0), all = make_uint2(0, 0);
    for (uint32_t i = threadIdx.x; i < tiles; i += kBlock) {
        const uint2 v = sums[i];
        all.x += v.x; all.y += v.y;
        if (i < blockIdx.x) { before.x += v.x; before.y += v.y; }
    }
    uint2 off, grand;
    block_exclusive_scan2(before, &off);
    block_exclusive_scan2(all, &grand);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        state_out[0] = grand.x; state_out[1] = created + grand.y;
        state_out[2] = state[2] | ((m > stop && (grand.x >= m || grand.y == 0)) ? 1u : 0u);      // a round without progress (never observed)
    }
    uint2 total;
    ploc_emit_tile(cl, nn, m, blockIdx.x * kPlocTile + threadIdx.x * kPlocItems, off, created, n, nxt,
                   child_left, child_right, parent, bounds, height, &total);
}

// The last rounds — once the list fits one tile — in ONE launch: the same three steps per round, separated by
// workgroup barriers instead of launches and a host read-back, with the cluster list, its boxes and the neighbour
// indices in LDS (on the 1 M-triangle terrain two thirds of the rounds work on fewer than 1024 clusters).
__global__ __launch_bounds__(kBlock)
void k_ploc_tail(const uint32_t* __restrict__ cl_in, const uint32_t* __restrict__ state, uint32_t n, int iter, int radius,
                 uint32_t* __restrict__ child_left, uint32_t* __restrict__ child_right, uint32_t* __restrict__ parent,
                 Box* __restrict__ bounds, const uint32_t* __restrict__ height, uint32_t* __restrict__ depth_out)
{
    __shared__ Box s_box[2][kPlocTile];
    __shared__ uint32_t s_cl[2][kPlocTile];
    __shared__ uint16_t s_h[2][kPlocTile];              // cluster heights (a tree deeper than 65535 levels is refused long before)
    __shared__ uint32_t s_nn[kPlocTile];
    uint32_t m = state[0], created = state[1];
    for (uint32_t i = threadIdx.x; i < m; i += kBlock) {
        const uint32_t id = cl_in[i];
        s_cl[0][i] = id; s_box[0][i] = bounds[id]; s_h[0][i] = (uint16_t)(id < n - 1u ? height[id] : 0u);
    }
    __syncthreads();
    int cur = 0;
    while (m > 1) {
        const int forced = iter >= 256;
#pragma unroll
        for (int k = 0; k < kPlocItems; k++) {
            const uint32_t i = threadIdx.x * kPlocItems + k;
            if (i >= m) continue;
            uint32_t bj = (i ^ 1u) < m ? (i ^ 1u) : i;                 // forced: plain pairing (ploc_nearest)
            if (!forced) {
                const Box me = s_box[cur][i];
                const uint32_t lo = i > (uint32_t)radius ? i - (uint32_t)radius : 0u;
                const uint32_t hi = i + (uint32_t)radius < m - 1 ? i + (uint32_t)radius : m - 1;
                float best = INFINITY; bj = i;
                for (uint32_t j = lo; j <= hi; j++) {
                    if (j == i) continue;
                    const float a = merged_area(me, s_box[cur][j]);
                    if (a < best || bj == i) { best = a; bj = j; }
                }
            }
            s_nn[i] = bj;
        }
        __syncthreads();
        const uint32_t base = threadIdx.x * kPlocItems;
        uint32_t al[kPlocItems], mg[kPlocItems];
        uint2 c = make_uint2(0, 0);
#pragma unroll
        for (int k = 0; k < kPlocItems; k++) { ploc_flags(s_nn, m, base + k, al[k], mg[k]); c.x += al[k]; c.y += mg[k]; }
        uint2 total;
        const uint2 ex = block_exclusive_scan2(c, &total);
        uint32_t pos = ex.x, mk = ex.y;
#pragma unroll
        for (int k = 0; k < kPlocItems; k++) {
            const uint32_t i = base + k;
            if (!al[k]) continue;
            uint32_t id = s_cl[cur][i];
            Box u = s_box[cur][i];
            uint16_t h = s_h[cur][i];
            if (mg[k]) {
                const uint32_t j = s_nn[i], a = id, b2 = s_cl[cur][j];
                id = n - 2u - (created + mk);                          // as ploc_emit_tile
                mk++;
                const Box bb = s_box[cur][j];
#pragma unroll
                for (int q = 0; q < 3; q++) { u.pmin[q] = fminf(u.pmin[q], bb.pmin[q]); u.pmax[q] = fmaxf(u.pmax[q], bb.pmax[q]); }
                bounds[id] = u;
                child_left[id] = a; child_right[id] = b2;
                parent[a] = id; parent[b2] = id;
                h = (uint16_t)(1u + max((uint32_t)h, (uint32_t)s_h[cur][j]));
            }
            s_cl[cur ^ 1][pos] = id; s_box[cur ^ 1][pos] = u; s_h[cur ^ 1][pos] = h;
            pos++;
        }
        __syncthreads();
        if (total.x >= m || total.y == 0) iter = 256;                  // no progress (never observed): pair up next round
        else { m = total.x; created += total.y; cur ^= 1; iter++; }
    }
    if (threadIdx.x == 0) *depth_out = s_h[cur][0];                    // the root
}

__global__ void k_set_root_parent(uint32_t* parent) { if (threadIdx.x == 0) parent[0] = 0xffffffffu; }

static int build_sah_top(AobBake* b, uint32_t n, const uint32_t* cl, uint32_t* other, const uint32_t* state, uint32_t bound);

// stop: the list length at which clustering ends (1: PLOC builds the whole tree; bvh_builder 2: kSahTopClusters)
static int build_ploc(AobBake* b, uint32_t n, uint32_t stop)
{
    hipStream_t s = b->stream;
    uint32_t* cl[2] = { b->prim_idx[0], (uint32_t*)b->codes[0] };      // both free once the sort is done
    uint32_t* nn = b->refit_flags;
    hipLaunchKernelGGL(k_ploc_init, dim3(blocks_for(n)), dim3(kBlock), 0, s, n, cl[0], b->ploc_totals);
    // state[k] = { list length, inner nodes made, no-progress flag, - }: round r reads state[r & 1] and writes the other
    uint32_t* state[2] = { b->ploc_totals, b->ploc_totals + 4 };
    uint32_t* host[2] = { b->pinned + 4, b->pinned + 8 };
    // a round is launched with the length read back after the round before last as its upper bound
    uint32_t known = n;                                                // the latest length the host has seen
    int cur = 0, iter = 0;
    while (known > max((uint32_t)kPlocTile, stop)) {
        const int forced = iter >= 256;                                 // never observed; guarantees termination
        const uint32_t tiles = (known + kPlocTile - 1) / kPlocTile;
        const uint32_t* st = state[iter & 1]; uint32_t* st_out = state[(iter & 1) ^ 1];
        hipLaunchKernelGGL(k_ploc_nn, dim3(blocks_for(known)), dim3(kBlock), 0, s, cl[cur], st, b->node_bounds, kPlocRadius, forced, stop, nn);
        hipLaunchKernelGGL(k_ploc_count, dim3(tiles), dim3(kBlock), 0, s, nn, st, b->ploc_sums);
        hipLaunchKernelGGL(k_ploc_emit, dim3(tiles), dim3(kBlock), 0, s, cl[cur], nn, st, b->ploc_sums, tiles, stop, st_out, n,
                           cl[cur ^ 1], b->child_left, b->child_right, b->parent, b->node_bounds, b->node_height);
        AOB_HIP(hipMemcpyAsync(host[iter & 1], st_out, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        AOB_HIP(hipEventRecord(b->ev_ploc[iter & 1], s));
        if (iter >= 1) {
            // the length after the round BEFORE this one: its copy finished before this round's kernels began
            AOB_HIP(hipEventSynchronize(b->ev_ploc[(iter - 1) & 1]));
            known = host[(iter - 1) & 1][0];
        }
        cur ^= 1; iter++;
        if (iter > 4096) { set_error("PLOC does not converge"); return AOB_ERR_DEVICE; }
    }
    if (iter >= 1) {
        AOB_HIP(hipEventSynchronize(b->ev_ploc[(iter - 1) & 1]));
        if (host[(iter - 1) & 1][2]) { set_error("PLOC made no progress"); return AOB_ERR_DEVICE; }
    }
    if (stop > 1u) {
        // bvh_builder 2: the clusters made so far (at most `stop` of them, or every leaf of a small scene) become the leaves of a
        // top-down binned-SAH tree
        AOB_TRY(build_sah_top(b, n, cl[cur], cl[cur ^ 1], state[iter & 1], min(known, max(stop, (uint32_t)kPlocTile))));
        hipLaunchKernelGGL(k_set_root_parent, dim3(1), dim3(64), 0, s, b->parent);
        return AOB_OK;
    }
    // the list now holds at most one tile (the last round launched may have run on less): the rest in one launch
    hipLaunchKernelGGL(k_ploc_tail, dim3(1), dim3(kBlock), 0, s, cl[cur], state[iter & 1], n, iter, kPlocRadius,
                       b->child_left, b->child_right, b->parent, b->node_bounds, b->node_height, b->depth_dev);
    hipLaunchKernelGGL(k_set_root_parent, dim3(1), dim3(64), 0, s, b->parent);
    return AOB_OK;
}

// ---- binned-SAH top (bvh_builder = 2) ------------------------------------------------------------
// The reference stack's own lever for traversal cost is its SAH builder (RadeonRays/radeon_rays.h:286-294: bvh.builder = "sah").
// Here: PLOC clusters the Morton-ordered triangles until at most kSahTopClusters clusters are left (a smaller scene: none at
// all), and the tree above them is built TOP-DOWN by binned SAH — where an agglomerated tree differs most from a SAH tree
// (1 M-triangle terrain: 14.24 instead of 15.00 node visits per ray; every triangle a cluster of its own would give 13.65,
// profiles/r03/sah_top/).  The algorithm (DESIGN.md §3.2 states it once, for this file and for its CPU twin): per list range the box of the clusters'
// centres, 16 bins per axis, the cheapest of the 3 x 15 planes, a STABLE partition; a node's id is the position its range was
// cut at (unique; the root and the node cut at position 0 swap ids so that the root is node 0) — so ranges can be split in
// any order: one workgroup per range while ranges are long, one thread per range (and all below it) once they are short.
static constexpr int kSahTopClusters = 131072;
static constexpr int kSahBins = 16;
static constexpr int kSahSmall = 32;               // ranges of at most this many clusters are finished by one wavefront each
static constexpr int kSahBig = 8192;               // ranges of more than this many are cut by kSahSlices workgroups together
static constexpr int kSahSlices = 16;
// A range of the list that is to be cut: parent << 1 | side (0xffffffff: the root); depth << 2 | boxes known << 1 | the list it
// lives in; its clusters' box and the box of their centres, handed down by the cut that made it (the bins of the parent on either
// side of the plane), so that a range is read twice, not three times.
struct SahRange { uint32_t begin, end, parent_side, depth_buf; float box[6], cbox[6]; };
struct SahShort { uint32_t begin, end, parent_side, depth_buf; };       // at most kSahSmall clusters: finished by one wavefront

__device__ __forceinline__ int sah_bin(float c, float cmin, float ext)
{
    int q = (int)((c - cmin) / ext * (float)kSahBins);
    return q > kSahBins - 1 ? kSahBins - 1 : q;
}
__device__ __forceinline__ float box_area(const float* b)          // b: min[3], max[3]
{
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
}

// The bins of one group (a workgroup or a wavefront) in LDS: per axis and bin the clusters' count, their box and the box of their
// centres (ordered-uint encoded: LDS atomics).
struct SahBins { uint32_t cnt[3][kSahBins]; uint32_t box[3][kSahBins][12]; };
__device__ __forceinline__ void sah_bins_clear(SahBins* bins, uint32_t tid, uint32_t threads)
{
    for (uint32_t i = tid; i < 3u * kSahBins; i += threads) {
        (&bins->cnt[0][0])[i] = 0u;
        for (int q = 0; q < 12; q++) (&bins->box[0][0][0])[i * 12u + q] = (q % 6) < 3 ? 0xffffffffu : 0u;
    }
}
// A thread walks a run of consecutive clusters and keeps, per axis, the box of the clusters it has seen since their bin last
// changed (neighbours in the list mostly share a bin: it is in Morton order); a change of bin — and the end of the run — sends
// the box to LDS with thirteen atomics.  (One atomic per cluster and value put up to 64 lanes on one LDS word; reducing across the
// lanes first kept the LDS pipe busy with ds_bpermute instead.)
struct SahRun { int bin; uint32_t cnt; float v[12]; };
__device__ __forceinline__ void sah_run_flush(SahBins* bins, int ax, SahRun& r)
{
    if (r.cnt) {
        atomicAdd(&bins->cnt[ax][r.bin], r.cnt);
        for (int q = 0; q < 12; q++) { if ((q % 6) < 3) atomicMin(&bins->box[ax][r.bin][q], enc_float(r.v[q])); else atomicMax(&bins->box[ax][r.bin][q], enc_float(r.v[q])); }
    }
    r.cnt = 0;
}
__device__ __forceinline__ void sah_run_add(SahBins* bins, int ax, SahRun& r, const Box& bx, float cmin, float ext)
{
    float c[3];
    for (int q = 0; q < 3; q++) c[q] = (bx.pmin[q] + bx.pmax[q]) * 0.5f;
    const int b = sah_bin(c[ax], cmin, ext);
    if (b != r.bin || r.cnt == 0u) {
        sah_run_flush(bins, ax, r);
        r.bin = b;
        for (int q = 0; q < 3; q++) { r.v[q] = INFINITY; r.v[3 + q] = -INFINITY; r.v[6 + q] = INFINITY; r.v[9 + q] = -INFINITY; }
    }
    r.cnt++;
    for (int q = 0; q < 3; q++) {
        r.v[q] = fminf(r.v[q], bx.pmin[q]); r.v[3 + q] = fmaxf(r.v[3 + q], bx.pmax[q]);
        r.v[6 + q] = fminf(r.v[6 + q], c[q]); r.v[9 + q] = fmaxf(r.v[9 + q], c[q]);
    }
}

// The cut of a range from its bins, by one whole wavefront: lane 16 * axis + q holds bin q of the axis; unions of the bins up to q
// and from q on by shuffles inside each group of 16 lanes; the plane behind bin q costs area(left) * count(left) + area(right) *
// count(right); the cheapest wins, the lowest lane among equals (the first plane in axis, then bin order: the oracle's sweep).
// Returns the axis (-1: no plane leaves clusters on both sides), and for either side count, box and box of the centres.
struct SahCut { int axis, split; uint32_t nl; float lbox[12], rbox[12]; };
__device__ __forceinline__ SahCut sah_select(const SahBins* bins, const float* ext)
{
    const uint32_t lane = threadIdx.x & 63u;
    const int ax = (int)(lane >> 4), q = (int)(lane & 15u);
    const bool valid = lane < 48u && ext[ax < 3 ? ax : 0] > 0.0f;
    uint32_t pc = valid ? bins->cnt[ax][q] : 0u;
    float pb[12];
    for (int k = 0; k < 12; k++) pb[k] = valid && pc ? dec_float(bins->box[ax][q][k]) : ((k % 6) < 3 ? INFINITY : -INFINITY);
    uint32_t sc = pc; float sb[12];
    for (int k = 0; k < 12; k++) sb[k] = pb[k];
    for (int off = 1; off < 16; off <<= 1) {
        const uint32_t oc = __shfl_up(pc, off, 16), oc2 = __shfl_down(sc, off, 16);
        float ob[12], ob2[12];
        for (int k = 0; k < 12; k++) { ob[k] = __shfl_up(pb[k], off, 16); ob2[k] = __shfl_down(sb[k], off, 16); }
        if (q >= off) { pc += oc; for (int k = 0; k < 12; k++) pb[k] = (k % 6) < 3 ? fminf(pb[k], ob[k]) : fmaxf(pb[k], ob[k]); }
        if (q + off < 16) { sc += oc2; for (int k = 0; k < 12; k++) sb[k] = (k % 6) < 3 ? fminf(sb[k], ob2[k]) : fmaxf(sb[k], ob2[k]); }
    }
    // the right side of the plane behind bin q: the bins from q + 1 on
    const uint32_t rc = __shfl_down(sc, 1, 16);
    float rb[12];
    for (int k = 0; k < 12; k++) rb[k] = __shfl_down(sb[k], 1, 16);
    float cost = INFINITY;
    if (valid && q < kSahBins - 1 && pc != 0u && rc != 0u) cost = box_area(pb) * (float)pc + box_area(rb) * (float)rc;
    float best = cost; uint32_t bl = lane;
    for (int off = 32; off > 0; off >>= 1) {
        const float oc = __shfl_xor(best, off); const uint32_t ol = __shfl_xor(bl, off);
        if (oc < best || (oc == best && ol < bl)) { best = oc; bl = ol; }
    }
    SahCut c;
    c.axis = best < INFINITY ? (int)(bl >> 4) : -1;
    c.split = (int)(bl & 15u) + 1;
    c.nl = __shfl(pc, (int)bl);
    for (int k = 0; k < 12; k++) { c.lbox[k] = __shfl(pb[k], (int)bl); c.rbox[k] = __shfl(rb[k], (int)bl); }
    return c;
}

// what a range that has been cut leaves behind: its node (id from the cut position), the link from its parent, its children —
// single clusters at once, longer ranges as new work
__device__ __forceinline__ void sah_emit(uint32_t begin, uint32_t end, uint32_t parent_side, uint32_t depth, uint32_t mid, const float* node_box,
                                         const SahCut* cut, const uint32_t* list, uint32_t out_buf,
                                         uint32_t n, uint32_t* root_k, uint32_t* child_left, uint32_t* child_right, uint32_t* parent, Box* bounds,
                                         const uint32_t* height, uint32_t* depth_out,
                                         SahRange* next, uint32_t* next_count, SahShort* small, uint32_t* small_count,
                                         SahRange* big = nullptr, uint32_t* big_count = nullptr, bool emit_single = true)
{
    const uint32_t k = mid - 1u;
    const bool root = parent_side == 0xffffffffu;
    if (root) *root_k = k;
    const uint32_t id = root ? 0u : (k == 0u ? *root_k : k);
    Box bx;
    for (int q = 0; q < 3; q++) { bx.pmin[q] = node_box[q]; bx.pmax[q] = node_box[3 + q]; }
    bounds[id] = bx;
    if (!root) {
        const uint32_t par = parent_side >> 1;
        if (parent_side & 1u) child_right[par] = id; else child_left[par] = id;
        parent[id] = par;
    }
    const uint32_t rb[2] = { begin, mid }, re[2] = { mid, end };
    for (uint32_t side = 0; side < 2u; side++) {
        const uint32_t cb = rb[side], ce = re[side];
        if (ce - cb == 1u) {
            if (!emit_single) continue;                                             // (k_big_move links it when it moves it)
            const uint32_t c = list[cb];
            if (side) child_right[id] = c; else child_left[id] = c;
            parent[c] = id;
            atomicMax(depth_out, depth + 1u + (c < n - 1u ? height[c] : 0u));       // a cluster is a PLOC subtree of that height (a leaf: 0)
        } else if (ce - cb > (uint32_t)kSahSmall) {
            SahRange ch; ch.begin = cb; ch.end = ce; ch.parent_side = id << 1 | side;
            const bool known = cut && cut->axis >= 0;
            ch.depth_buf = (depth + 1u) << 2 | (known ? 2u : 0u) | out_buf;
            const float* sbx = known ? (side ? cut->rbox : cut->lbox) : node_box;
            for (int q = 0; q < 6; q++) { ch.box[q] = sbx[q]; ch.cbox[q] = known ? sbx[6 + q] : 0.0f; }
            if (big && known && ce - cb > (uint32_t)kSahBig) big[atomicAdd(big_count, 1u)] = ch;
            else next[atomicAdd(next_count, 1u)] = ch;
        } else {
            SahShort ch; ch.begin = cb; ch.end = ce; ch.parent_side = id << 1 | side; ch.depth_buf = (depth + 1u) << 2 | out_buf;
            small[atomicAdd(small_count, 1u)] = ch;
        }
    }
}

// The clusters of a range travel with their boxes: list `buf` is ids[buf][i] + boxes[buf][i], contiguous per range, so that every
// pass streams its range (gathering node_bounds[ids[i]] made the 131 072-cluster root range alone cost milliseconds).
struct SahLists { uint32_t* ids[2]; Box* boxes[2]; };

// one workgroup per range of more than kSahSmall clusters: bins in LDS, a stable partition into the other list
static constexpr int kSahBlock = 1024;
__global__ __launch_bounds__(kSahBlock)
void k_sah_level(const SahRange* __restrict__ ranges, const uint32_t* __restrict__ count, SahLists lists,
                 uint32_t n, uint32_t* root_k, uint32_t* __restrict__ child_left, uint32_t* __restrict__ child_right, uint32_t* __restrict__ parent,
                 Box* __restrict__ bounds, const uint32_t* __restrict__ height, uint32_t* depth_out,
                 SahRange* __restrict__ next, uint32_t* next_count, SahShort* __restrict__ small, uint32_t* small_count)
{
    constexpr int kWaves = kSahBlock / 64;
    __shared__ uint32_t s_enc[12];                              // node box (6) and box of the centres (6), ordered-uint encoded
    __shared__ SahBins s_bins;
    __shared__ SahCut s_cut;
    __shared__ uint32_t s_wave[kWaves];
    if (blockIdx.x >= *count) return;
    const SahRange r = ranges[blockIdx.x];
    const uint32_t from = r.depth_buf & 1u, to = from ^ 1u;
    const uint32_t* src = lists.ids[from]; const Box* src_box = lists.boxes[from];
    uint32_t* dst = lists.ids[to]; Box* dst_box = lists.boxes[to];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t len = r.end - r.begin;
    sah_bins_clear(&s_bins, tid, kSahBlock);
    float node_box[6], cbox[6];
    if (r.depth_buf & 2u) {
        for (int q = 0; q < 6; q++) { node_box[q] = r.box[q]; cbox[q] = r.cbox[q]; }
        __syncthreads();
    } else {
        // (the root, and the halves of a range no plane could cut) the clusters' box and the box of their centres
        if (tid < 12u) s_enc[tid] = (tid % 6u) < 3u ? 0xffffffffu : 0u;
        __syncthreads();
        float v[12];
        for (int q = 0; q < 12; q++) v[q] = (q % 6) < 3 ? INFINITY : -INFINITY;
        for (uint32_t i = r.begin + tid; i < r.end; i += kSahBlock) {
            const Box bx = src_box[i];
            for (int q = 0; q < 3; q++) {
                v[q] = fminf(v[q], bx.pmin[q]); v[3 + q] = fmaxf(v[3 + q], bx.pmax[q]);
                const float c = (bx.pmin[q] + bx.pmax[q]) * 0.5f;
                v[6 + q] = fminf(v[6 + q], c); v[9 + q] = fmaxf(v[9 + q], c);
            }
        }
        for (int q = 0; q < 12; q++)
            for (int off = 32; off > 0; off >>= 1) v[q] = (q % 6) < 3 ? fminf(v[q], __shfl_xor(v[q], off)) : fmaxf(v[q], __shfl_xor(v[q], off));
        if (lane == 0u) for (int q = 0; q < 12; q++) { if ((q % 6) < 3) atomicMin(&s_enc[q], enc_float(v[q])); else atomicMax(&s_enc[q], enc_float(v[q])); }
        __syncthreads();
        for (int q = 0; q < 6; q++) { node_box[q] = dec_float(s_enc[q]); cbox[q] = dec_float(s_enc[6 + q]); }
    }
    float cmin[3], ext[3];
    for (int q = 0; q < 3; q++) { cmin[q] = cbox[q]; ext[q] = cbox[3 + q] - cbox[q]; }
    // the bins: every thread a run of consecutive clusters
    {
        const uint32_t chunk = (len + kSahBlock - 1u) / kSahBlock;
        const uint32_t i0 = r.begin + tid * chunk, i1 = min(r.end, i0 + chunk);
        SahRun run[3];
        for (int ax = 0; ax < 3; ax++) { run[ax].bin = -1; run[ax].cnt = 0u; }
        for (uint32_t i = i0; i < i1; i++) {
            const Box bx = src_box[i];
            for (int ax = 0; ax < 3; ax++) if (ext[ax] > 0.0f) sah_run_add(&s_bins, ax, run[ax], bx, cmin[ax], ext[ax]);
        }
        for (int ax = 0; ax < 3; ax++) sah_run_flush(&s_bins, ax, run[ax]);
    }
    __syncthreads();
    if (wave == 0u) { const SahCut c = sah_select(&s_bins, ext); if (lane == 0u) s_cut = c; }
    __syncthreads();
    const int axis = s_cut.axis, split = s_cut.split;
    const uint32_t nl = axis >= 0 ? s_cut.nl : len / 2u;        // no plane separates the centres: the list is cut in the middle
    // stable partition into the other list, a tile of kSahBlock clusters at a time
    uint32_t lbase = r.begin, rbase = r.begin + nl;
    for (uint32_t t0 = r.begin; t0 < r.end; t0 += kSahBlock) {
        const uint32_t i = t0 + tid;
        const bool in = i < r.end;
        uint32_t c = 0; bool goes_left = false; Box bx;
        if (in) {
            c = src[i]; bx = src_box[i];
            if (axis < 0) goes_left = i - r.begin < nl;
            else goes_left = sah_bin((bx.pmin[axis] + bx.pmax[axis]) * 0.5f, cmin[axis], ext[axis]) < split;
        }
        const uint64_t lm = __ballot(in && goes_left), rm = __ballot(in && !goes_left);
        if (lane == 0u) s_wave[wave] = (uint32_t)__popcll(lm) | (uint32_t)__popcll(rm) << 16;
        __syncthreads();
        uint32_t lbefore = 0, rbefore = 0, ltotal = 0, rtotal = 0;
        for (uint32_t w = 0; w < (uint32_t)kWaves; w++) {
            const uint32_t v = s_wave[w];
            if (w < wave) { lbefore += v & 0xffffu; rbefore += v >> 16; }
            ltotal += v & 0xffffu; rtotal += v >> 16;
        }
        const uint64_t below = lane ? (~0ull >> (64u - lane)) : 0ull;
        if (in) {
            const uint32_t at = goes_left ? lbase + lbefore + (uint32_t)__popcll(lm & below) : rbase + rbefore + (uint32_t)__popcll(rm & below);
            dst[at] = c; dst_box[at] = bx;
        }
        lbase += ltotal; rbase += rtotal;
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0)
        sah_emit(r.begin, r.end, r.parent_side, r.depth_buf >> 2, r.begin + nl, node_box, &s_cut, dst, to, n, root_k, child_left, child_right, parent, bounds,
                 height, depth_out, next, next_count, small, small_count);
}

// one WAVEFRONT per range of at most kSahSmall (= 32) clusters: the range and everything below it, a lane per cluster, the clusters
// and the stack of sub-ranges in LDS.  The same cut as above, without bins: lane 15 * axis + q takes the plane behind bin q and
// walks the sub-range's clusters (broadcast reads), collecting what lies on either side of it.
static constexpr int kSahWaveBlock = 256;
__global__ __launch_bounds__(kSahWaveBlock)
void k_sah_wave(const SahShort* __restrict__ ranges, const uint32_t* __restrict__ count, SahLists lists,
                uint32_t n, uint32_t* root_k, uint32_t* __restrict__ child_left, uint32_t* __restrict__ child_right, uint32_t* __restrict__ parent,
                Box* __restrict__ bounds, const uint32_t* __restrict__ height, uint32_t* depth_out)
{
    constexpr int kWaves = kSahWaveBlock / 64;
    __shared__ Box s_box[kWaves][kSahSmall];
    __shared__ uint32_t s_id[kWaves][kSahSmall], s_bin3[kWaves][kSahSmall];
    __shared__ uint32_t s_todo[kWaves][kSahSmall][4];           // begin | end << 8, parent << 1 | side, depth, boxes known
    __shared__ float s_tbox[kWaves][kSahSmall][12];             // ... and the sub-range's box and box of centres, if known
    __shared__ float s_cut[kWaves][26];                         // the winning plane: count, left box + centres, right box + centres
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t t = blockIdx.x * kWaves + wave;
    if (t >= *count) return;                                    // (wave-uniform; no workgroup barrier below)
    const SahShort r0 = ranges[t];
    const uint32_t from = r0.depth_buf & 1u;
    const uint32_t total = r0.end - r0.begin;
    Box* box = s_box[wave]; uint32_t* id = s_id[wave]; uint32_t* bin3 = s_bin3[wave];
    if (lane < total) { box[lane] = lists.boxes[from][r0.begin + lane]; id[lane] = lists.ids[from][r0.begin + lane]; }
    if (lane == 0) { s_todo[wave][0][0] = 0u | total << 8; s_todo[wave][0][1] = r0.parent_side; s_todo[wave][0][2] = r0.depth_buf >> 2; s_todo[wave][0][3] = 0u; }
    int top = 1;
    uint32_t my_root_k = 0xffffffffu;
    uint32_t deepest = 0;                                       // (lane 0) one atomic per wavefront at the end: every leaf's own atomicMax on the one
                                                                // word took 1.5 ms for the 160 000 clusters of the terrain
    const uint32_t shared_root_k = r0.parent_side == 0xffffffffu ? 0u : *root_k;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    while (top) {
        top--;
        const uint32_t be = s_todo[wave][top][0], parent_side = s_todo[wave][top][1], depth = s_todo[wave][top][2], known = s_todo[wave][top][3];
        const uint32_t b = be & 0xffu, e = be >> 8, cnt = e - b;
        const bool in = lane >= b && lane < e;
        float v[12];                                            // box min, box max, centres min, centres max (wave-uniform)
        if (known) { for (int q = 0; q < 12; q++) v[q] = s_tbox[wave][top][q]; }
        else {
            for (int q = 0; q < 12; q++) v[q] = (q % 6) < 3 ? INFINITY : -INFINITY;
            for (uint32_t i = b; i < e; i++) {
                const Box bx = box[i];
                for (int q = 0; q < 3; q++) {
                    v[q] = fminf(v[q], bx.pmin[q]); v[3 + q] = fmaxf(v[3 + q], bx.pmax[q]);
                    const float c = (bx.pmin[q] + bx.pmax[q]) * 0.5f;
                    v[6 + q] = fminf(v[6 + q], c); v[9 + q] = fmaxf(v[9 + q], c);
                }
            }
        }
        float cmin[3], ext[3];
        for (int q = 0; q < 3; q++) { cmin[q] = v[6 + q]; ext[q] = v[9 + q] - v[6 + q]; }
        Box bx;
        uint32_t me = 0, mybins = 0;
        if (in) {
            bx = box[lane]; me = id[lane];
            for (int ax = 0; ax < 3; ax++) if (ext[ax] > 0.0f) mybins |= (uint32_t)sah_bin((bx.pmin[ax] + bx.pmax[ax]) * 0.5f, cmin[ax], ext[ax]) << (4 * ax);
            bin3[lane] = mybins;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // lane 15 * ax + q: the plane behind bin q of axis ax
        const int cax = (int)(lane / 15u), cq = (int)(lane % 15u);
        float cost = INFINITY;
        uint32_t lc = 0;
        float lb[12], rb[12];
        for (int q = 0; q < 12; q++) { lb[q] = rb[q] = (q % 6) < 3 ? INFINITY : -INFINITY; }
        if (lane < 45u && ext[cax < 3 ? cax : 0] > 0.0f) {
            for (uint32_t i = b; i < e; i++) {
                const Box ib = box[i];
                const bool left = (int)((bin3[i] >> (4 * cax)) & 15u) <= cq;
                lc += left ? 1u : 0u;
                for (int q = 0; q < 3; q++) {
                    // (neutral elements on the side the cluster is not on: no indexing of registers by side)
                    const float c = (ib.pmin[q] + ib.pmax[q]) * 0.5f;
                    lb[q] = fminf(lb[q], left ? ib.pmin[q] : INFINITY); lb[3 + q] = fmaxf(lb[3 + q], left ? ib.pmax[q] : -INFINITY);
                    lb[6 + q] = fminf(lb[6 + q], left ? c : INFINITY); lb[9 + q] = fmaxf(lb[9 + q], left ? c : -INFINITY);
                    rb[q] = fminf(rb[q], left ? INFINITY : ib.pmin[q]); rb[3 + q] = fmaxf(rb[3 + q], left ? -INFINITY : ib.pmax[q]);
                    rb[6 + q] = fminf(rb[6 + q], left ? INFINITY : c); rb[9 + q] = fmaxf(rb[9 + q], left ? -INFINITY : c);
                }
            }
            const uint32_t rc = cnt - lc;
            if (lc != 0u && rc != 0u) cost = box_area(lb) * (float)lc + box_area(rb) * (float)rc;
        }
        float best = cost; uint32_t bl = lane;
        for (int off = 32; off > 0; off >>= 1) {
            const float oc = __shfl_xor(best, off); const uint32_t ol = __shfl_xor(bl, off);
            if (oc < best || (oc == best && ol < bl)) { best = oc; bl = ol; }
        }
        const int axis = best < INFINITY ? (int)(bl / 15u) : -1;
        const int split = (int)(bl % 15u) + 1;
        if (axis >= 0 && lane == bl) { s_cut[wave][0] = __uint_as_float(lc); for (int q = 0; q < 12; q++) { s_cut[wave][1 + q] = lb[q]; s_cut[wave][13 + q] = rb[q]; } }
        // stable partition of the lanes b .. e-1
        bool goes_left = in && lane - b < cnt / 2u;
        if (axis >= 0) goes_left = in && (int)((mybins >> (4 * axis)) & 15u) < split;
        const uint64_t lm = __ballot(goes_left), rm = __ballot(in && !goes_left);
        const uint32_t nl = (uint32_t)__popcll(lm);
        const uint64_t below = lane ? (~0ull >> (64u - lane)) : 0ull;
        if (in) {
            const uint32_t at = goes_left ? b + (uint32_t)__popcll(lm & below) : b + nl + (uint32_t)__popcll(rm & below);
            box[at] = bx; id[at] = me;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const uint32_t mid = b + nl;
        // the node (lane 0 writes; every lane keeps the stack pointer)
        const uint32_t k = r0.begin + mid - 1u;
        const bool root = parent_side == 0xffffffffu;
        if (root) my_root_k = k;
        const uint32_t rk = my_root_k != 0xffffffffu ? my_root_k : shared_root_k;
        const uint32_t nid = root ? 0u : (k == 0u ? rk : k);
        if (lane == 0) {
            if (root) *root_k = k;
            Box nb;
            for (int q = 0; q < 3; q++) { nb.pmin[q] = v[q]; nb.pmax[q] = v[3 + q]; }
            bounds[nid] = nb;
            if (!root) {
                const uint32_t par = parent_side >> 1;
                if (parent_side & 1u) child_right[par] = nid; else child_left[par] = nid;
                parent[nid] = par;
            }
        }
        const uint32_t rb2[2] = { b, mid }, re2[2] = { mid, e };
        for (int side = 1; side >= 0; side--) {
            if (re2[side] - rb2[side] == 1u) {
                if (lane == 0) {
                    const uint32_t c = id[rb2[side]];
                    if (side) child_right[nid] = c; else child_left[nid] = c;
                    parent[c] = nid;
                    deepest = max(deepest, depth + 1u + (c < n - 1u ? height[c] : 0u));
                }
            } else {
                if (lane == 0) { s_todo[wave][top][0] = rb2[side] | re2[side] << 8; s_todo[wave][top][1] = nid << 1 | (uint32_t)side; s_todo[wave][top][2] = depth + 1u; s_todo[wave][top][3] = axis >= 0 ? 1u : 0u; }
                if (axis >= 0 && lane < 12u) s_tbox[wave][top][lane] = s_cut[wave][1 + 12 * side + lane];
                top++;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) atomicMax(depth_out, deepest);
}

// ---- ranges of more than kSahBig clusters: kSahSlices workgroups each ----
// One workgroup walking 100 000 clusters three times is most of a millisecond; the longest ranges — the first few levels — are
// cut by kSahSlices workgroups each: every workgroup bins its slice (k_big_bins), one wavefront per range adds the slices' bins
// up, picks the plane, makes the node and tells every slice where its left and right clusters go (k_big_cut), the workgroups move
// them (k_big_move).  Same bins, same plane, same order of the clusters as one workgroup would have made.
struct SahBigCut { int axis, split; uint32_t node, depth, nl, lbase[kSahSlices], rbase[kSahSlices]; float cmin[3], ext[3]; };

__device__ __forceinline__ void sah_slice(const SahRange& r, uint32_t g, uint32_t& s0, uint32_t& s1)
{
    const uint32_t len = r.end - r.begin, per = (len + kSahSlices - 1u) / kSahSlices;
    s0 = min(r.end, r.begin + g * per); s1 = min(r.end, s0 + per);
}

__global__ __launch_bounds__(kBlock)
void k_big_bins(const SahRange* __restrict__ ranges, const uint32_t* __restrict__ count, SahLists lists, SahBins* __restrict__ gbins)
{
    __shared__ SahBins s_bins;
    const uint32_t ri = blockIdx.x / kSahSlices, g = blockIdx.x % kSahSlices;
    if (ri >= *count) return;
    const SahRange r = ranges[ri];
    const Box* src_box = lists.boxes[r.depth_buf & 1u];
    uint32_t s0, s1;
    sah_slice(r, g, s0, s1);
    const uint32_t tid = threadIdx.x;
    sah_bins_clear(&s_bins, tid, kBlock);
    __syncthreads();
    float cmin[3], ext[3];
    for (int q = 0; q < 3; q++) { cmin[q] = r.cbox[q]; ext[q] = r.cbox[3 + q] - r.cbox[q]; }
    const uint32_t chunk = (s1 - s0 + kBlock - 1u) / kBlock;
    const uint32_t i0 = s0 + tid * chunk, i1 = min(s1, i0 + chunk);
    SahRun run[3];
    for (int ax = 0; ax < 3; ax++) { run[ax].bin = -1; run[ax].cnt = 0u; }
    for (uint32_t i = i0; i < i1; i++) {
        const Box bx = src_box[i];
        for (int ax = 0; ax < 3; ax++) if (ext[ax] > 0.0f) sah_run_add(&s_bins, ax, run[ax], bx, cmin[ax], ext[ax]);
    }
    for (int ax = 0; ax < 3; ax++) sah_run_flush(&s_bins, ax, run[ax]);
    __syncthreads();
    uint32_t* out = (uint32_t*)&gbins[(uint64_t)ri * kSahSlices + g];
    for (uint32_t i = tid; i < sizeof(SahBins) / 4u; i += kBlock) out[i] = ((const uint32_t*)&s_bins)[i];
}

__global__ __launch_bounds__(64)
void k_big_cut(const SahRange* __restrict__ ranges, const uint32_t* __restrict__ count, const SahBins* __restrict__ gbins, SahBigCut* __restrict__ cuts,
               uint32_t n, uint32_t* root_k, uint32_t* __restrict__ child_left, uint32_t* __restrict__ child_right, uint32_t* __restrict__ parent,
               Box* __restrict__ bounds, const uint32_t* __restrict__ height, uint32_t* depth_out,
               SahRange* __restrict__ big_next, uint32_t* big_next_count, SahRange* __restrict__ next, uint32_t* next_count,
               SahShort* __restrict__ small, uint32_t* small_count)
{
    __shared__ SahBins s_bins;
    const uint32_t ri = blockIdx.x, lane = threadIdx.x;
    if (ri >= *count) return;
    const SahRange r = ranges[ri];
    // the slices' bins, added up (sums, minima and maxima of ordered-uint codes: any order)
    for (uint32_t i = lane; i < 3u * kSahBins; i += 64u) {
        uint32_t c = 0;
        for (uint32_t g = 0; g < (uint32_t)kSahSlices; g++) c += (&gbins[(uint64_t)ri * kSahSlices + g].cnt[0][0])[i];
        (&s_bins.cnt[0][0])[i] = c;
    }
    for (uint32_t i = lane; i < 3u * kSahBins * 12u; i += 64u) {
        const bool is_min = (i % 6u) < 3u;
        uint32_t v = is_min ? 0xffffffffu : 0u;
        for (uint32_t g = 0; g < (uint32_t)kSahSlices; g++) { const uint32_t x = (&gbins[(uint64_t)ri * kSahSlices + g].box[0][0][0])[i]; v = is_min ? min(v, x) : max(v, x); }
        (&s_bins.box[0][0][0])[i] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    float ext[3];
    for (int q = 0; q < 3; q++) ext[q] = r.cbox[3 + q] - r.cbox[q];
    const SahCut c = sah_select(&s_bins, ext);
    SahBigCut* out = &cuts[ri];
    const uint32_t depth = r.depth_buf >> 2;
    if (c.axis < 0) {
        // no plane separates the centres: the range goes to the one-workgroup pass as it is (it cuts the list in the middle)
        if (lane == 0) { out->axis = -1; next[atomicAdd(next_count, 1u)] = r; }
        return;
    }
    // where every slice's left and right clusters go
    uint32_t mine = 0;
    if (lane < (uint32_t)kSahSlices) for (int q = 0; q < c.split; q++) mine += gbins[(uint64_t)ri * kSahSlices + lane].cnt[c.axis][q];
    uint32_t before = mine;
    for (int off = 1; off < kSahSlices; off <<= 1) { const uint32_t o = __shfl_up(before, off); if ((int)lane >= off) before += o; }
    before -= mine;
    uint32_t s0, s1;
    sah_slice(r, min(lane, (uint32_t)kSahSlices - 1u), s0, s1);
    if (lane < (uint32_t)kSahSlices) { out->lbase[lane] = r.begin + before; out->rbase[lane] = r.begin + c.nl + (s0 - r.begin) - before; }
    if (lane == 0) {
        const uint32_t mid = r.begin + c.nl;
        const uint32_t k = mid - 1u;
        const bool root = r.parent_side == 0xffffffffu;
        if (root) *root_k = k;
        out->axis = c.axis; out->split = c.split; out->nl = c.nl; out->depth = depth;
        out->node = root ? 0u : (k == 0u ? *root_k : k);
        for (int q = 0; q < 3; q++) { out->cmin[q] = r.cbox[q]; out->ext[q] = ext[q]; }
        sah_emit(r.begin, r.end, r.parent_side, depth, mid, r.box, &c, nullptr, (r.depth_buf & 1u) ^ 1u, n, root_k, child_left, child_right, parent, bounds,
                 height, depth_out, next, next_count, small, small_count, big_next, big_next_count, false);
    }
}

__global__ __launch_bounds__(kBlock)
void k_big_move(const SahRange* __restrict__ ranges, const uint32_t* __restrict__ count, const SahBigCut* __restrict__ cuts, SahLists lists,
                uint32_t n, uint32_t* __restrict__ child_left, uint32_t* __restrict__ child_right, uint32_t* __restrict__ parent,
                const uint32_t* __restrict__ height, uint32_t* depth_out)
{
    constexpr int kWaves = kBlock / 64;
    __shared__ uint32_t s_wave[kWaves];
    const uint32_t ri = blockIdx.x / kSahSlices, g = blockIdx.x % kSahSlices;
    if (ri >= *count) return;
    const SahBigCut cut = cuts[ri];
    if (cut.axis < 0) return;
    const SahRange r = ranges[ri];
    const uint32_t from = r.depth_buf & 1u, to = from ^ 1u;
    const uint32_t* src = lists.ids[from]; const Box* src_box = lists.boxes[from];
    uint32_t* dst = lists.ids[to]; Box* dst_box = lists.boxes[to];
    uint32_t s0, s1;
    sah_slice(r, g, s0, s1);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t nr = (r.end - r.begin) - cut.nl;
    uint32_t lbase = cut.lbase[g], rbase = cut.rbase[g];
    for (uint32_t t0 = s0; t0 < s1; t0 += kBlock) {
        const uint32_t i = t0 + tid;
        const bool in = i < s1;
        uint32_t c = 0; bool goes_left = false; Box bx;
        if (in) {
            c = src[i]; bx = src_box[i];
            goes_left = sah_bin((bx.pmin[cut.axis] + bx.pmax[cut.axis]) * 0.5f, cut.cmin[cut.axis], cut.ext[cut.axis]) < cut.split;
        }
        const uint64_t lm = __ballot(in && goes_left), rm = __ballot(in && !goes_left);
        if (lane == 0u) s_wave[wave] = (uint32_t)__popcll(lm) | (uint32_t)__popcll(rm) << 16;
        __syncthreads();
        uint32_t lbefore = 0, rbefore = 0, ltotal = 0, rtotal = 0;
        for (uint32_t w = 0; w < (uint32_t)kWaves; w++) {
            const uint32_t v = s_wave[w];
            if (w < wave) { lbefore += v & 0xffffu; rbefore += v >> 16; }
            ltotal += v & 0xffffu; rtotal += v >> 16;
        }
        const uint64_t below = lane ? (~0ull >> (64u - lane)) : 0ull;
        if (in) {
            const uint32_t at = goes_left ? lbase + lbefore + (uint32_t)__popcll(lm & below) : rbase + rbefore + (uint32_t)__popcll(rm & below);
            dst[at] = c; dst_box[at] = bx;
            // a side of one cluster: its link (sah_emit left it to the thread that knows which cluster it is)
            if ((goes_left && cut.nl == 1u) || (!goes_left && nr == 1u)) {
                if (goes_left) child_left[cut.node] = c; else child_right[cut.node] = c;
                parent[c] = cut.node;
                atomicMax(depth_out, cut.depth + 1u + (c < n - 1u ? height[c] : 0u));
            }
        }
        lbase += ltotal; rbase += rtotal;
        __syncthreads();
    }
}

// the clusters' boxes next to their ids; per workgroup their box and the box of their centres (for the first range)
__global__ __launch_bounds__(kBlock)
void k_sah_begin(const uint32_t* __restrict__ cl, const uint32_t* __restrict__ state, const Box* __restrict__ bounds, SahLists lists, float* __restrict__ partial)
{
    __shared__ float s_red[kBlock / 64][12];
    const uint32_t m = state[0];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    float v[12];
    for (int q = 0; q < 12; q++) v[q] = (q % 6) < 3 ? INFINITY : -INFINITY;
    if (i < m) {
        const uint32_t c = cl[i]; const Box bx = bounds[c];
        lists.ids[0][i] = c; lists.boxes[0][i] = bx;
        for (int q = 0; q < 3; q++) { v[q] = bx.pmin[q]; v[3 + q] = bx.pmax[q]; v[6 + q] = v[9 + q] = (bx.pmin[q] + bx.pmax[q]) * 0.5f; }
    }
    for (int q = 0; q < 12; q++)
        for (int off = 32; off > 0; off >>= 1) v[q] = (q % 6) < 3 ? fminf(v[q], __shfl_xor(v[q], off)) : fmaxf(v[q], __shfl_xor(v[q], off));
    if ((threadIdx.x & 63u) == 0u) for (int q = 0; q < 12; q++) s_red[threadIdx.x >> 6][q] = v[q];
    __syncthreads();
    if (threadIdx.x < 12u) {
        const int q = (int)threadIdx.x;
        float x = s_red[0][q];
        for (int w = 1; w < kBlock / 64; w++) x = (q % 6) < 3 ? fminf(x, s_red[w][q]) : fmaxf(x, s_red[w][q]);
        partial[12u * blockIdx.x + q] = x;
    }
}
// ... and the first range: to the workgroups-per-range pass, the workgroup-per-range pass or the wavefront-per-range pass by length
__global__ __launch_bounds__(64)
void k_sah_root(const uint32_t* __restrict__ state, const float* __restrict__ partial, uint32_t blocks, SahRange* big0, uint32_t* big_count,
                SahRange* level0, uint32_t* counts, SahShort* small, uint32_t* small_count)
{
    const uint32_t m = state[0], lane = threadIdx.x;
    const uint32_t used = min(blocks, (m + kBlock - 1u) / kBlock);
    float v[12];
    for (int q = 0; q < 12; q++) v[q] = (q % 6) < 3 ? INFINITY : -INFINITY;
    for (uint32_t bk = lane; bk < used; bk += 64u)
        for (int q = 0; q < 12; q++) v[q] = (q % 6) < 3 ? fminf(v[q], partial[12u * bk + q]) : fmaxf(v[q], partial[12u * bk + q]);
    for (int q = 0; q < 12; q++)
        for (int off = 32; off > 0; off >>= 1) v[q] = (q % 6) < 3 ? fminf(v[q], __shfl_xor(v[q], off)) : fmaxf(v[q], __shfl_xor(v[q], off));
    if (lane != 0) return;
    if (m > (uint32_t)kSahSmall) {
        SahRange r; r.begin = 0; r.end = m; r.parent_side = 0xffffffffu; r.depth_buf = 2u;
        for (int q = 0; q < 6; q++) { r.box[q] = v[q]; r.cbox[q] = v[6 + q]; }
        if (m > (uint32_t)kSahBig) { big0[0] = r; big_count[0] = 1u; }
        else { level0[0] = r; counts[0] = 1u; }
    } else if (m > 1u) {
        SahShort r; r.begin = 0; r.end = m; r.parent_side = 0xffffffffu; r.depth_buf = 0u;
        small[0] = r; *small_count = 1u;
    }
}

static constexpr int kSahMaxLevels = 1024;         // counters of the level-by-level pass (a level that finds no range ends it)
static constexpr int kSahLevelBatch = 12;          // levels launched between two looks at the range counts
static constexpr int kSahBigLevels = 64;
static constexpr int kSahBigBatch = 5;

static int build_sah_top(AobBake* b, uint32_t n, const uint32_t* cl, uint32_t* other, const uint32_t* state, uint32_t bound)
{
    hipStream_t s = b->stream;
    // workspace free at this point: the level tables in refit_flags (n words), the short ranges in codes[1] (2 n words), the lists
    // with their boxes in sah_lists (2 x bound x 28 B), the long ranges' tables, bins and cuts in sah_big, counters of their own
    // (sah_counts: [0 .. kSahMaxLevels] ranges per level, short ranges, the root's cut, then the long ranges per level)
    (void)other;
    const uint32_t per_level = bound / (uint32_t)(kSahSmall + 1) + 1u;
    SahRange* table[2] = { (SahRange*)b->refit_flags, (SahRange*)b->refit_flags + per_level };
    // the two tables live in refit_flags' slice of the arena: n words rounded up to the arena's 256-byte alignment (api.hip).  They
    // fit because a range in a table holds more than kSahSmall clusters — checked here, not assumed: the slice after this one is
    // the tree depth, which sizes the traversal stacks
    static_assert(sizeof(SahRange) == 64, "the level tables of build_sah_top are sized for 64-byte SahRange records");
    {
        const uint64_t slice = (((uint64_t)n > 0 ? (uint64_t)n : 1u) * sizeof(uint32_t) + 255u) & ~(uint64_t)255u;
        if (2ull * per_level * sizeof(SahRange) > slice) {
            set_error("SAH top: two level tables of %u ranges need %llu bytes, the workspace slice holds %llu", per_level,
                      (unsigned long long)(2ull * per_level * sizeof(SahRange)), (unsigned long long)slice);
            return AOB_ERR_DEVICE;
        }
    }
    SahShort* small = (SahShort*)b->codes[1];
    uint32_t* counts = b->sah_counts;
    uint32_t* small_count = counts + kSahMaxLevels + 1, * root_k = counts + kSahMaxLevels + 2;
    uint32_t* big_counts = counts + kSahMaxLevels + 16;
    constexpr uint32_t kMaxBig = kSahTopClusters / kSahBig + 1;           // long ranges at one level
    SahRange* bigtab[2] = { (SahRange*)b->sah_big, (SahRange*)b->sah_big + kMaxBig };
    SahBigCut* cuts = (SahBigCut*)(bigtab[1] + kMaxBig);
    SahBins* gbins = (SahBins*)(cuts + kMaxBig);
    float* partial = (float*)(gbins + kMaxBig * kSahSlices);              // 12 floats per workgroup of k_sah_begin
    SahLists lists;
    const uint64_t cap = b->sah_list_capacity;
    lists.boxes[0] = (Box*)b->sah_lists; lists.boxes[1] = lists.boxes[0] + cap;
    lists.ids[0] = (uint32_t*)(lists.boxes[1] + cap); lists.ids[1] = lists.ids[0] + cap;
    if (bound > cap) { set_error("SAH top: %u clusters, room for %llu", bound, (unsigned long long)cap); return AOB_ERR_DEVICE; }
    AOB_HIP(hipMemsetAsync(counts, 0, sizeof(uint32_t) * (kSahMaxLevels + 16 + kSahBigLevels + 2), s));
    hipLaunchKernelGGL(k_sah_begin, dim3(blocks_for(bound)), dim3(kBlock), 0, s, cl, state, b->node_bounds, lists, partial);
    hipLaunchKernelGGL(k_sah_root, dim3(1), dim3(64), 0, s, state, partial, blocks_for(bound), bigtab[0], big_counts, table[0], counts, small, small_count);
    uint32_t* host = b->pinned + 12;
    // the long ranges, level by level (their shorter children collect in table[0])
    if (bound > (uint32_t)kSahBig) {
        int level = 0;
        for (;;) {
            for (int k = 0; k < kSahBigBatch && level < kSahBigLevels; k++, level++) {
                const SahRange* cur = bigtab[level & 1]; const uint32_t* cnt = big_counts + level;
                hipLaunchKernelGGL(k_big_bins, dim3(kMaxBig * kSahSlices), dim3(kBlock), 0, s, cur, cnt, lists, gbins);
                hipLaunchKernelGGL(k_big_cut, dim3(kMaxBig), dim3(64), 0, s, cur, cnt, gbins, cuts, n, root_k, b->child_left, b->child_right, b->parent, b->node_bounds,
                                   b->node_height, b->depth_dev, bigtab[(level & 1) ^ 1], big_counts + level + 1, table[0], counts, small, small_count);
                hipLaunchKernelGGL(k_big_move, dim3(kMaxBig * kSahSlices), dim3(kBlock), 0, s, cur, cnt, cuts, lists, n, b->child_left, b->child_right, b->parent,
                                   b->node_height, b->depth_dev);
            }
            AOB_HIP(hipMemcpyAsync(host, big_counts + level, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            AOB_HIP(hipStreamSynchronize(s));
            if (host[0] == 0u) break;
            if (level >= kSahBigLevels) { set_error("the SAH top needs more than %d levels of long ranges", kSahBigLevels); return AOB_ERR_BVH_TOO_DEEP; }
        }
    }
    // the others level by level, a batch of levels between two looks at the range counts (a level without ranges is an empty launch)
    int level = 0;
    for (;;) {
        for (int k = 0; k < kSahLevelBatch && level < kSahMaxLevels; k++, level++)
            hipLaunchKernelGGL(k_sah_level, dim3(per_level), dim3(kSahBlock), 0, s, table[level & 1], counts + level, lists, n, root_k, b->child_left, b->child_right,
                               b->parent, b->node_bounds, b->node_height, b->depth_dev, table[(level & 1) ^ 1], counts + level + 1, small, small_count);
        AOB_HIP(hipMemcpyAsync(host, counts + level, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        AOB_HIP(hipStreamSynchronize(s));
        if (host[0] == 0u) break;
        if (level >= kSahMaxLevels) { set_error("the SAH top needs more than %d levels", kSahMaxLevels); return AOB_ERR_BVH_TOO_DEEP; }
    }
    hipLaunchKernelGGL(k_sah_wave, dim3(bound / 2u / (kSahWaveBlock / 64) + 1u), dim3(kSahWaveBlock), 0, s, small, small_count, lists, n, root_k, b->child_left, b->child_right,
                       b->parent, b->node_bounds, b->node_height, b->depth_dev);
    return AOB_OK;
}

// prim ids written by k_world_triangles are mesh-local already (t), shape ids start at 1
// (RadeonRays/radeon_rays.h:238).


int build_bvh(AobBake* b)
{
    hipStream_t s = b->stream;
    const uint64_t n = b->triangle_count;
    b->node_count = 0; b->bvh_depth = 0; b->node4_count = 0;
    if (n == 0) { b->bvh_built = true; return AOB_OK; }
    if (n > 0x7fffffffull) { set_error("too many triangles (%llu)", (unsigned long long)n); return AOB_ERR_UNSUPPORTED; }

    hipLaunchKernelGGL(k_init_scene_bounds, dim3(1), dim3(64), 0, s, b->scene_bounds_enc);
    for (size_t mi = 0; mi < b->meshes.size(); mi++) {
        const DeviceMesh& m = b->meshes[mi];
        if (!m.triangle_count) continue;
        MeshXform xf;
        for (int k = 0; k < 16; k++) xf.m[k] = m.model[k];
        hipLaunchKernelGGL(k_world_triangles, dim3(blocks_for(m.triangle_count)), dim3(kBlock), 0, s,
                           m.positions, m.indices, m.triangle_count, xf, (uint32_t)mi + 1,
                           b->tri_verts + 3ull * m.triangle_offset, b->prim_bounds + m.triangle_offset,
                           b->scene_bounds_enc);
    }
    if (b->opt.morton_adaptive) hipLaunchKernelGGL(k_morton<1>, dim3(blocks_for(n)), dim3(kBlock), 0, s, b->prim_bounds, (uint32_t)n,
                                                   b->scene_bounds_enc, b->codes[0], b->prim_idx[0]);
    else hipLaunchKernelGGL(k_morton<0>, dim3(blocks_for(n)), dim3(kBlock), 0, s, b->prim_bounds, (uint32_t)n,
                            b->scene_bounds_enc, b->codes[0], b->prim_idx[0]);
    // stable LSD radix sort of (code, index) pairs: equal codes stay in index order
    size_t temp = b->sort_temp_bytes;
    AOB_HIP(rocprim::radix_sort_pairs(b->sort_temp, temp, b->codes[0], b->codes[1], b->prim_idx[0], b->prim_idx[1],
                                      (size_t)n, 0u, 63u, s));
    if (n > 1 && b->opt.bvh_builder >= 1) {
        AOB_HIP(hipMemsetAsync(b->depth_dev, 0, sizeof(uint32_t), s));
        hipLaunchKernelGGL(k_leaf_bounds, dim3(blocks_for(n)), dim3(kBlock), 0, s, b->prim_bounds, b->prim_idx[1],
                           (uint32_t)n, b->node_bounds);
        AOB_TRY(build_ploc(b, (uint32_t)n, b->opt.bvh_builder == 2 ? (uint32_t)kSahTopClusters : 1u));
    } else if (n > 1) {
        AOB_HIP(hipMemsetAsync(b->refit_flags, 0, sizeof(uint32_t) * (n - 1), s));
        AOB_HIP(hipMemsetAsync(b->depth_dev, 0, sizeof(uint32_t), s));
        hipLaunchKernelGGL(k_hierarchy, dim3(blocks_for(n - 1)), dim3(kBlock), 0, s, b->codes[1], (uint32_t)n,
                           b->child_left, b->child_right, b->parent);
        hipLaunchKernelGGL(k_leaf_bounds, dim3(blocks_for(n)), dim3(kBlock), 0, s, b->prim_bounds, b->prim_idx[1],
                           (uint32_t)n, b->node_bounds);
        hipLaunchKernelGGL(k_refit, dim3(blocks_for(n)), dim3(kBlock), 0, s, (uint32_t)n, b->parent, b->child_left,
                           b->child_right, b->node_bounds, b->refit_flags, b->depth_dev);
    }
    b->node_count = n == 1 ? 1 : 2 * n - 1;
    // the node arrays of the other traversal structures are written when something asks for them (a parity test, a
    // trace with those options): the default walk reads nodes4q and leaf_tris only
    b->nodes_emitted = false; b->nodes2q_emitted = false;
    // the 4-ary structure the quad-per-ray traversal walks (bvh_width 4)
    b->node4_count = n == 1 ? 1 : n - 1;
    AOB_HIP(hipMemsetAsync(b->nodes4, 0, sizeof(float4) * 8 * b->node4_count, s));
    if (n == 1) hipLaunchKernelGGL(k_single_leaf4, dim3(1), dim3(64), 0, s, b->prim_bounds, b->nodes4);
    else if (b->opt.bvh_width == 4 && b->opt.bvh_quantised)
        hipLaunchKernelGGL(k_collapse4<1>, dim3(blocks_for(n - 1)), dim3(kBlock), 0, s, (uint32_t)n, b->parent, b->child_left,
                           b->child_right, b->node_bounds, b->nodes4);
    else hipLaunchKernelGGL(k_collapse4<0>, dim3(blocks_for(n - 1)), dim3(kBlock), 0, s, (uint32_t)n, b->parent, b->child_left,
                            b->child_right, b->node_bounds, b->nodes4);
    hipLaunchKernelGGL(k_quant_frame, dim3(1), dim3(64), 0, s, b->scene_bounds_enc, b->quant_frame);
    {
        uint32_t* live = b->refit_flags;                     // free once the tree is built
        uint32_t* packed = b->packed_index;
        hipLaunchKernelGGL(k_live4, dim3(blocks_for(b->node4_count)), dim3(kBlock), 0, s, (uint32_t)b->node4_count, b->nodes4, live);
        size_t temp = b->sort_temp_bytes;
        AOB_HIP(rocprim::exclusive_scan(b->sort_temp, temp, live, packed, 0u, (size_t)b->node4_count, rocprim::plus<uint32_t>(), s));
        hipLaunchKernelGGL(k_emit_nodes4q, dim3(blocks_for(b->node4_count)), dim3(kBlock), 0, s, (uint32_t)b->node4_count, b->nodes4,
                           b->quant_frame, live, packed, b->nodes4q, b->node4q_count_dev);
        b->node4q_count = UINT64_MAX;                        // read back on demand (aob_bake_node4q_count)
    }
    hipLaunchKernelGGL(k_leaf_tris, dim3(blocks_for(n)), dim3(kBlock), 0, s, (uint32_t)n, b->prim_idx[1], b->tri_verts,
                       b->leaf_tris);
    AOB_HIP(hipGetLastError());
    b->bvh_built = true;
    if (b->opt.bvh_width != 4 && !b->opt.bvh_quantised) AOB_TRY(ensure_nodes(b));
    if (b->opt.bvh_width != 4 && b->opt.bvh_quantised) AOB_TRY(ensure_nodes2q(b));
    return AOB_OK;
}

int ensure_nodes(AobBake* b)
{
    if (b->nodes_emitted || !b->bvh_built || b->triangle_count == 0) return AOB_OK;
    hipLaunchKernelGGL(k_emit_nodes, dim3(blocks_for(b->node_count)), dim3(kBlock), 0, b->stream, (uint32_t)b->triangle_count, b->child_left,
                       b->child_right, b->node_bounds, b->prim_idx[1], b->tri_verts, (float4*)b->nodes);
    AOB_HIP(hipGetLastError());
    b->nodes_emitted = true;
    return AOB_OK;
}

int ensure_nodes2q(AobBake* b)
{
    if (b->nodes2q_emitted || !b->bvh_built || b->triangle_count < 2) return AOB_OK;
    hipLaunchKernelGGL(k_emit_nodes_q, dim3(blocks_for(b->triangle_count - 1)), dim3(kBlock), 0, b->stream, (uint32_t)b->triangle_count,
                       b->child_left, b->child_right, b->node_bounds, b->quant_frame, b->nodes2q);
    AOB_HIP(hipGetLastError());
    b->nodes2q_emitted = true;
    return AOB_OK;
}

} // namespace aob

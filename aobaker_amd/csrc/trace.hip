// trace.hip — fused ray generation + BVH traversal + scoring + accumulation (gfx950).
//
// Replaces the reference's per-chunk round trip (BakeEngine.cpp:598-675):
//   GenerateRaysComputeShader.hlsl:31-55   (48-byte ray records, numthreads(1,1,1))
//   IntersectionApi::QueryIntersection     (RadeonRays; kernels/CL/intersect_bvh2_lds.cl)
//   ProcessIntersectionsComputeShader.hlsl:52-92 (2 atomics per sample, numthreads(1,1,1))
// with ONE kernel: rays are generated in registers (no ray or hit records ever reach
// HBM), walk the 64-byte-node BVH2 with the reference's near-child-first order and a
// per-lane LDS stack (intersect_bvh2_lds.cl:78,123-153), are scored in registers, and
// each texel's result cell is written once.
//
// Work distribution.  A wavefront owns 64 consecutive surfels; their samples form a pool
// of (surfel, sample block) items — 16 blocks per surfel — that the wave's lanes draw
// from with a wave-uniform counter (__ballot + mbcnt ranking, no atomics, no LDS queue).
// A lane that finishes its block early takes the next item of ANY surfel of its wave, so
// lanes whose surfels have cheap rays do not idle behind the expensive ones.  Per-surfel
// sums live in LDS as 64-bit integers (int32-range quantised sums in the reference's mode,
// exact Q32.32 sums in float mode), which makes the result independent of which lane
// traced which block, and bit-identical to the oracle's sequential loop.
//
// Wave scheduling (the kernel is VALU-issue bound, not HBM bound — profiles/r01): one
// loop whose three sections are entered on wave-wide votes:
//   refill   idle lanes pick their next ray (and, at block ends, their next item) once
//            `refill_idle` of them are waiting;
//   leaf     lanes parked at a leaf (leaf-ness is known from the address alone: leaves
//            live at [leaf_base, 2n-1)) run Möller–Trumbore once `leaf_wait` are waiting;
//   inner    every lane at an inner node does its two slab tests each iteration.
// Keeping the leaf test out of the common iteration stops the 64 lanes from serialising
// both code paths every step.  Each ray still visits its nodes in exactly the reference's
// depth-first order, so the per-ray arithmetic — and the result — does not depend on it.
//
// The hot data is the node array (64 B per inner visit, 48 B per leaf visit, shared by
// neighbouring lanes and served from L1/L2/Infinity Cache); the per-surfel record (64 B)
// is read once per sample block and the result cell (8 B) written once per surfel.
#include "aob_internal.h"
#include "aob_math.h"

#include <cstdlib>

namespace aob {

static constexpr int kBlock = 256;
static constexpr int kBlocksPerSurfel = 16;       // sample blocks per surfel
static constexpr int kLdsStack = 16;              // traversal stack entries per lane kept in LDS
static constexpr uint32_t kInvalid = 0xffffffffu;

struct TraceParams {
    const float4* nodes;
    const uint4* nodesq;           // QUANT 1: 2 x uint4 per node (AobBvh2QNode)
    const uint4* nodes4q;          // QUANT 2: 4 x uint4 per node (AobBvh4QNode)
    const float4* leaf_tris;       // QUANT: 3 x float4 per triangle
    const float* quant_frame;      // QUANT: origin[3], scale[3]
    const float4* frames;          // 4 x float4 per surfel (k_surfel_frames)
    AobResult* results;
    const float* hemisphere;
    uint32_t* gstack;              // overflow stack in HBM: entries beyond the LDS stack
    uint64_t begin, end;
    uint32_t node_count;
    uint32_t leaf_base;            // first leaf address (n - 1)
    int32_t spp;
    int32_t block_samples;         // ceil(spp / kBlocksPerSurfel)
    uint32_t seed_hash;
    float tmax, bias, falloff;
    CosineGrid grid;
    uint32_t nblocks, blocks_per_xcd;
    int32_t refill_idle;           // idle lanes that trigger the refill section
    int32_t leaf_wait;             // lanes parked at a leaf that trigger the leaf section
};

// exact float -> Q32.32 (a float times 2^32 is exact; the conversion truncates)
__device__ __forceinline__ long long f2fix(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) v = 2147483520.0f;
    if (v <= -2147483648.0f) v = -2147483520.0f;
    return (long long)(v * 4294967296.0f);
}

// LDS_N: stack entries per lane kept in LDS; deeper entries spill to p.gstack.
// QUANT 1: walk the 32-byte BVH2 nodes with 16-bit quantised child boxes (two 16-byte fetches per visit
// instead of four) and the 48-byte leaf triangles.  QUANT 2: the same boxes in 4-ary nodes (AobBvh4QNode,
// every second level collapsed): 68 instead of 76 slab tests per ray and half the loop iterations.
#ifndef AOB_TRACE_WAVES
#define AOB_TRACE_WAVES 8          // waves per SIMD the register allocator must leave room for (<= 64 VGPRs):
                                   // measured 4.98 / 5.14 / 5.55 Grays/s at 6 / 7 / 8 — latency hiding beats the few spills
#endif
template <int SAMPLER, int ANYHIT, int ACCUM, int LDS_N, int QUANT>
__global__ __launch_bounds__(kBlock, AOB_TRACE_WAVES)
void k_trace(TraceParams p)
{
    __shared__ uint32_t lds_stack[LDS_N * kBlock];
    __shared__ unsigned long long acc[kBlock][2];       // per surfel of this block: total, weight

    // Workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8).  Give each XCD
    // a contiguous run of surfel tiles so that its private 4 MiB L2 keeps seeing the same part
    // of the atlas, hence the same part of the BVH.
    const uint32_t lb = (blockIdx.x & 7u) * p.blocks_per_xcd + (blockIdx.x >> 3);
    if (lb >= p.nblocks) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint64_t wave_first = p.begin + (uint64_t)lb * kBlock + (uint64_t)wave * 64u;   // first surfel of this wave
    const uint32_t nsurf = wave_first >= p.end ? 0u : (uint32_t)min((uint64_t)64, p.end - wave_first);
    const uint32_t total_items = nsurf ? 64u * kBlocksPerSurfel : 0u;      // item = block * 64 + surfel slot
    uint32_t item_next = 0;                                                // wave-uniform

    f3 q_origin = make_f3(0.f, 0.f, 0.f), q_scale = make_f3(0.f, 0.f, 0.f);
    if (QUANT) {
        q_origin = make_f3(p.quant_frame[0], p.quant_frame[1], p.quant_frame[2]);
        q_scale = make_f3(p.quant_frame[3], p.quant_frame[4], p.quant_frame[5]);
    }

    acc[tid][0] = 0ull; acc[tid][1] = 0ull;
    lds_stack[tid] = kInvalid;       // entry 0: the INVALID sentinel (intersect_bvh2_lds.cl:103): a pop is one read
    // overflow stack entry e of this lane lives at gstack[e * (gridDim.x*256) + blockIdx.x*256 + tid] (computed on use:
    // the path is cold and the registers are better spent on occupancy)
    auto gs_slot = [&](int e) -> uint32_t* { return p.gstack + ((uint64_t)e * gridDim.x + blockIdx.x) * kBlock + tid; };

    // current block of samples: [cur, blk_end) of surfel slot `slot`
    int32_t cur = 0, blk_end = 0;
    uint32_t slot = 0;
    bool has_partial = false;
    // block sums.  COSINE sampler: every weight is exactly 1, so the sums are counters (UNITW);
    // REFERENCE_COMPAT: 64-bit sums of the per-ray weight cW converted at ray generation.
    constexpr bool UNITW = SAMPLER == AOB_SAMPLER_COSINE;
    constexpr long long kUnit = ACCUM == AOB_ACCUMULATE_QUANTISED ? 255ll : (1ll << 32);   // ftoi(1*255) / f2fix(1)
    long long pT = 0, pW = 0, cW = 0;
    int32_t nW = 0, nMiss = 0;
    f3 P, N, X, Z;
    P = N = X = Z = make_f3(0.f, 0.f, 0.f);
    uint32_t hash_base = 0;
    const float* table = nullptr;

    bool active = false;
    f3 org, dir, invdir, oxinvdir;
    org = dir = invdir = oxinvdir = make_f3(0.f, 0.f, 0.f);
    float w = 0.0f, closest_t = 0.0f;
    uint3 rot = make_uint3(0u, 0u, 0u);        // QUANT: 16 for axes the ray travels down, else 0
    bool hit = false;
    uint32_t addr = kInvalid;
    int sp = 1;

    auto push = [&](uint32_t v) {
        if (sp < LDS_N) lds_stack[sp * kBlock + tid] = v;
        else *gs_slot(sp - LDS_N) = v;
        sp++;
    };
    auto pop = [&]() -> uint32_t {
        sp--;
        uint32_t v;
        if (sp < LDS_N) { v = lds_stack[sp * kBlock + tid]; asm volatile("" : "+v"(v)); }
        else { v = *gs_slot(sp - LDS_N); asm volatile("" : "+v"(v)); }
        return v;
    };
    // ProcessIntersectionsComputeShader.hlsl:69-90
    auto finish = [&]() {
        if (ANYHIT) {
            // r is 0 or 1: the sample adds its weight to the weight sum and, if unoccluded, to the total
            // ((1.0f * w) * 255.0f == w * 255.0f exactly).  cW was converted when the ray was made, with the
            // whole refill group active; this path runs with one or two lanes.
            if (UNITW) { nW++; if (!hit) nMiss++; }
            else { pW += cW; if (!hit) pT += cW; }
        } else {
            const float r = (hit && closest_t < p.falloff) ? closest_t / p.falloff : 1.0f;
            if (UNITW) {
                nW++;
                pT += ACCUM == AOB_ACCUMULATE_QUANTISED ? (long long)ftoi(r * 1.0f * 255.0f) : f2fix(r * 1.0f);
            } else {
                if (ACCUM == AOB_ACCUMULATE_QUANTISED) { pT += ftoi(r * w * 255.0f); pW += cW; }
                else { pT += f2fix(r * w); pW += cW; }
            }
        }
        active = false;
    };

    for (;;) {
        const bool pool_empty = item_next >= total_items;
        const bool refillable = !active && (cur < blk_end || !pool_empty);
        const uint64_t active_m = __ballot(active);
        const int n_refill = __popcll(__ballot(refillable));
        if (active_m == 0ull && n_refill == 0) break;

        // ---- refill section ----
        if (n_refill >= p.refill_idle || active_m == 0ull) {
            const bool need_item = !active && cur >= blk_end;
            if (need_item && has_partial) {
                // block finished: fold its sums into the surfel's accumulators
                atomicAdd(&acc[wave * 64u + slot][0], (unsigned long long)(UNITW && ANYHIT ? nMiss * kUnit : pT));
                atomicAdd(&acc[wave * 64u + slot][1], (unsigned long long)(UNITW ? nW * kUnit : pW));
                has_partial = false;
            }
            const uint64_t want_m = __ballot(need_item && !pool_empty);
            if (want_m != 0ull) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(want_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want_m, 0u));
                const uint32_t item = item_next + rank;
                item_next += (uint32_t)__popcll(want_m);
                if (need_item && !pool_empty && item < total_items) {
                    const uint32_t s = item & 63u, blk = item >> 6;
                    if (s < nsurf) {
                        const float4* fr = p.frames + 4ull * (wave_first + s);
                        const float4 a = fr[0], b = fr[1], c = fr[2], d = fr[3];
                        if (__float_as_uint(d.z) != 0u) {
                            P = make_f3(a.x, a.y, a.z); N = make_f3(a.w, b.x, b.y);
                            X = make_f3(b.z, b.w, c.x); Z = make_f3(c.y, c.z, c.w);
                            slot = s;
                            cur = (int32_t)blk * p.block_samples;
                            blk_end = min(p.spp, cur + p.block_samples);
                            pT = 0; pW = 0; nW = 0; nMiss = 0;
                            has_partial = cur < blk_end;
                            hash_base = pcg_hash(__float_as_uint(d.x) + p.seed_hash);
                            if (SAMPLER == AOB_SAMPLER_REFERENCE_COMPAT)
                                table = p.hemisphere + 3ull * __float_as_uint(d.y);   // Hemisphere[DTid.x * SampleCount + i]
                        }
                    }
                }
            }
            if (!active && cur < blk_end) {
                f3 h;
                if (SAMPLER == AOB_SAMPLER_REFERENCE_COMPAT) h = make_f3(table[3 * cur], table[3 * cur + 1], table[3 * cur + 2]);
                else h = cosine_sample(hash_base, cur, p.grid);
                cur++;
                dir = make_f3(fmaf(h.z, Z.x, fmaf(h.y, N.x, h.x * X.x)),
                              fmaf(h.z, Z.y, fmaf(h.y, N.y, h.x * X.y)),
                              fmaf(h.z, Z.z, fmaf(h.y, N.z, h.x * X.z)));                       // GenerateRays:46
                org = make_f3(fmaf(dir.x, p.bias, P.x), fmaf(dir.y, p.bias, P.y), fmaf(dir.z, p.bias, P.z)); // :47
                w = SAMPLER == AOB_SAMPLER_REFERENCE_COMPAT ? dot(N, dir) : 1.0f;
                if (!UNITW) cW = ACCUM == AOB_ACCUMULATE_QUANTISED ? (long long)ftoi(w * 255.0f) : f2fix(w);
                invdir = make_f3(safe_inv1(dir.x), safe_inv1(dir.y), safe_inv1(dir.z));
                oxinvdir = make_f3(-org.x * invdir.x, -org.y * invdir.y, -org.z * invdir.z);
                if (QUANT) {
                    rot = make_uint3((__float_as_uint(invdir.x) >> 31) << 4, (__float_as_uint(invdir.y) >> 31) << 4, (__float_as_uint(invdir.z) >> 31) << 4);
                    // fold the de-quantisation into the ray: (origin + q*scale - o) * invdir = q*(scale*invdir) + (origin*invdir + oxinvdir)
                    oxinvdir = make_f3(fmaf(q_origin.x, invdir.x, oxinvdir.x), fmaf(q_origin.y, invdir.y, oxinvdir.y), fmaf(q_origin.z, invdir.z, oxinvdir.z));
                    invdir = make_f3(q_scale.x * invdir.x, q_scale.y * invdir.y, q_scale.z * invdir.z);
                }
                closest_t = p.tmax; hit = false; sp = 1;
                active = true;
                addr = 0;
                if (p.node_count == 0) finish();          // empty scene: every ray misses
            }
        }

        // ---- leaf section: :162-182 (closest) / :323-339 (any hit); the triangle is inlined in the node ----
        const bool at_leaf = active && addr >= p.leaf_base;
        const bool inner = active && addr < p.leaf_base;
        const uint64_t inner_m = __ballot(inner);
        if (__popcll(__ballot(at_leaf)) >= p.leaf_wait || inner_m == 0ull) {
            if (at_leaf) {
                const float4* np = QUANT ? p.leaf_tris + 3ull * (addr - p.leaf_base) : p.nodes + 4ull * addr;
                const float4 n0 = np[0], n1 = np[1], n2 = np[2];
                const float t = intersect_triangle(org, dir, make_f3(n0.x, n0.y, n0.z), make_f3(n1.x, n1.y, n1.z),
                                                   make_f3(n2.x, n2.y, n2.z), closest_t);
                const bool closer = t < closest_t;
                if (closer) { closest_t = t; hit = true; }
                addr = (ANYHIT && closer) ? kInvalid : pop();
                if (addr == kInvalid) finish();
            }
        }

        // ---- inner section: intersect_bvh2_lds.cl:108-160 ----
        if (inner) {
          if (QUANT == 2) {
            // ---- 4-ary node: four slab tests per visit, children ranked only when the order matters ----
            const uint4* np = p.nodes4q + 4ull * addr;
            const uint4 c0 = np[0], c1 = np[1], c2 = np[2], c3 = np[3];
            auto slab = [&](const uint4& c, float& tn) -> bool {
                const uint32_t wx = __builtin_amdgcn_alignbit(c.x, c.x, rot.x), wy = __builtin_amdgcn_alignbit(c.y, c.y, rot.y),
                               wz = __builtin_amdgcn_alignbit(c.z, c.z, rot.z);
                tn = fmaxf(fmaxf(fmaxf(fmaf((float)(wx & 0xffffu), invdir.x, oxinvdir.x), fmaf((float)(wy & 0xffffu), invdir.y, oxinvdir.y)),
                                 fmaf((float)(wz & 0xffffu), invdir.z, oxinvdir.z)), 0.0f);
                const float tf = fminf(fminf(fminf(fmaf((float)(wx >> 16), invdir.x, oxinvdir.x), fmaf((float)(wy >> 16), invdir.y, oxinvdir.y)),
                                             fmaf((float)(wz >> 16), invdir.z, oxinvdir.z)), closest_t);
                return tn <= tf && c.w != kInvalid;            // an empty slot is excluded by its ref
            };
            float k0, k1, k2, k3;
            const bool h0 = slab(c0, k0), h1 = slab(c1, k1), h2 = slab(c2, k2), h3 = slab(c3, k3);
            if (!(h0 || h1 || h2 || h3)) {
                addr = pop();
                if (addr == kInvalid) finish();
            } else if (ANYHIT) {
                // any-hit: the answer does not depend on the visiting order — follow the first hit child, defer the rest
                if (h3 && (h0 || h1 || h2)) push(c3.w);
                if (h2 && (h0 || h1)) push(c2.w);
                if (h1 && h0) push(c1.w);
                addr = h0 ? c0.w : (h1 ? c1.w : (h2 ? c2.w : c3.w));
            } else {
                // closest hit: nearest child first, the others deferred farthest first, ties by child index
                // (the 4-ary form of intersect_bvh2_lds.cl:123-153; the oracle's traverse4q order)
                const float e0 = h0 ? k0 : INFINITY, e1 = h1 ? k1 : INFINITY, e2 = h2 ? k2 : INFINITY, e3 = h3 ? k3 : INFINITY;
                const int b01 = e0 <= e1, b02 = e0 <= e2, b03 = e0 <= e3, b12 = e1 <= e2, b13 = e1 <= e3, b23 = e2 <= e3;
                const int r0 = (1 - b01) + (1 - b02) + (1 - b03), r1 = b01 + (1 - b12) + (1 - b13);
                const int r2 = b02 + b12 + (1 - b23), r3 = b03 + b13 + b23;
                const int nh = (int)h0 + (int)h1 + (int)h2 + (int)h3;
                auto place = [&](bool h, int r, uint32_t ref) {
                    if (h && r > 0) {
                        const int e = sp + (nh - 1) - r;
                        if (e < LDS_N) lds_stack[e * kBlock + tid] = ref; else *gs_slot(e - LDS_N) = ref;
                    }
                };
                place(h0, r0, c0.w); place(h1, r1, c1.w); place(h2, r2, c2.w); place(h3, r3, c3.w);
                sp += nh - 1;
                addr = (h0 && r0 == 0) ? c0.w : ((h1 && r1 == 0) ? c1.w : ((h2 && r2 == 0) ? c2.w : c3.w));
            }
          } else {
            float s0x, s0y, s1x, s1y;
            uint32_t al, ar;
            if (QUANT == 1) {
                const uint4* np = p.nodesq + 2ull * addr;
                const uint4 a = np[0], b = np[1];
                // one word per axis: min | max << 16.  Rotating it by 16 for rays that travel down the axis puts the
                // NEAR plane in the low half, so min(t(min), t(max)) == t(low half) and max(..) == t(high half) —
                // exactly (t is monotonic in q) — without the six min/max per box.  invdir/oxinvdir hold the folded
                // frame here, so the slab test runs on the raw 16-bit grid values (intersect_bvh2_lds.cl:54-63).
                const uint32_t lx = __builtin_amdgcn_alignbit(a.x, a.x, rot.x), ly = __builtin_amdgcn_alignbit(a.y, a.y, rot.y),
                               lz = __builtin_amdgcn_alignbit(a.z, a.z, rot.z), rx = __builtin_amdgcn_alignbit(a.w, a.w, rot.x),
                               ry = __builtin_amdgcn_alignbit(b.x, b.x, rot.y), rz = __builtin_amdgcn_alignbit(b.y, b.y, rot.z);
                s0x = fmaxf(fmaxf(fmaxf(fmaf((float)(lx & 0xffffu), invdir.x, oxinvdir.x), fmaf((float)(ly & 0xffffu), invdir.y, oxinvdir.y)),
                                  fmaf((float)(lz & 0xffffu), invdir.z, oxinvdir.z)), 0.0f);
                s0y = fminf(fminf(fminf(fmaf((float)(lx >> 16), invdir.x, oxinvdir.x), fmaf((float)(ly >> 16), invdir.y, oxinvdir.y)),
                                  fmaf((float)(lz >> 16), invdir.z, oxinvdir.z)), closest_t);
                s1x = fmaxf(fmaxf(fmaxf(fmaf((float)(rx & 0xffffu), invdir.x, oxinvdir.x), fmaf((float)(ry & 0xffffu), invdir.y, oxinvdir.y)),
                                  fmaf((float)(rz & 0xffffu), invdir.z, oxinvdir.z)), 0.0f);
                s1y = fminf(fminf(fminf(fmaf((float)(rx >> 16), invdir.x, oxinvdir.x), fmaf((float)(ry >> 16), invdir.y, oxinvdir.y)),
                                  fmaf((float)(rz >> 16), invdir.z, oxinvdir.z)), closest_t);
                al = b.z; ar = b.w;
            } else {
                const float4* np = p.nodes + 4ull * addr;
                const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
                intersect_bbox(make_f3(n0.x, n0.y, n0.z), make_f3(n1.x, n1.y, n1.z), invdir, oxinvdir, closest_t, s0x, s0y);
                intersect_bbox(make_f3(n2.x, n2.y, n2.z), make_f3(n3.x, n3.y, n3.z), invdir, oxinvdir, closest_t, s1x, s1y);
                al = __float_as_uint(n0.w); ar = __float_as_uint(n2.w);
            }
            const bool c0 = s0x <= s0y, c1 = s1x <= s1y;
            const bool right_first = (c1 && (s0x > s1x)) || !c0;
            if (c0 && c1) push(right_first ? al : ar);
            addr = right_first ? ar : al;
            if (!(c0 || c1)) {
                addr = pop();
                if (addr == kInvalid) finish();
            }
          }
        }
    }

    // blocks still held by lanes when the pool ran dry
    if (has_partial) {
        atomicAdd(&acc[wave * 64u + slot][0], (unsigned long long)(UNITW && ANYHIT ? nMiss * kUnit : pT));
        atomicAdd(&acc[wave * 64u + slot][1], (unsigned long long)(UNITW ? nW * kUnit : pW));
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's LDS atomics have landed

    // ProcessIntersectionsComputeShader.hlsl:89-90: one surfel per cell per launch, so a plain
    // read-modify-write replaces the 2*spp InterlockedAdds
    if (lane < nsurf) {
        const float4 d = p.frames[4ull * (wave_first + lane) + 3];
        if (__float_as_uint(d.z) != 0u) {
            const long long sT = (long long)acc[tid][0], sW = (long long)acc[tid][1];
            AobResult* cell = p.results + __float_as_uint(d.x);
            AobResult c = *cell;
            if (ACCUM == AOB_ACCUMULATE_QUANTISED) {
                c.total_bits = (uint32_t)((int32_t)c.total_bits + (int32_t)sT);
                c.weight_bits = (uint32_t)((int32_t)c.weight_bits + (int32_t)sW);
            } else {
                // exact sums, rounded to float once
                c.total_bits = __float_as_uint(__uint_as_float(c.total_bits) + (float)sT * 2.3283064365386962890625e-10f);
                c.weight_bits = __float_as_uint(__uint_as_float(c.weight_bits) + (float)sW * 2.3283064365386962890625e-10f);
            }
            *cell = c;
        }
    }
}

template <int SAMPLER, int ANYHIT, int ACCUM>
static void launch_stack(int quant, dim3 grid, hipStream_t s, const TraceParams& p)
{
    if (quant == 2) hipLaunchKernelGGL((k_trace<SAMPLER, ANYHIT, ACCUM, kLdsStack, 2>), grid, dim3(kBlock), 0, s, p);
    else if (quant == 1) hipLaunchKernelGGL((k_trace<SAMPLER, ANYHIT, ACCUM, kLdsStack, 1>), grid, dim3(kBlock), 0, s, p);
    else hipLaunchKernelGGL((k_trace<SAMPLER, ANYHIT, ACCUM, kLdsStack, 0>), grid, dim3(kBlock), 0, s, p);
}
template <int SAMPLER, int ANYHIT>
static void launch_acc(int accum, int quant, dim3 grid, hipStream_t s, const TraceParams& p)
{
    if (accum == AOB_ACCUMULATE_QUANTISED) launch_stack<SAMPLER, ANYHIT, AOB_ACCUMULATE_QUANTISED>(quant, grid, s, p);
    else launch_stack<SAMPLER, ANYHIT, AOB_ACCUMULATE_FLOAT>(quant, grid, s, p);
}
template <int SAMPLER>
static void launch_hit(int anyhit, int accum, int quant, dim3 grid, hipStream_t s, const TraceParams& p)
{
    if (anyhit) launch_acc<SAMPLER, 1>(accum, quant, grid, s, p);
    else launch_acc<SAMPLER, 0>(accum, quant, grid, s, p);
}

uint64_t chunk_texel_max(const AobOptions& opt)
{
    // BakeEngine.cpp:359-364
    uint64_t spp = opt.sample_count > 0 ? (uint64_t)opt.sample_count : 1;
    uint64_t c = (uint64_t)(opt.ray_chunk_size > 0 ? opt.ray_chunk_size : 0) / spp;
    if (c > 65535) c = 65535;
    if (c < 1) c = 1;
    return c;
}

static int env_int(const char* name, int fallback)
{
    const char* v = getenv(name);
    return v && *v ? atoi(v) : fallback;
}

int trace_range(AobBake* b, int32_t material, uint64_t begin, uint64_t end)
{
    DeviceMaterial& m = b->materials[material];
    const AobOptions& o = b->opt;
    if (end > m.surfel_count) end = m.surfel_count;
    hipStream_t s = b->stream;
    AOB_HIP(hipEventRecord(b->ev_trace[0], s));
    if (begin < end && o.bvh_width == 4 && !o.bvh_quantised) {
        AOB_TRY(trace4_range(b, material, begin, end));        // quad-per-ray, 4-ary nodes (trace4.hip)
    } else if (begin < end) {
        // a lane's stack holds the sentinel plus at most one deferred sibling per tree level
        // (the 4-ary walk defers up to three children per 4-ary level)
        const uint32_t need = (o.bvh_quantised && o.bvh_width == 4) ? 3u * ((b->bvh_depth + 2) / 2) + 2u : b->bvh_depth + 2;
        // 16 stack entries per lane live in LDS (16 KB + 4 KB of sums per block = 8 blocks per CU);
        // deeper entries — rare — go to an HBM array.  24 and 32 entries measured 11 % and 23 % slower.
        const int lds_eff = kLdsStack;
        const uint32_t overflow = need > (uint32_t)lds_eff ? need - (uint32_t)lds_eff : 0u;
        TraceParams p;
        p.nodes = (const float4*)b->nodes;
        p.nodesq = b->nodes2q; p.nodes4q = b->nodes4q; p.leaf_tris = b->leaf_tris; p.quant_frame = b->quant_frame;
        const int quant = o.bvh_quantised ? (o.bvh_width == 4 ? 2 : 1) : 0;
        p.frames = m.frames; p.results = m.results;
        p.hemisphere = b->hemisphere;
        p.node_count = (uint32_t)b->node_count;
        p.leaf_base = b->triangle_count ? (uint32_t)(b->triangle_count - 1) : 0u;
        if (quant == 2) p.leaf_base = 0x80000000u;          // 4-ary refs: leaf = 0x80000000 | triangle, inner = node index
        p.spp = o.sample_count;
        p.block_samples = (o.sample_count + kBlocksPerSurfel - 1) / kBlocksPerSurfel;
        p.seed_hash = host_pcg(o.seed);
        p.tmax = o.ray_tmax;
        if (!o.use_any_hit && o.clamp_tmax_to_falloff && o.falloff_distance < p.tmax) p.tmax = o.falloff_distance;
        p.bias = o.ray_bias; p.falloff = o.falloff_distance;
        p.refill_idle = max(1, env_int("AOB_TRACE_REFILL_IDLE", 20));
        p.leaf_wait = max(1, env_int("AOB_TRACE_LEAF_WAIT", 12));
        cosine_grid(o.sample_count, p.grid);
        // one launch per section: two sections of a material may cover the same texel, and
        // their contributions are added in section order (BakeEngine.cpp:473,659)
        for (size_t si = 0; si + 1 < m.section_offsets.size(); si++) {
            uint64_t sb = m.section_offsets[si] > begin ? m.section_offsets[si] : begin;
            uint64_t se = m.section_offsets[si + 1] < end ? m.section_offsets[si + 1] : end;
            if (sb >= se) continue;
            p.begin = sb; p.end = se;
            p.nblocks = (uint32_t)((se - sb + kBlock - 1) / kBlock);
            p.blocks_per_xcd = (p.nblocks + 7) / 8;
            dim3 grid(p.blocks_per_xcd * 8);
            const uint64_t words = (uint64_t)grid.x * kBlock * (overflow ? overflow : 1u);
            if (overflow && words > b->gstack_words) {
                AOB_HIP(hipStreamSynchronize(s));
                if (b->gstack) AOB_HIP(hipFree(b->gstack));
                b->gstack = nullptr; b->gstack_words = 0;
                AOB_HIP(hipMalloc((void**)&b->gstack, sizeof(uint32_t) * words));
                b->gstack_words = words;
            }
            p.gstack = b->gstack;
            if (o.sampler == AOB_SAMPLER_REFERENCE_COMPAT) launch_hit<AOB_SAMPLER_REFERENCE_COMPAT>(o.use_any_hit, o.accumulate, quant, grid, s, p);
            else launch_hit<AOB_SAMPLER_COSINE>(o.use_any_hit, o.accumulate, quant, grid, s, p);
        }
        AOB_HIP(hipGetLastError());
    }
    AOB_HIP(hipEventRecord(b->ev_trace[1], s));
    b->trace_timed = true;
    return AOB_OK;
}

} // namespace aob

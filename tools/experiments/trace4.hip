// trace4.hip — quad-per-ray traversal of the fp32 4-ary BVH (gfx950).  A RETIRED EXPERIMENT (round 5): not compiled into
// libaobaker_hip.so, not in the parity matrices; `bvh_width 4, bvh_quantised 0` is rejected by aob_bake_create.  It was
// parity-green through round 4 and 2.7x slower than trace.hip (2.35 vs 6.4 Grays/s on BASELINE configs[2] in round 1, when it was
// last competitive) because 16 rays per wave cost 2.3x the VALU instructions per ray, and VALU issue — not L1 — is what bounds the
// lane-per-ray kernel once its nodes are quantised (the A/B record: tools/experiments/README.md).  To build it again:
// add it to SRCS in aobaker_amd/csrc/Makefile, declare trace4_range in aob_internal.h and call it from trace_range.
//
// Same fused stage as trace.hip (ray generation + traversal + scoring + accumulation, replacing
// BakeEngine.cpp:598-675 / GenerateRaysComputeShader.hlsl / QueryIntersection /
// ProcessIntersectionsComputeShader.hlsl), re-shaped around the L1 access count: in the
// lane-per-ray kernel every lane's 16-byte load is its own L1 access (0.93 accesses/clk/CU with
// fp32 nodes, profiles/r01), so a 64-byte BVH2 node costs a ray four accesses that each return a
// quarter line.
//
// Here FOUR lanes (a DPP quad) trace one ray through the 4-ary node array built by k_collapse4:
//   * lane j of the quad fetches child j's {min, ref} and {max} — the quad's four 16-byte
//     requests fall in ONE 64-byte line per load, so a node visit is 2 L1 accesses for 4 boxes
//     instead of 4 accesses for 2 boxes, and there are ~2.2x fewer visits;
//   * each lane runs one slab test (intersect_bvh2_lds.cl:54-63), the quad ranks its hits by
//     entry distance with DPP quad_perm moves, follows the nearest child and pushes the others
//     on the ray's stack in LDS (farthest first) — the 4-ary form of the reference's
//     near-child-first order (intersect_bvh2_lds.cl:123-153);
//   * leaves hold one triangle (48-byte AobLeafTri), tested with common.cl:177-218 by all four
//     lanes redundantly; ray generation and scoring are likewise replicated, so the four lanes
//     always agree and no cross-lane traffic is needed outside the node step.
// A wavefront carries 16 rays.  Work distribution (wave-level pool of 64 surfels x 16 sample
// blocks, per-surfel 64-bit sums in LDS) and the vote-gated refill / leaf / inner sections are
// those of trace.hip.
#include "aob_internal.h"
#include "aob_math.h"

#include <cstdlib>

namespace aob {

static constexpr int kBlock = 256;
static constexpr int kBlocksPerSurfel = 16;
static constexpr int kLds4 = 48;                  // stack entries per ray kept in LDS
static constexpr uint32_t kInvalid = 0xffffffffu;
static constexpr uint32_t kLeafFlag = 0x80000000u;
static constexpr unsigned long long kLeaders = 0x1111111111111111ull;   // lane 0 of every quad

struct Trace4Params {
    const float4* nodes4;          // 8 x float4 per node
    const float4* leaf_tris;       // 3 x float4 per triangle
    const float4* frames;
    AobResult* results;
    const float* hemisphere;
    uint32_t* gstack;
    uint64_t begin, end;
    uint32_t node4_count;
    int32_t spp, block_samples;
    uint32_t seed_hash;
    float tmax, bias, falloff;
    CosineGrid grid;
    uint32_t nblocks, blocks_per_xcd;
    int32_t refill_idle, leaf_wait;    // in rays (quads)
};

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __uint_as_float(dpp_u<CTRL>(__float_as_uint(v))); }

__device__ __forceinline__ long long f2fix4(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) v = 2147483520.0f;
    if (v <= -2147483648.0f) v = -2147483520.0f;
    return (long long)(v * 4294967296.0f);
}

template <int SAMPLER, int ANYHIT, int ACCUM>
__global__ __launch_bounds__(kBlock)
void k_trace4(Trace4Params p)
{
    __shared__ uint32_t lds_stack[kLds4 * 64];            // [entry][ray slot of the block]
    __shared__ unsigned long long acc[kBlock][2];         // per surfel of this block: total, weight

    const uint32_t lb = (blockIdx.x & 7u) * p.blocks_per_xcd + (blockIdx.x >> 3);   // XCD-contiguous tiles
    if (lb >= p.nblocks) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t j = lane & 3u;                          // my child slot
    const uint32_t quad_shift = lane & ~3u;
    const uint32_t rslot = wave * 16u + (lane >> 2);       // ray slot inside the block
    const uint64_t wave_first = p.begin + (uint64_t)lb * kBlock + (uint64_t)wave * 64u;
    const uint32_t nsurf = wave_first >= p.end ? 0u : (uint32_t)min((uint64_t)64, p.end - wave_first);
    const uint32_t total_items = nsurf ? 64u * kBlocksPerSurfel : 0u;
    uint32_t item_next = 0;

    acc[tid][0] = 0ull; acc[tid][1] = 0ull;
    if (j == 0) lds_stack[rslot] = kInvalid;               // entry 0: sentinel
    uint32_t* gs = p.gstack + ((uint64_t)blockIdx.x * 64u + rslot);
    const uint64_t gstride = (uint64_t)gridDim.x * 64u;

    int32_t cur = 0, blk_end = 0;
    uint32_t slot = 0;
    bool has_partial = false;
    long long pT = 0, pW = 0, cW = 0;        // block sums; this ray's weight contribution
    f3 P, N, X, Z;
    P = N = X = Z = make_f3(0.f, 0.f, 0.f);
    uint32_t hash_base = 0;
    const float* table = nullptr;

    bool active = false;
    f3 org, dir, invdir, oxinvdir;
    org = dir = invdir = oxinvdir = make_f3(0.f, 0.f, 0.f);
    float w = 0.0f, closest_t = 0.0f;
    bool hit = false;
    uint32_t node = kInvalid;
    int sp = 1;

    auto pop = [&]() -> uint32_t {
        sp--;
        uint32_t v;
        if (sp < kLds4) { v = lds_stack[sp * 64 + rslot]; asm volatile("" : "+v"(v)); }
        else { v = gs[(uint64_t)(sp - kLds4) * gstride]; asm volatile("" : "+v"(v)); }
        return v;
    };
    auto finish = [&]() {
        if (ANYHIT) {
            // r is 0 or 1: the sample adds cW to the weight and, if unoccluded, cW to the total
            // ((1.0f * w) * 255.0f == w * 255.0f exactly).  cW was converted when the ray was made,
            // with the whole refill group active; this path runs with one or two lanes.
            pW += cW;
            if (!hit) pT += cW;
        } else {
            const float r = (hit && closest_t < p.falloff) ? closest_t / p.falloff : 1.0f;
            if (ACCUM == AOB_ACCUMULATE_QUANTISED) { pT += ftoi(r * w * 255.0f); pW += cW; }
            else { pT += f2fix4(r * w); pW += cW; }
        }
        active = false;
    };

    for (;;) {
        const bool pool_empty = item_next >= total_items;
        const bool refillable = !active && (cur < blk_end || !pool_empty);
        const uint64_t active_m = __ballot(active);
        const int n_refill = __popcll(__ballot(refillable) & kLeaders);
        if (active_m == 0ull && n_refill == 0) break;

        // ---- refill section (all four lanes of a quad do the same thing) ----
        if (n_refill >= p.refill_idle || active_m == 0ull) {
            const bool need_item = !active && cur >= blk_end;
            if (need_item && has_partial) {
                if (j == 0) {
                    atomicAdd(&acc[wave * 64u + slot][0], (unsigned long long)pT);
                    atomicAdd(&acc[wave * 64u + slot][1], (unsigned long long)pW);
                }
                has_partial = false;
            }
            const uint64_t want_m = __ballot(need_item && !pool_empty) & kLeaders;
            if (want_m != 0ull) {
                const uint32_t rank = (uint32_t)__popcll(want_m & ((1ull << quad_shift) - 1ull));
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
                            pT = 0; pW = 0;
                            has_partial = cur < blk_end;
                            hash_base = pcg_hash(__float_as_uint(d.x) + p.seed_hash);
                            if (SAMPLER == AOB_SAMPLER_REFERENCE_COMPAT)
                                table = p.hemisphere + 3ull * __float_as_uint(d.y);
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
                cW = ACCUM == AOB_ACCUMULATE_QUANTISED ? (long long)ftoi(w * 255.0f) : f2fix4(w);
                invdir = make_f3(safe_inv1(dir.x), safe_inv1(dir.y), safe_inv1(dir.z));
                oxinvdir = make_f3(-org.x * invdir.x, -org.y * invdir.y, -org.z * invdir.z);
                closest_t = p.tmax; hit = false; sp = 1;
                active = true;
                node = 0;
                if (p.node4_count == 0) finish();
            }
        }

        // ---- leaf section ----
        const bool at_leaf = active && (node & kLeafFlag) != 0u;       // node != kInvalid while active
        const bool inner = active && (node & kLeafFlag) == 0u;
        const uint64_t inner_m = __ballot(inner);
        if (__popcll(__ballot(at_leaf) & kLeaders) >= p.leaf_wait || inner_m == 0ull) {
            if (at_leaf) {
                const float4* tp = p.leaf_tris + 3ull * (node & ~kLeafFlag);
                const float4 v0 = tp[0], v1 = tp[1], v2 = tp[2];
                const float t = intersect_triangle(org, dir, make_f3(v0.x, v0.y, v0.z), make_f3(v1.x, v1.y, v1.z),
                                                   make_f3(v2.x, v2.y, v2.z), closest_t);
                const bool closer = t < closest_t;
                if (closer) { closest_t = t; hit = true; }
                node = (ANYHIT && closer) ? kInvalid : pop();
                if (node == kInvalid) finish();
            }
        }

        // ---- inner section: one child box per lane ----
        if (inner) {
            const float4* np = p.nodes4 + 8ull * node;
            const float4 lo = np[j], hi = np[4 + j];
            float t0, t1;
            intersect_bbox(make_f3(lo.x, lo.y, lo.z), make_f3(hi.x, hi.y, hi.z), invdir, oxinvdir, closest_t, t0, t1);
            const uint32_t ref = __float_as_uint(lo.w);
            // (the slab test is symmetric in min/max, so an empty slot is excluded by its ref)
            const bool chit = t0 <= t1 && ref != kInvalid;
            const uint32_t qbits = (uint32_t)(__ballot(chit) >> quad_shift) & 0xfu;
            const int nhit = __popc(qbits);
            if (nhit == 0) {
                node = pop();
                if (node == kInvalid) finish();
            } else {
                // rank of my child among the quad's hits by (entry distance, child index)
                const float key = chit ? t0 : INFINITY;
                const float k0 = dpp_f<0x00>(key), k1 = dpp_f<0x55>(key), k2 = dpp_f<0xAA>(key), k3 = dpp_f<0xFF>(key);
                int rank = 0;
                rank += (k0 < key) || (k0 == key && 0u < j);
                rank += (k1 < key) || (k1 == key && 1u < j);
                rank += (k2 < key) || (k2 == key && 2u < j);
                rank += (k3 < key) || (k3 == key && 3u < j);
                // nearest child is followed, the others go on the stack farthest first
                uint32_t first = (chit && rank == 0) ? ref : 0u;
                first |= dpp_u<0xB1>(first);
                first |= dpp_u<0x4E>(first);
                if (chit && rank > 0) {
                    const int e = sp + (nhit - 1) - rank;
                    if (e < kLds4) lds_stack[e * 64 + rslot] = ref;
                    else gs[(uint64_t)(e - kLds4) * gstride] = ref;
                }
                sp += nhit - 1;
                node = first;
            }
        }
    }

    if (has_partial && j == 0) {
        atomicAdd(&acc[wave * 64u + slot][0], (unsigned long long)pT);
        atomicAdd(&acc[wave * 64u + slot][1], (unsigned long long)pW);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's LDS atomics have landed

    // one surfel per cell per launch: plain read-modify-write (ProcessIntersections:89-90)
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
                c.total_bits = __float_as_uint(__uint_as_float(c.total_bits) + (float)sT * 2.3283064365386962890625e-10f);
                c.weight_bits = __float_as_uint(__uint_as_float(c.weight_bits) + (float)sW * 2.3283064365386962890625e-10f);
            }
            *cell = c;
        }
    }
}

template <int SAMPLER, int ANYHIT>
static void launch_acc4(int accum, dim3 grid, hipStream_t s, const Trace4Params& p)
{
    if (accum == AOB_ACCUMULATE_QUANTISED) hipLaunchKernelGGL((k_trace4<SAMPLER, ANYHIT, AOB_ACCUMULATE_QUANTISED>), grid, dim3(kBlock), 0, s, p);
    else hipLaunchKernelGGL((k_trace4<SAMPLER, ANYHIT, AOB_ACCUMULATE_FLOAT>), grid, dim3(kBlock), 0, s, p);
}
template <int SAMPLER>
static void launch_hit4(int anyhit, int accum, dim3 grid, hipStream_t s, const Trace4Params& p)
{
    if (anyhit) launch_acc4<SAMPLER, 1>(accum, grid, s, p);
    else launch_acc4<SAMPLER, 0>(accum, grid, s, p);
}

int trace4_range(AobBake* b, int32_t material, uint64_t begin, uint64_t end)
{
    DeviceMaterial& m = b->materials[material];
    const AobOptions& o = b->opt;
    hipStream_t s = b->stream;
    // stack: the sentinel plus at most three deferred children per 4-ary level
    const uint32_t levels4 = (b->bvh_depth + 2) / 2;
    const uint32_t need = 3u * levels4 + 2u;
    const uint32_t overflow = need > (uint32_t)kLds4 ? need - (uint32_t)kLds4 : 0u;
    Trace4Params p;
    p.nodes4 = b->nodes4; p.leaf_tris = b->leaf_tris;
    p.frames = m.frames; p.results = m.results; p.hemisphere = b->hemisphere;
    p.node4_count = (uint32_t)b->node4_count;
    p.spp = o.sample_count;
    p.block_samples = (o.sample_count + kBlocksPerSurfel - 1) / kBlocksPerSurfel;
    p.seed_hash = host_pcg(o.seed);
    p.tmax = o.ray_tmax; p.falloff = o.falloff_distance;
    if (o.material_ranges && m.ao_max_range > 0.0f) {          // MaterialObject::AOMaxRange, as in trace.hip
        if (m.ao_max_range < p.tmax) p.tmax = m.ao_max_range;
        p.falloff = m.ao_max_range;
    }
    if (!o.use_any_hit && o.clamp_tmax_to_falloff && p.falloff < p.tmax) p.tmax = p.falloff;
    p.bias = o.ray_bias;
    // (the schedule in rays: a quad of lanes walks one ray — AobOptions' thresholds are in lanes)
    p.refill_idle = b->opt.trace_refill_idle > 0 ? max(1, b->opt.trace_refill_idle / 4) : 4;
    p.leaf_wait = b->opt.trace_leaf_wait > 0 ? max(1, b->opt.trace_leaf_wait / 4) : 4;
    cosine_grid(o.sample_count, p.grid);
    for (size_t si = 0; si + 1 < m.section_offsets.size(); si++) {
        uint64_t sb = m.section_offsets[si] > begin ? m.section_offsets[si] : begin;
        uint64_t se = m.section_offsets[si + 1] < end ? m.section_offsets[si + 1] : end;
        if (sb >= se) continue;
        p.begin = sb; p.end = se;
        p.nblocks = (uint32_t)((se - sb + kBlock - 1) / kBlock);
        p.blocks_per_xcd = (p.nblocks + 7) / 8;
        dim3 grid(p.blocks_per_xcd * 8);
        const uint64_t words = (uint64_t)grid.x * 64u * (overflow ? overflow : 1u);
        if (overflow && words > b->gstack_words) {
            AOB_HIP(hipStreamSynchronize(s));
            if (b->gstack) AOB_HIP(hipFree(b->gstack));
            b->gstack = nullptr; b->gstack_words = 0;
            AOB_HIP(hipMalloc((void**)&b->gstack, sizeof(uint32_t) * words));
            b->gstack_words = words;
        }
        p.gstack = b->gstack;
        if (o.sampler == AOB_SAMPLER_REFERENCE_COMPAT) launch_hit4<AOB_SAMPLER_REFERENCE_COMPAT>(o.use_any_hit, o.accumulate, grid, s, p);
        else launch_hit4<AOB_SAMPLER_COSINE>(o.use_any_hit, o.accumulate, grid, s, p);
    }
    AOB_HIP(hipGetLastError());
    return AOB_OK;
}

} // namespace aob

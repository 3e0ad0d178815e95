// trace.hip — fused ray generation + BVH traversal + scoring + accumulation (gfx950).
//
// Replaces the reference's per-chunk round trip (BakeEngine.cpp:598-675):
//   GenerateRaysComputeShader.hlsl:31-55   (48-byte ray records, numthreads(1,1,1))
//   IntersectionApi::QueryIntersection     (RadeonRays; kernels/CL/intersect_bvh2_lds.cl)
//   ProcessIntersectionsComputeShader.hlsl:52-92 (2 atomics per sample, numthreads(1,1,1))
// with ONE kernel in which a lane owns a surfel: it generates each of its rays in
// registers (no ray or hit records ever reach HBM), walks the 64-byte-node BVH2 with
// the reference's near-child-first order and a per-lane LDS stack
// (intersect_bvh2_lds.cl:78,123-153), scores the result and keeps the two running sums
// in registers; the texel's result cell is written once, without atomics.
//
// Wavefront-level scheduling: lanes run asynchronously through their sample lists.
// A lane whose ray has terminated (any-hit vote or empty stack) idles only until
// kRefillIdle lanes of its wave are idle, then all idle lanes generate their next ray
// together — the while-while / ray-refetch scheme, with __ballot as the vote.
//
// The hot data is the node array (64 B per visit, shared by neighbouring lanes of a
// wave and served from L1/L2/Infinity Cache); the surfel read (40 B) and the result
// write (8 B) happen once per spp rays.
#include "aob_internal.h"
#include "aob_math.h"

namespace aob {

static constexpr int kBlock = 256;
static constexpr int kLdsStack = 16;          // entries per lane kept in LDS
static constexpr int kRefillIdle = 16;        // idle lanes that trigger a wave-wide refill
static constexpr uint32_t kInvalid = 0xffffffffu;

struct TraceParams {
    const float4* nodes;
    const AobSurfel* surfels;
    const uint32_t* section_index;
    AobResult* results;
    const float* hemisphere;
    uint64_t begin, end;
    uint32_t node_count;
    uint32_t chunk_max;
    int32_t width, height;
    int32_t spp;
    uint32_t seed_hash;
    float tmax, bias, falloff;
    CosineGrid grid;
    uint32_t nblocks, blocks_per_xcd;
};

template <int SAMPLER, int ANYHIT, int ACCUM, int OVF>
__global__ __launch_bounds__(kBlock)
void k_trace(TraceParams p)
{
    __shared__ uint32_t lds_stack[kLdsStack * kBlock];
    uint32_t ovf[OVF];

    // Workgroups are dealt round-robin over the 8 XCDs (block b -> XCD b % 8).  Give each XCD
    // a contiguous run of surfel tiles so that its private 4 MiB L2 keeps seeing the same part
    // of the atlas, hence the same part of the BVH.
    const uint32_t lb = (blockIdx.x & 7u) * p.blocks_per_xcd + (blockIdx.x >> 3);
    if (lb >= p.nblocks) return;
    const uint32_t tid = threadIdx.x;
    const uint64_t k = p.begin + (uint64_t)lb * kBlock + tid;
    bool valid = k < p.end;

    AobSurfel sf;
    if (valid) sf = p.surfels[k];
    else { sf.texel_x = sf.texel_y = 0; sf.position[0] = sf.position[1] = sf.position[2] = 0.f; sf.normal[0] = sf.normal[2] = 0.f; sf.normal[1] = 1.f; }
    const f3 P = make_f3(sf.position[0], sf.position[1], sf.position[2]);
    const f3 N = make_f3(sf.normal[0], sf.normal[1], sf.normal[2]);
    valid = valid && sf.texel_x >= 0 && sf.texel_y >= 0 && sf.texel_x < p.width && sf.texel_y < p.height;
    // GenerateRaysComputeShader.hlsl:34-41
    f3 X = make_f3(1.0f, 0.0f, 0.0f), Z = make_f3(0.0f, 0.0f, 1.0f);
    if (N.y < 0.9999f && N.y > -0.9999f) {
        X = normalize(make_f3(-N.z, 0.0f, N.x));          // cross(N, (0,1,0))
        Z = normalize(cross(X, N));
    }
    // a non-finite frame gives NaN weights, and (int)NaN == 0 in HLSL: such a surfel adds nothing
    valid = valid && finite3(P) && finite3(N) && finite3(X) && finite3(Z);

    const uint32_t texel = (uint32_t)sf.texel_y * (uint32_t)p.width + (uint32_t)sf.texel_x;
    const uint32_t base = pcg_hash(texel + p.seed_hash);
    const float* table = nullptr;
    if (SAMPLER == AOB_SAMPLER_REFERENCE_COMPAT && valid) {
        // GenerateRaysComputeShader.hlsl:45: Hemisphere[DTid.x * SampleCount + i], DTid.x chunk-local
        uint64_t local = p.section_index[k];
        table = p.hemisphere + 3ull * ((local % p.chunk_max) * (uint64_t)p.spp);
    }

    int32_t next = valid ? 0 : p.spp;      // next sample to generate
    bool active = false;
    int32_t qT = 0, qW = 0;
    float fT = 0.0f, fW = 0.0f;

    f3 org, dir, invdir, oxinvdir;
    float w = 0.0f, closest_t = 0.0f;
    bool hit = false;
    uint32_t addr = kInvalid;
    int sp = 0;
    org = dir = invdir = oxinvdir = make_f3(0.f, 0.f, 0.f);

    auto push = [&](uint32_t v) {
        if (sp < kLdsStack) lds_stack[sp * kBlock + tid] = v; else ovf[sp - kLdsStack] = v;
        sp++;
    };
    auto pop = [&]() -> uint32_t {
        if (sp == 0) return kInvalid;
        sp--;
        return sp < kLdsStack ? lds_stack[sp * kBlock + tid] : ovf[sp - kLdsStack];
    };
    // ProcessIntersectionsComputeShader.hlsl:69-90
    auto finish = [&]() {
        float r;
        if (ANYHIT) r = hit ? 0.0f : 1.0f;
        else r = (hit && closest_t < p.falloff) ? closest_t / p.falloff : 1.0f;
        if (ACCUM == AOB_ACCUMULATE_QUANTISED) { qT += ftoi(r * w * 255.0f); qW += ftoi(w * 255.0f); }
        else { fT += r * w; fW += w; }
        active = false;
    };

    for (;;) {
        // ---- refill: every idle lane with samples left generates its next ray ----
        if (!active && next < p.spp) {
            f3 h;
            if (SAMPLER == AOB_SAMPLER_REFERENCE_COMPAT) h = make_f3(table[3 * next], table[3 * next + 1], table[3 * next + 2]);
            else h = cosine_sample(base, next, p.grid);
            next++;
            dir = make_f3(fmaf(h.z, Z.x, fmaf(h.y, N.x, h.x * X.x)),
                          fmaf(h.z, Z.y, fmaf(h.y, N.y, h.x * X.y)),
                          fmaf(h.z, Z.z, fmaf(h.y, N.z, h.x * X.z)));                       // :46
            org = make_f3(fmaf(dir.x, p.bias, P.x), fmaf(dir.y, p.bias, P.y), fmaf(dir.z, p.bias, P.z)); // :47
            w = SAMPLER == AOB_SAMPLER_REFERENCE_COMPAT ? dot(N, dir) : 1.0f;
            invdir = make_f3(safe_inv1(dir.x), safe_inv1(dir.y), safe_inv1(dir.z));
            oxinvdir = make_f3(-org.x * invdir.x, -org.y * invdir.y, -org.z * invdir.z);
            closest_t = p.tmax; hit = false; sp = 0;
            active = true;
            addr = 0;
            if (p.node_count == 0) finish();          // empty scene: every ray misses
        }
        if (__ballot(active) == 0ull) {
            if (__ballot(next < p.spp) == 0ull) break;   // nothing in flight, nothing left to generate
            continue;                                    // (only an empty scene gets here)
        }

        // ---- traverse until enough lanes are idle (and refillable) ----
        for (;;) {
            if (active) {
                const float4* np = p.nodes + 4ull * addr;
                const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
                if (__float_as_uint(n0.w) != kInvalid) {
                    // intersect_bvh2_lds.cl:108-160
                    float s0x, s0y, s1x, s1y;
                    intersect_bbox(make_f3(n0.x, n0.y, n0.z), make_f3(n1.x, n1.y, n1.z), invdir, oxinvdir, closest_t, s0x, s0y);
                    intersect_bbox(make_f3(n2.x, n2.y, n2.z), make_f3(n3.x, n3.y, n3.z), invdir, oxinvdir, closest_t, s1x, s1y);
                    const bool c0 = s0x <= s0y, c1 = s1x <= s1y;
                    const bool c1first = c1 && (s0x > s1x);
                    const uint32_t al = __float_as_uint(n0.w), ar = __float_as_uint(n2.w);
                    if (c0 || c1) {
                        const bool right_first = c1first || !c0;
                        addr = right_first ? ar : al;
                        if (c0 && c1) push(right_first ? al : ar);
                    } else {
                        addr = pop();
                    }
                } else {
                    // :162-182 (closest) / :323-339 (any hit)
                    const float t = intersect_triangle(org, dir, make_f3(n0.x, n0.y, n0.z), make_f3(n1.x, n1.y, n1.z),
                                                       make_f3(n2.x, n2.y, n2.z), closest_t);
                    if (t < closest_t) {
                        closest_t = t; hit = true;
                        addr = ANYHIT ? kInvalid : pop();
                    } else {
                        addr = pop();
                    }
                }
                if (addr == kInvalid) finish();
            }
            const uint64_t am = __ballot(active);
            const uint64_t refillable = __ballot(!active && next < p.spp);
            if (am == 0ull || __popcll(refillable) >= kRefillIdle) break;
        }
    }

    // ProcessIntersectionsComputeShader.hlsl:89-90: one surfel per cell per launch, so a plain
    // read-modify-write replaces the 2*spp InterlockedAdds
    if (valid) {
        AobResult* cell = p.results + texel;
        AobResult c = *cell;
        if (ACCUM == AOB_ACCUMULATE_QUANTISED) {
            c.total_bits = (uint32_t)((int32_t)c.total_bits + qT);
            c.weight_bits = (uint32_t)((int32_t)c.weight_bits + qW);
        } else {
            c.total_bits = __float_as_uint(__uint_as_float(c.total_bits) + fT);
            c.weight_bits = __float_as_uint(__uint_as_float(c.weight_bits) + fW);
        }
        *cell = c;
    }
}

template <int SAMPLER, int ANYHIT, int ACCUM>
static void launch_ovf(int ovf, dim3 grid, hipStream_t s, const TraceParams& p)
{
    if (ovf <= 48) hipLaunchKernelGGL((k_trace<SAMPLER, ANYHIT, ACCUM, 48>), grid, dim3(kBlock), 0, s, p);
    else hipLaunchKernelGGL((k_trace<SAMPLER, ANYHIT, ACCUM, 112>), grid, dim3(kBlock), 0, s, p);
}
template <int SAMPLER, int ANYHIT>
static void launch_acc(int accum, int ovf, dim3 grid, hipStream_t s, const TraceParams& p)
{
    if (accum == AOB_ACCUMULATE_QUANTISED) launch_ovf<SAMPLER, ANYHIT, AOB_ACCUMULATE_QUANTISED>(ovf, grid, s, p);
    else launch_ovf<SAMPLER, ANYHIT, AOB_ACCUMULATE_FLOAT>(ovf, grid, s, p);
}
template <int SAMPLER>
static void launch_hit(int anyhit, int accum, int ovf, dim3 grid, hipStream_t s, const TraceParams& p)
{
    if (anyhit) launch_acc<SAMPLER, 1>(accum, ovf, grid, s, p);
    else launch_acc<SAMPLER, 0>(accum, ovf, grid, s, p);
}

static void cosine_grid(int32_t spp, CosineGrid& g)
{
    int32_t a = 1;
    while ((int64_t)a * a < spp) a++;
    g.sx = a; g.sy = (spp + a - 1) / a;
    g.inv_sx = 1.0f / (float)g.sx; g.inv_sy = 1.0f / (float)g.sy;
}

static uint32_t host_pcg(uint32_t v)
{
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
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

int trace_range(AobBake* b, int32_t material, uint64_t begin, uint64_t end)
{
    DeviceMaterial& m = b->materials[material];
    const AobOptions& o = b->opt;
    if (end > m.surfel_count) end = m.surfel_count;
    hipStream_t s = b->stream;
    AOB_HIP(hipEventRecord(b->ev_trace[0], s));
    if (begin < end) {
        // the traversal stack holds at most one deferred sibling per level
        if (b->bvh_depth > (uint32_t)(kLdsStack + 112)) {
            set_error("BVH depth %u exceeds the traversal stack (%d)", b->bvh_depth, kLdsStack + 112);
            return AOB_ERR_BVH_TOO_DEEP;
        }
        const int ovf = b->bvh_depth > (uint32_t)(kLdsStack + 48) ? 112 : 48;
        TraceParams p;
        p.nodes = (const float4*)b->nodes;
        p.surfels = m.surfels; p.section_index = m.section_index; p.results = m.results;
        p.hemisphere = b->hemisphere;
        p.node_count = (uint32_t)b->node_count;
        p.chunk_max = (uint32_t)chunk_texel_max(o);
        p.width = m.width; p.height = m.height; p.spp = o.sample_count;
        p.seed_hash = host_pcg(o.seed);
        p.tmax = o.ray_tmax;
        if (!o.use_any_hit && o.clamp_tmax_to_falloff && o.falloff_distance < p.tmax) p.tmax = o.falloff_distance;
        p.bias = o.ray_bias; p.falloff = o.falloff_distance;
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
            if (o.sampler == AOB_SAMPLER_REFERENCE_COMPAT) launch_hit<AOB_SAMPLER_REFERENCE_COMPAT>(o.use_any_hit, o.accumulate, ovf, grid, s, p);
            else launch_hit<AOB_SAMPLER_COSINE>(o.use_any_hit, o.accumulate, ovf, grid, s, p);
        }
        AOB_HIP(hipGetLastError());
    }
    AOB_HIP(hipEventRecord(b->ev_trace[1], s));
    b->trace_timed = true;
    return AOB_OK;
}

} // namespace aob

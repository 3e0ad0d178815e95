// trace.hip — fused ray generation + BVH traversal + scoring + accumulation (gfx950).
//
// Replaces the reference's per-chunk round trip (BakeEngine.cpp:598-675):
//   GenerateRaysComputeShader.hlsl:31-55   (48-byte ray records, numthreads(1,1,1))
//   IntersectionApi::QueryIntersection     (RadeonRays; kernels/CL/intersect_bvh2_lds.cl)
//   ProcessIntersectionsComputeShader.hlsl:52-92 (2 atomics per sample, numthreads(1,1,1))
// with ONE kernel: rays never reach HBM.  They are generated a wavefront at a time into a
// small LDS queue, walk the BVH one lane per ray with a per-lane LDS stack
// (intersect_bvh2_lds.cl:78,123-153), are scored in registers and added to per-surfel
// LDS accumulators; each texel's result cell is written once.
//
// Work distribution.  The launch is as many one-wavefront workgroups as the chip holds; each takes
// work units off a per-XCD queue until the queues are dry.  A unit is 64 consecutive surfels of the
// trace order (an 8x8 texel tile, raster.hip) x one part of the samples.  Three sections of one loop
// are entered on wave-wide votes:
//   refill   once `refill_idle` lanes have no ray, they pop rays from the wave's queue; when
//            the queue runs dry ALL 64 lanes generate the next sample of the 64 surfels —
//            lane = surfel, every lane produces a ray — so the sampler (a hash, a sincos
//            polynomial, two square roots, three divisions) always runs at full occupancy
//            however few lanes are waiting;
//   leaf     lanes parked at a leaf (leaf-ness is known from the address alone) run
//            Möller–Trumbore once `leaf_wait` are waiting;
//   inner    every lane at an inner node does its slab tests each iteration.
// A ray is traced by whichever lane pops it; it carries its surfel's slot and adds its score
// to that surfel's LDS accumulators (int32-range quantised sums in the reference's mode, exact
// Q32.32 sums in float mode, a plain count of unoccluded samples when every weight is 1), so the
// result does not depend on which lane traced which ray and is bit-identical to the oracle's
// sequential loop.  Each ray still visits its
// nodes in exactly the reference's depth-first order.
//
// The hot data is the node array (64 B per inner visit, 48 B per leaf visit, shared by
// neighbouring lanes and served from L1/L2/Infinity Cache); the per-surfel frame (64 B) is
// re-read from L1 once per generated sample and the result cell (8 B) written once per surfel.
#include "aob_internal.h"
#include "aob_math.h"

#include <cmath>
#include <cstring>
#include <type_traits>

namespace aob {

static constexpr int kWave = 64;                  // threads per workgroup: one wavefront
// A lane's state is its `addr`: a node to visit (inner below leaf_base, a leaf from there on), or one of three values above every
// node reference — no flag registers, and every section's vote is one compare of the address.
static constexpr uint32_t kInvalid = 0xffffffffu;         // the stack's sentinel: nothing left to visit — the ray ends, unoccluded so far
static constexpr uint32_t kEndHit = 0xfffffffeu;          // any hit: the ray ended on a triangle
static constexpr uint32_t kIdle = 0xfffffffdu;            // the lane has no ray
typedef float v2f __attribute__((ext_vector_type(2)));    // a register pair for v_pk_fma_f32
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 as_uint4(v4u v) { return make_uint4(v.x, v.y, v.z, v.w); }

// Measurement hooks (tools/build_variants.py, profiles/r03/roofs/): N more vector-ALU instructions / N more 4-byte loads
// from the lane's own node per inner visit.  Compiled out (0) in the product; they never change a result.
#ifndef AOB_EXP_VALU
#define AOB_EXP_VALU 0
#endif
#ifndef AOB_EXP_LOAD
#define AOB_EXP_LOAD 0
#endif
#ifndef AOB_EXP_VALU_SLOW
#define AOB_EXP_VALU_SLOW 0        // N more v_perm_b32 (a 4.2-cycle instruction; AOB_EXP_VALU's v_xor_b32 is a 2.3-cycle one) per inner visit
#endif
#ifndef AOB_SLAB_FMA2
#define AOB_SLAB_FMA2 1
#endif
#ifndef AOB_VISIT_BRANCHLESS
#define AOB_VISIT_BRANCHLESS 1
#endif

#ifndef AOB_TRACE_CUT
#define AOB_TRACE_CUT 1            // tile cuts (k_tile_cuts + the entry phase of k_trace); 0 builds the walk from the root, for A/B runs
#endif
#ifndef AOB_CUT_WEDGES
#define AOB_CUT_WEDGES 1           // 0: every unit tests every entry of its tile's cut (A/B runs)
#endif
#ifndef AOB_CUT_LAZY
#define AOB_CUT_LAZY 1              // 1: a ray keeps the entries it has still to walk as a mask and takes the next when its stack runs dry; 0: it pushes them all when it starts
#endif
#ifndef AOB_LEAF_BOTH
#define AOB_LEAF_BOTH 1             // any hit: both triangles of a leaf pair fetched in one go (one memory round trip per leaf section instead of two)
#endif
#ifndef AOB_LEAF_HELPERS
#define AOB_LEAF_HELPERS 1          // leaf pairs (QUANT 2): the second triangle of a parked lane's pair is tested by a lane that is NOT parked at a leaf
                                    // — one Möller–Trumbore pass per leaf section instead of two (round 5); 0: the parked lane tests both itself; 2: any-hit rays only
#endif
#ifndef AOB_CUT_SHRINK_PCT
#define AOB_CUT_SHRINK_PCT 25        // k_tile_cuts opens the slots that meet the tile's box PULLED IN by this share of its extent on every side
#endif
#ifndef AOB_GEN_ORDER
#define AOB_GEN_ORDER 1             // 1: the COSINE sampler's units generate their samples from the last to the first; 0: in index order
#endif
#ifndef AOB_CUT_SCALAR
#define AOB_CUT_SCALAR 0           // CUT mode: scalar steps after a refill, from the node its first ray starts on (measured: slower)
#endif

#ifndef AOB_TRACE_STATS
#define AOB_TRACE_STATS 0          // diagnostic build (tools/build_variants.py stats=-DAOB_TRACE_STATS=1): loop counters in g_trace_stats
#endif
#if AOB_TRACE_STATS
__device__ unsigned long long g_trace_stats[16];
#define AOB_STAT(i, v) do { if (lane == 0) st_[i] += (unsigned long long)(v); } while (0)
#else
#define AOB_STAT(i, v) do { } while (0)
#endif

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
    uint32_t seed_hash;
    float tmax, bias, falloff;
    float inv_tmax;                // 1 / tmax (QUANT 2, any hit: box distances are walked in units of tmax)
    CosineGrid grid;
    // work units: a unit is one tile (64 surfels) x one part of the samples
    uint32_t ntiles;               // wavefront-sized tiles in [begin, end)
    uint32_t parts;                // sample parts per tile (1: the workgroup owns its result cells; > 1: it adds to them atomically)
    uint32_t chunk;                // tiles per chunk; whole chunks are dealt to the 8 XCDs in turn
    uint32_t xcd_chunk_tiles;      // tiles every XCD gets from the whole groups of 8 chunks
    uint32_t full_tiles;           // tiles in those groups; the rest is split into 8 runs of rest_per_xcd
    uint32_t rest_per_xcd;
    uint32_t band_row;             // fully covered atlas: tiles per tile row (chunk = a band of 8 tile rows), else 0
    uint32_t units_per_xcd;        // coarse_units + fine tiles * parts_fine
    uint32_t coarse_units;         // the first units of a queue: coarse_tiles tiles x parts
    uint32_t coarse_tiles;
    uint32_t parts_fine;           // the last tiles of every queue are cut into this many parts: a shorter tail of idle CUs at the end
    uint32_t* unit_next;           // [8] next unit of each XCD's queue (zeroed before the launch)
    unsigned long long* acc_cells; // parts > 1 with float cells that hold fractions: the parts' exact sums meet here (2 per cell), k_acc_to_cells rounds them once
    // MASKED: section visibility (AOB_MESH_*)
    const uint32_t* hidden_from_others;    // bit s: the triangles of shape s are ignored by rays of other shapes
    uint32_t self_shape;           // shape id of the section whose surfels this launch traces (mesh index + 1)
    uint32_t self_hidden;          // ... and whether its rays ignore its own triangles
    int32_t refill_idle;           // lanes without a ray that trigger the refill section
    int32_t leaf_wait;             // lanes parked at a leaf that trigger the leaf section
    uint32_t scalar_min;           // lanes on one node that make a scalar step (refill section) worth its pass
    const uint32_t* tile_cuts;     // CUT: kCutWords words per tile of [begin, end) (k_tile_cuts), or null
    uint32_t cut_records;          // CUT: records per tile — one per sample part (coarse parts first, then the fine ones), or 1
    const uint32_t* horizon;       // COSINE sampler, QUANT 2: the surfels' horizon tables (horizon.hip: free columns of every azimuth row, a byte each), or null
    uint32_t horizon_words;        // ... words per surfel (4 or 8)
    unsigned long long* cast_total;   // ... the samples the launch generated (a sum per wavefront, added when it ends)
};

// ---- tile cuts (CUT mode of k_trace) ----
// The rays of one tile (64 neighbouring surfels) all start inside one chain of nodes, and each of them tests that chain's boxes
// on its way down only to find itself inside them.  A CUT of the tree for a tile is a list of child slots {ref, box} that every
// root-to-leaf path crosses exactly once: breadth first from the root's four slots, every inner slot whose box meets the tile's
// box (the surfels' positions, pulled in by a quarter of its extent) is replaced by its own slots while the list fits kCutMax entries — what is left
// is the chain's SIBLINGS at every level plus whatever the chain ends in.  A new ray tests the boxes of all entries when it is
// GENERATED — all 64 lanes busy, the boxes in scalar registers, no stack, no vector memory — and starts its walk below the
// entries it entered (a 32-bit mask in its queue record).  Child boxes are nested in their parent's on the quantisation grid and
// the slab test is monotone in the planes (an fma with the ray's constants), so a slot whose ancestor fails the test fails it
// too: the triangles a ray reaches are exactly those of the walk from the root, whatever the cut (the oracle's orc_tile_cut /
// orc_cut_walk count it; tools/measure_tile_cuts.py).  Any hit only: the order of the walk differs, the answer cannot.
static constexpr int kCutMax = 32;
static constexpr int kCutWedges = 16;              // sample parts (coarse + fine) a tile's entries are culled for
static constexpr int kCutLines = 8;                // distinct boundaries of those parts' azimuth wedges
static constexpr int kCutWords = 4 * kCutMax;      // per record: ref[kCutMax], then {x, y, z plane words}[kCutMax] (a box as in AobBvh4QNode);
                                                   // a tile has one record per sample part (CutWedges::count; one in all without culling)
static constexpr int kCutList = 128;               // working list of k_tile_cuts (4 + 4 per expansion)
static constexpr uint32_t kEmptyPlanes = 0x8000ffffu;   // an empty slot's inverted box (bvh_build.hip k_emit_nodes4q); here: an unused entry

// Which entries of its tile's cut can a UNIT's rays reach at all?  A unit traces one part of the samples: with the COSINE
// sampler's stratification that is a range of rows of the grid, i.e. of the azimuth in every surfel's own frame — rays of part
// p of 4 leave a surfel inside one quadrant of its tangent plane.  Such a ray stays in the cone {P + a u0 + b N + c u1 : a, b, c
// >= 0} (u0, u1 the wedge's two boundary directions in the tangent plane): an entry whose box — grown by a margin far above the rounding of
// any of this — lies wholly on the negative side of one of the three planes for EVERY surfel of the tile cannot be entered
// by any ray of the unit, and the entry phase of k_trace skips it for the whole unit (a third of the entries on BASELINE
// configs[2]; tests/test_gpu_parity.py::test_tile_cuts_match_the_oracle_and_every_part_record_reaches_what_the_root_walk_reaches checks
// that the leaf slots a part's rays reach from its record are exactly those they reach from the root).
struct CutWedges {
    int32_t count;                          // wedges = records per tile (0: no culling, one record with every entry)
    int32_t lines;                          // distinct boundary lines
    float line_s[kCutLines], line_c[kCutLines];      // boundary line l in the surfel's tangent plane: direction c X + s Z (its plane through N: normal (c X + s Z) x N)
    uint32_t need_pos[kCutLines];           // wedges whose cone lies on the positive side of line l ...
    uint32_t need_neg[kCutLines];           // ... on the negative side
};

__global__ __launch_bounds__(64)
void k_tile_cuts(const float4* __restrict__ frames, uint64_t begin, uint64_t end, const uint4* __restrict__ nodes4q,
                 const float* __restrict__ quant_frame, uint32_t node_count, float shrink, int max_entries, float reach2, CutWedges W,
                 uint32_t* __restrict__ out)
{
    __shared__ uint4 list[kCutList];
    __shared__ uint32_t dead[kCutList];
    __shared__ uint4 fin[kCutMax];
    const uint32_t lane = threadIdx.x;
    const uint64_t k = begin + (uint64_t)blockIdx.x * 64u + lane;       // one wavefront per tile, lane = surfel
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    bool valid = false;
    f3 P = make_f3(0.f, 0.f, 0.f), N = P, X = P, Z = P;
    if (k < end) {
        const float4 a = frames[4ull * k], b = frames[4ull * k + 1], c = frames[4ull * k + 2], d = frames[4ull * k + 3];
        valid = __float_as_uint(d.z) != 0u;
        P = make_f3(a.x, a.y, a.z); N = make_f3(a.w, b.x, b.y); X = make_f3(b.z, b.w, c.x); Z = make_f3(c.y, c.z, c.w);
        if (valid) { lo[0] = hi[0] = a.x; lo[1] = hi[1] = a.y; lo[2] = hi[2] = a.z; }
    }
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { lo[q] = fminf(lo[q], __shfl_xor(lo[q], off)); hi[q] = fmaxf(hi[q], __shfl_xor(hi[q], off)); }
    // the tile's box in units of the quantisation grid (a stored plane q sits at origin + q * scale)
    float glo[3], ghi[3], qo[3], qs[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const float o = quant_frame[q], sc = quant_frame[3 + q];
        qo[q] = o; qs[q] = sc;
        // Which slots to open is a choice, not a condition (any cut gives the same answers): the box of the surfels' positions pulled in by a
        // quarter of its extent on every side opens the chain MOST of the tile's rays start in and stops opening both sides of every plane
        // the tile merely straddles — the entries go deeper: 6.9 -> 6.1 node visits per ray below the cut on BASELINE configs[2] against the
        // box grown by the ray bias, 3.4 -> 3.1 on configs[1] (tools/measure_tile_cuts.py --shrink; 0 / 0.125 / 0.25 / 0.375 / 0.5 -> 6.42 / 6.12 /
        // 6.07 / 6.08 / 6.13 and 3.43 / 3.17 / 3.06 / 3.18 / 3.47).
        const float pull = shrink * (hi[q] - lo[q]);
        glo[q] = sc > 0.0f ? ((lo[q] + pull) - o) / sc : -INFINITY;
        ghi[q] = sc > 0.0f ? ((hi[q] - pull) - o) / sc : INFINITY;
    }
    // Rays of bounded REACH (round 5: closest-hit scoring with tmax clamped to the falloff distance, a material range): the tile's rays start at
    // the surfels' positions — inside the box [lo, hi] — and end within `reach` of them, so a slot whose box lies farther than that from the box
    // holds no triangle any of them can hit.  Such a slot never joins the list (not when the root's slots start it, not when an opened entry's
    // slots are appended), and an opened entry counts only the slots it appends against the budget: the cut spends its entries on the part of
    // the tree the rays can reach.  The distance in float, exactly as the oracle's cut_within_reach computes it (orc_tile_cut_reach).
    const float flo[3] = { lo[0], lo[1], lo[2] }, fhi[3] = { hi[0], hi[1], hi[2] };
    auto kept = [&](const uint4& c) -> bool {
        if (c.x == kEmptyPlanes && c.y == kEmptyPlanes && c.z == kEmptyPlanes) return false;
        if (!(reach2 > 0.0f)) return true;
        const uint32_t w[3] = { c.x, c.y, c.z };
        float d2 = 0.0f;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const float bmin = fmaf((float)(w[q] & 0x7fffu), qs[q], qo[q]), bmax = fmaf((float)((w[q] >> 16) & 0x7fffu), qs[q], qo[q]);
            const float gap = fmaxf(fmaxf(bmin - fhi[q], flo[q] - bmax), 0.0f);
            d2 = fmaf(gap, gap, d2);
        }
        return d2 <= reach2;
    };
    // Breadth first, a level at a time: every lane looks at one entry of the level — an inner slot whose box meets the tile's box is
    // a candidate and fetches its node — then the candidates are admitted IN LIST ORDER while the cut still fits (the same cut as
    // opening them one by one, orc_tile_cut: a candidate's children go to the end of the list, behind the whole level), and the
    // admitted ones append their slots.  One memory round trip per level instead of one per opened node.
    int n = 0, live = 0;
    if (node_count) {
        {
            uint4 c = make_uint4(0u, 0u, 0u, 0u);
            bool there = false;
            if (lane < 4u) { c = nodes4q[lane]; there = kept(c); }
            const uint64_t m0 = __ballot(there);
            if (there) { const int at = __popcll(m0 & ((1ull << lane) - 1ull)); list[at] = make_uint4(c.w, c.x, c.y, c.z); dead[at] = 0u; }
            n = live = __popcll(m0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        for (int lo_e = 0; lo_e < n; ) {
            const int hi_e = n;
            for (int base = lo_e; base < hi_e; base += 64) {
                const int e = base + (int)lane;
                uint4 en = make_uint4(0x80000000u, 0u, 0u, 0u);
                if (e < hi_e) en = list[e];
                bool cand = e < hi_e && !(en.x & 0x80000000u);              // not a leaf or a pair
                if (cand) {
                    const float x0 = (float)(en.y & 0x7fffu), x1 = (float)((en.y >> 16) & 0x7fffu), y0 = (float)(en.z & 0x7fffu), y1 = (float)((en.z >> 16) & 0x7fffu),
                                z0 = (float)(en.w & 0x7fffu), z1 = (float)((en.w >> 16) & 0x7fffu);
                    cand = x0 <= ghi[0] && x1 >= glo[0] && y0 <= ghi[1] && y1 >= glo[1] && z0 <= ghi[2] && z1 >= glo[2];
                }
                uint4 k0 = make_uint4(0u, 0u, 0u, 0u), k1 = k0, k2 = k0, k3 = k0;
                uint32_t present = 0u;
                if (cand) {
                    const uint4* np = nodes4q + 4ull * en.x;
                    k0 = np[0]; k1 = np[1]; k2 = np[2]; k3 = np[3];
                    present = (kept(k0) ? 1u : 0u) | (kept(k1) ? 2u : 0u) | (kept(k2) ? 4u : 0u) | (kept(k3) ? 8u : 0u);
                }
                const int kids = __popc(present);
                // admit in list order (wave-uniform loop over the candidates)
                int my_at = -1;
                for (uint64_t cm = __ballot(cand); cm; cm &= cm - 1ull) {
                    const int l = (int)__builtin_ctzll(cm);
                    const int kl = __builtin_amdgcn_readlane(kids, l);
                    if (live - 1 + kl > max_entries || n + kl > kCutList) continue;
                    if ((int)lane == l) my_at = n;
                    n += kl; live += kl - 1;
                }
                if (my_at >= 0) {
                    dead[e] = 1u;
                    int at = my_at;
                    if (present & 1u) { list[at] = make_uint4(k0.w, k0.x, k0.y, k0.z); dead[at] = 0u; at++; }
                    if (present & 2u) { list[at] = make_uint4(k1.w, k1.x, k1.y, k1.z); dead[at] = 0u; at++; }
                    if (present & 4u) { list[at] = make_uint4(k2.w, k2.x, k2.y, k2.z); dead[at] = 0u; at++; }
                    if (present & 8u) { list[at] = make_uint4(k3.w, k3.x, k3.y, k3.z); dead[at] = 0u; at++; }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
            lo_e = hi_e;
        }
    }
    // compact in list order; unused entries hold an empty slot's planes
    int m = 0;
    if (lane < (uint32_t)kCutMax) fin[lane] = make_uint4(0x80000000u, kEmptyPlanes, kEmptyPlanes, kEmptyPlanes);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int base = 0; base < n; base += 64) {
        const int e = base + (int)lane;
        const bool keep = e < n && dead[e] == 0u;
        const uint64_t km = __ballot(keep);
        if (keep) fin[m + __popcll(km & ((1ull << lane) - 1ull))] = list[e];
        m += __popcll(km);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();

    // ---- the entries each part of the samples can reach (lane = surfel; lane w < count ends up with part w's mask) ----
    const uint32_t every = m >= 32 ? 0xffffffffu : (1u << m) - 1u;
    uint32_t mine = every;
    if (W.count > 0 && m > 0) {
        const uint32_t all_w = (1u << W.count) - 1u;
        // a margin that dwarfs the rounding of the frames, of the sampler's sine and cosine and of the slab test itself
        const float eps = 1e-4f * 32767.0f * fmaxf(fmaxf(qs[0], qs[1]), qs[2]);
        // the planes that bound a part's cone of rays {a u0 + b N + c u1 : a, b, c >= 0} (u = cos X + sin Z at the wedge's two boundary
        // azimuths): through N and a boundary direction — normal u x N, = -sin X + cos Z in an orthonormal frame, and exact also where
        // the frame is not (normals within 0.0001 of the y axis keep X and Z on the axes, k_surfel_frames) — and the tangent plane,
        // Z x X; all of them times the frame's handedness (X, N, Z is left-handed for those axis frames under a down-facing normal)
        const float det = dot(cross(X, N), Z);
        const float hand = det > 0.0f ? 1.0f : -1.0f;
        const bool framed = valid && det != 0.0f && det == det;
        f3 ln[kCutLines];
#pragma unroll
        for (int l = 0; l < kCutLines; l++) {
            const f3 c = cross(make_f3(fmaf(W.line_s[l], Z.x, W.line_c[l] * X.x), fmaf(W.line_s[l], Z.y, W.line_c[l] * X.y), fmaf(W.line_s[l], Z.z, W.line_c[l] * X.z)), N);
            ln[l] = make_f3(c.x * hand, c.y * hand, c.z * hand);
        }
        const f3 zx = cross(Z, X);
        const f3 up = make_f3(zx.x * hand, zx.y * hand, zx.z * hand);
        mine = 0u;
        for (int e = 0; e < m; e++) {
            const uint4 en = fin[e];
            const f3 r0 = make_f3((fmaf((float)(en.y & 0x7fffu), qs[0], qo[0]) - eps) - P.x, (fmaf((float)(en.z & 0x7fffu), qs[1], qo[1]) - eps) - P.y,
                                  (fmaf((float)(en.w & 0x7fffu), qs[2], qo[2]) - eps) - P.z);
            const f3 r1 = make_f3((fmaf((float)((en.y >> 16) & 0x7fffu), qs[0], qo[0]) + eps) - P.x, (fmaf((float)((en.z >> 16) & 0x7fffu), qs[1], qo[1]) + eps) - P.y,
                                  (fmaf((float)((en.w >> 16) & 0x7fffu), qs[2], qo[2]) + eps) - P.z);
            // the largest and the smallest value of (c - P) . v over the box
            auto most = [&](const f3& v) { return fmaxf(r0.x * v.x, r1.x * v.x) + fmaxf(r0.y * v.y, r1.y * v.y) + fmaxf(r0.z * v.z, r1.z * v.z); };
            auto least = [&](const f3& v) { return fminf(r0.x * v.x, r1.x * v.x) + fminf(r0.y * v.y, r1.y * v.y) + fminf(r0.z * v.z, r1.z * v.z); };
            uint32_t ok = 0u;
            if (valid) {
                ok = all_w;
                if (framed) {
                    if (!(most(up) >= 0.0f)) ok = 0u;                   // wholly below the surfel's tangent plane: no part reaches it
#pragma unroll
                    for (int l = 0; l < kCutLines; l++) {
                        if (l < W.lines) {
                            if (!(most(ln[l]) >= 0.0f)) ok &= ~W.need_pos[l];
                            if (!(least(ln[l]) <= 0.0f)) ok &= ~W.need_neg[l];
                        }
                    }
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) ok |= __shfl_xor(ok, off);
            if (lane < (uint32_t)W.count && ((ok >> lane) & 1u)) mine |= 1u << e;
        }
    }
    // one record per part: its live entries in list order, then empty slots
    const int records = W.count > 0 ? W.count : 1;
    for (int w = 0; w < records; w++) {
        const uint32_t lv = W.count > 0 ? (uint32_t)__builtin_amdgcn_readlane((int)mine, w) & every : every;
        uint32_t* dst = out + ((uint64_t)blockIdx.x * records + w) * kCutWords;
        if (lane < (uint32_t)kCutMax) {
            const bool in = ((lv >> lane) & 1u) != 0u;
            const int at = in ? __popc(lv & ((1u << lane) - 1u)) : __popc(lv) + __popc(~lv & ((1u << lane) - 1u));
            const uint4 en = in ? fin[lane] : make_uint4(0x80000000u, kEmptyPlanes, kEmptyPlanes, kEmptyPlanes);
            dst[at] = en.x; dst[kCutMax + 3 * at] = en.y; dst[kCutMax + 3 * at + 1] = en.z; dst[kCutMax + 3 * at + 2] = en.w;
        }
    }
}

// One slab test of the any-hit walk over quantised 4-ary nodes: two v_perm_b32 per axis build the near and the far plane's float
// (byte selectors per ray: min for rays that travel up the axis, max for rays that travel down it), (t_near, t_far) = (f_near, f_far)
// * invdir + oxinvdir in units of tmax: the ray's span is [0, 1].  A box wholly beyond 1 passes (its near distance clamps to 1 <=
// far) — harmless, the triangle test has the last word (the oracle's traverse4q does the same).
__device__ __forceinline__ bool slab4q_anyhit(const uint4& c, const f3& invdir, const f3& oxinvdir, const uint3& nsel, const uint3& fsel, float& tn)
{
    constexpr uint32_t kHalf = 0x3f000000u;
    const float nx = __uint_as_float(__builtin_amdgcn_perm(kHalf, c.x, nsel.x)), fx = __uint_as_float(__builtin_amdgcn_perm(kHalf, c.x, fsel.x));
    const float ny = __uint_as_float(__builtin_amdgcn_perm(kHalf, c.y, nsel.y)), fy = __uint_as_float(__builtin_amdgcn_perm(kHalf, c.y, fsel.y));
    const float nz = __uint_as_float(__builtin_amdgcn_perm(kHalf, c.z, nsel.z)), fz = __uint_as_float(__builtin_amdgcn_perm(kHalf, c.z, fsel.z));
#if AOB_SLAB_FMA2
    // Six v_fma_f32, not three v_pk_fma_f32: a plain FMA issues in ~2.3 cycles and overlaps the ~4.2-cycle v_perm_b32 beside it; the
    // packed one is a 4.2-cycle instruction itself (tools/valu_probe.hip, profiles/r03/roofs/valu_probe.jsonl): -3.9 % on BASELINE
    // configs[2].  (The closest-hit walk, full of compares and selects already, loses 3 % with it and keeps the packed form.)
    // Inline asm: the SLP vectoriser would pack them again.
    auto fma1 = [](float a, float b, float cc) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(cc)); return r; };
    const float tnx = fma1(nx, invdir.x, oxinvdir.x), tfx = fma1(fx, invdir.x, oxinvdir.x);
    const float tny = fma1(ny, invdir.y, oxinvdir.y), tfy = fma1(fy, invdir.y, oxinvdir.y);
    const float tnz = fma1(nz, invdir.z, oxinvdir.z), tfz = fma1(fz, invdir.z, oxinvdir.z);
#else
    const v2f tx = __builtin_elementwise_fma((v2f){ nx, fx }, (v2f){ invdir.x, invdir.x }, (v2f){ oxinvdir.x, oxinvdir.x });
    const v2f ty = __builtin_elementwise_fma((v2f){ ny, fy }, (v2f){ invdir.y, invdir.y }, (v2f){ oxinvdir.y, oxinvdir.y });
    const v2f tz = __builtin_elementwise_fma((v2f){ nz, fz }, (v2f){ invdir.z, invdir.z }, (v2f){ oxinvdir.z, oxinvdir.z });
    const float tnx = tx.x, tfx = tx.y, tny = ty.x, tfy = ty.y, tnz = tz.x, tfz = tz.y;
#endif
    tn = fminf(fmaxf(fmaxf(fmaxf(tnx, tny), tnz), 0.0f), 1.0f);
    return tn <= fminf(fminf(tfx, tfy), tfz);
}

// ... and of the closest-hit walk (intersect_bvh2_lds.cl:54-63 on the plane floats), for the entry phase of a ray that has no hit yet:
// near clamped at 0, far at the ray's span.  The expressions of visit4's closest-hit slab, on plain FMAs (the same values).
__device__ __forceinline__ bool slab4q_closest(const uint4& c, const f3& invdir, const f3& oxinvdir, const uint3& nsel, const uint3& fsel, float bound)
{
    constexpr uint32_t kHalf = 0x3f000000u;
    const float nx = __uint_as_float(__builtin_amdgcn_perm(kHalf, c.x, nsel.x)), fx = __uint_as_float(__builtin_amdgcn_perm(kHalf, c.x, fsel.x));
    const float ny = __uint_as_float(__builtin_amdgcn_perm(kHalf, c.y, nsel.y)), fy = __uint_as_float(__builtin_amdgcn_perm(kHalf, c.y, fsel.y));
    const float nz = __uint_as_float(__builtin_amdgcn_perm(kHalf, c.z, nsel.z)), fz = __uint_as_float(__builtin_amdgcn_perm(kHalf, c.z, fsel.z));
    const float tnx = fmaf(nx, invdir.x, oxinvdir.x), tfx = fmaf(fx, invdir.x, oxinvdir.x);
    const float tny = fmaf(ny, invdir.y, oxinvdir.y), tfy = fmaf(fy, invdir.y, oxinvdir.y);
    const float tnz = fmaf(nz, invdir.z, oxinvdir.z), tfz = fmaf(fz, invdir.z, oxinvdir.z);
    const float tn = fmaxf(fmaxf(fmaxf(tnx, tny), tnz), 0.0f);
    return tn <= fminf(fminf(fminf(tfx, tfy), tfz), bound);
}

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
// every second level collapsed): fewer slab tests per ray and half the loop iterations.
#ifndef AOB_TRACE_WAVES
#define AOB_TRACE_WAVES 8          // waves per SIMD the register allocator must leave room for (<= 64 VGPRs): the walk
                                   // lives on latency hiding — 27 instead of 32 wavefronts per CU cost 11 %
#endif
#ifndef AOB_TRACE_WAVES_HZ
#define AOB_TRACE_WAVES_HZ 7       // ... and for the launches under horizon tables: their rays are the long walks near the horizon, fewer and heavier — 72 registers and
                                   // 7 wavefronts per SIMD beat 64 and 8 there (BASELINE configs[2]: 36.1 -> 33.6 ms, 6 wavefronts 34.9; closest-hit terrain 65.2 -> 62.7;
                                   // the launches without tables are even at 7 and lose at 6: profiles/r06/horizon/waves_per_simd.jsonl)
#endif
template <int SAMPLER, int ANYHIT, int ACCUM, int LDS_N, int QUANT, int MASKED, int CUTMODE, int HZMODE>
__global__ __launch_bounds__(kWave, (HZMODE ? AOB_TRACE_WAVES_HZ : AOB_TRACE_WAVES))
void k_trace(TraceParams p)
{
    // 5 KB per wavefront: 32 wavefronts per CU fill its 160 KB
    __shared__ uint32_t lds_stack[LDS_N * kWave];
    // per surfel: total, weight as 64-bit sums — or, with unit weights and any-hit scoring, one 32-bit count of the
    // unoccluded samples (the weight sum is the sample count); the 768 B saved hold three more stack entries per lane
    constexpr bool COUNT32 = SAMPLER == AOB_SAMPLER_COSINE && ANYHIT;
    // (unit weights with falloff scoring: only the totals — the weight sum is the sample count — and two more stack entries)
    constexpr bool TOTAL_ONLY = SAMPLER == AOB_SAMPLER_COSINE && !ANYHIT;
    // (QUANTISED sums are the reference's int32 cells: 4-byte sums are the same arithmetic, and free two or three entries more)
    typedef typename std::conditional<ACCUM == AOB_ACCUMULATE_QUANTISED, unsigned int, unsigned long long>::type acc_t;
    __shared__ acc_t acc[COUNT32 ? 1 : kWave][TOTAL_ONLY ? 1 : 2];
    // CUT: the rays start below their tile's cut (k_tile_cuts).  Unit weights leave the queue record's weight word to the mask of
    // the entries a ray entered; the miss counters shrink to 16 bits (a part is at most 65 535 samples: trace_range) and the
    // 128 bytes they free hold the entries' refs.
    constexpr bool CUT = CUTMODE != 0;
    static_assert(!CUT || QUANT == 2, "tile cuts: the walk of the quantised 4-ary nodes");
    // Closest-hit rays below a cut (round 5): the entries a ray entered are walked one after the other, the LAST of the list first — the
    // deepest, the small boxes around the tile, where a short ray's nearest hit is — each in the usual near-first order, the closest hit
    // carried from one to the next.  The cut's subtrees partition the tree, so the triangles a ray can reach are those of the walk from the
    // root; the minimum over the hits does not depend on the order they are found in, except where a hit lies within rounding of a box
    // face that a nearer hit has already culled — the oracle walks the product's own cuts in the same order (orc_cut_walk_closest,
    // tests/test_gpu_parity.py::test_closest_hit_below_tile_cuts_is_the_oracles_walk_of_the_same_cuts) and the walk from the root besides.
    // (the table sampler's weight dot(N, d) is made again when the ray is popped — the same expression on the same numbers, for one more
    // 16-byte load of the surfel's record — and the refs take a stack entry's room: launch_stack)
    __shared__ uint32_t miss_count[COUNT32 ? (CUT ? kWave / 2 : kWave) : 1];
    __shared__ uint32_t cut_refs[CUT ? kCutMax : 1];
    __shared__ uint4 queue[kWave * 2];                  // ray queue: {dir.xyz, slot}, {1/dir.xyz, weight — CUT: the ray's entry mask}

    const uint32_t lane = threadIdx.x;
#if AOB_TRACE_STATS
    unsigned long long st_[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
    f3 q_origin = make_f3(0.f, 0.f, 0.f), q_scale = make_f3(0.f, 0.f, 0.f);
    if (QUANT) {
        q_origin = make_f3(p.quant_frame[0], p.quant_frame[1], p.quant_frame[2]);
        q_scale = make_f3(p.quant_frame[3], p.quant_frame[4], p.quant_frame[5]);
    }
    if (QUANT == 2) {
        // the 4-ary nodes hand over a plane q as the float f = 1 + q/32768:  origin + q*scale = (origin - 32768*scale) + f * (32768*scale)
        q_origin = make_f3(fmaf(-32768.0f, q_scale.x, q_origin.x), fmaf(-32768.0f, q_scale.y, q_origin.y), fmaf(-32768.0f, q_scale.z, q_origin.z));
        q_scale = make_f3(q_scale.x * 32768.0f, q_scale.y * 32768.0f, q_scale.z * 32768.0f);
    }
    if (QUANT) {
        // (the same in every lane: in scalar registers they are operands of the pop's arithmetic instead of six vector registers held —
        // and, at the 64-register cap, spilt — across the walk)
        auto uniform = [](float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); };
        q_origin = make_f3(uniform(q_origin.x), uniform(q_origin.y), uniform(q_origin.z));
        q_scale = make_f3(uniform(q_scale.x), uniform(q_scale.y), uniform(q_scale.z));
    }
    // the quantised 4-ary nodes as a raw buffer (base in SGPRs, 4 GiB window: 64 Mi nodes — trace_range checks the count)
    const __amdgpu_buffer_rsrc_t nodes_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.nodes4q, 0, 0xffffffff, 0x00020000);
    // any hit: the rays of a refill start together and visit the top of the tree together, in the refill section
    constexpr bool SCALAR_STEPS = QUANT == 2 && ANYHIT && (!CUT || AOB_CUT_SCALAR);
    // lanes on one node that make a scalar step worth its pass: all but two of a refill's rays (p.scalar_min; measured on BASELINE
    // configs[2] at 42 rays per refill: 16 / 24 / 32 / 40 / 48 lanes -> 73.3 / 71.9 / 71.2 / 70.9 / 72.4 ms; the root alone 71.8, none 73.9)
    const uint32_t kScalarMin = p.scalar_min;
    // overflow stack entry e of this lane lives at gstack[(e * gridDim.x + blockIdx.x) * 64 + lane] (computed on use:
    // the path is cold and the registers are better spent on occupancy)
    // (the base stays in scalar registers — it passes an empty asm — instead of a vector-register copy that the allocator, at the 64-register cap,
    // spilt and re-loaded from scratch INSIDE the main loop: the only scratch accesses there were these, tools/scratch_report.py)
    auto gs_slot = [&](int e) -> uint32_t* { uint32_t* base = p.gstack; asm volatile("" : "+s"(base)); return base + ((uint64_t)e * gridDim.x + blockIdx.x) * kWave + lane; };

    // ---- persistent wavefronts ----
    // The launch is as many wavefronts as the chip holds (32 per CU); each takes work units — one tile (64 surfels) x
    // one part of the samples — off its XCD's queue until all queues are dry.  (One workgroup per unit left a quarter of
    // the wave slots empty with short units, and the launch ended with a tail of idle CUs one wavefront lifetime long.)
    // Work is cut into chunks of consecutive tiles and the chunks are dealt to the 8 XCD queues in turn: while an XCD
    // works through a chunk its private 4 MiB L2 keeps seeing the same part of the atlas, hence the same part of the BVH,
    // and every XCD gets chunks from all over the atlas.  The parts of a tile are consecutive units of one queue: they
    // run at the same time and share their nodes in L2 (four parts: 54 % fewer bytes from HBM than one).
    const uint32_t home = blockIdx.x & 7u;             // workgroups are dealt round-robin over the XCDs: this one's
    uint32_t tried = 0;                                // queues found dry, starting with the home queue
    uint32_t cast_acc = 0;                             // wave-uniform: samples this wavefront generated (horizon tables: fewer than it was given)
  for (;;) {
    uint32_t xcd, rank;
    for (;;) {
        xcd = (home + tried) & 7u;
        uint32_t r = 0;
        if (lane == 0) r = atomicAdd(&p.unit_next[xcd], 1u);
        rank = __builtin_amdgcn_readfirstlane(r);
        if (rank < p.units_per_xcd) break;
        if (++tried == 8u) break;
    }
    if (tried == 8u) break;                            // every queue is dry
    const bool fine = rank >= p.coarse_units;
    const uint32_t nparts = fine ? p.parts_fine : p.parts;
    const uint32_t part = fine ? (rank - p.coarse_units) % p.parts_fine : rank % p.parts;
    const uint32_t kt = fine ? p.coarse_tiles + (rank - p.coarse_units) / p.parts_fine : rank / p.parts;
    uint32_t tile;
    if (kt < p.xcd_chunk_tiles) {
        const uint32_t c = kt / p.chunk, within = kt - c * p.chunk;
        tile = (c * 8u + xcd) * p.chunk + within;
        if (p.band_row) {
            // the chunk is a band of 8 tile rows: walk it in blocks of 8x8 tiles (64x64 texels) instead of row by row
            const uint32_t cb = within >> 6, in_block = within & 63u;
            tile = ((c * 8u + xcd) * 8u + (in_block >> 3)) * p.band_row + cb * 8u + (in_block & 7u);
        }
    } else {
        tile = p.full_tiles + xcd * p.rest_per_xcd + (kt - p.xcd_chunk_tiles);
    }
    if (tile >= p.ntiles) continue;
    const uint64_t wave_first = p.begin + (uint64_t)tile * kWave;            // first surfel of this unit
    const uint32_t nsurf = (uint32_t)min((uint64_t)kWave, p.end - wave_first);

    if (COUNT32) { if (!CUT || lane < kWave / 2) miss_count[lane] = 0u; }
    else { acc[lane][0] = 0; if (!TOTAL_ONLY) acc[lane][1] = 0; }
    lds_stack[lane] = kInvalid;      // entry 0: the INVALID sentinel (intersect_bvh2_lds.cl:103): a pop is one read
    // CUT: this tile's cut — the boxes stay in memory (scalar loads where they are tested), the refs go to LDS
    constexpr bool cut_on = CUT;
    // (one record per sample part where the entries were culled by part, k_tile_cuts)
    const uint32_t* tile_cut = cut_on ? p.tile_cuts + (uint64_t)kCutWords * ((uint64_t)tile * p.cut_records + (p.cut_records > 1u ? (fine ? p.parts + part : part) : 0u)) : nullptr;
    if (cut_on && lane < (uint32_t)kCutMax) cut_refs[lane] = tile_cut[lane];
    // (entries in use: the list ends with empty slots' planes; entry e of n is bit n - 1 - e of a ray's mask)
    uint32_t cut_count = 0;
    if (cut_on) cut_count = (uint32_t)__popcll(__ballot(lane < (uint32_t)kCutMax && tile_cut[kCutMax + 3u * (lane & (kCutMax - 1))] != kEmptyPlanes));
    cut_count = (cut_count + 3u) & ~3u;               // tested four at a time
    const uint32_t cut_shift = 32u - cut_count;       // clz(mask bit of entry e) = 32 - n + e


    // ---- the generator: lane = surfel; sample `gen_next` of every surfel is made in one go ----
    constexpr bool UNITW = SAMPLER == AOB_SAMPLER_COSINE;   // every weight is exactly 1
    constexpr long long kUnit = ACCUM == AOB_ACCUMULATE_QUANTISED ? 255ll : (1ll << 32);   // ftoi(1*255) / f2fix(1)
    bool valid = false;              // this lane's surfel exists and has a tangent frame
    uint32_t gen_key = 0;            // COSINE: hash base of the texel; REFERENCE_COMPAT: first hemisphere-table entry
    if (lane < nsurf) {
        const float4 d = p.frames[4ull * (wave_first + lane) + 3];
        valid = __float_as_uint(d.z) != 0u;
        gen_key = SAMPLER == AOB_SAMPLER_REFERENCE_COMPAT ? __float_as_uint(d.y)        // Hemisphere[DTid.x * SampleCount + i]
                                                           : pcg_hash(__float_as_uint(d.x) + p.seed_hash);
    }
    const uint64_t valid_m = __ballot(valid);
    const uint32_t n_valid = (uint32_t)__popcll(valid_m);
    // this workgroup's part of the samples
    const int32_t gen_begin = (int32_t)((int64_t)p.spp * part / nparts);
    const int32_t gen_end = n_valid ? (int32_t)((int64_t)p.spp * (part + 1u) / nparts) : gen_begin;
    int32_t gen_next = gen_begin;    // wave-uniform: next sample index
    // The ORDER in which a unit's samples are generated is free: a sample's ray depends on its index alone, and the sums do not care.  The COSINE
    // sampler's units go through their samples from the LAST to the first: every row of the stratified grid from its last cell — the most grazing
    // directions, the longest walks — to its first, the steepest and shortest.  The last rays of a unit, which the wavefront drains with most of its
    // lanes idle before it takes the next unit, are then its cheapest.  Inner sections after a unit's last ray was generated: 5.8 % of all at 10.8
    // lanes -> 2.0 % at 5.0; 44.0 -> 43.1 ms on BASELINE configs[2], 3.58 -> 3.42 on configs[1], closest hit (configs[4]'s scene) 177.2 -> 173.1
    // (profiles/r04/cuts/generation_order_ab.jsonl; column-major and alternating orders: no better).
    constexpr bool GEN_DOWN = AOB_GEN_ORDER == 1 && SAMPLER == AOB_SAMPLER_COSINE;
    const int32_t gen_first = GEN_DOWN && gen_end > gen_begin ? gen_end - 1 : gen_begin;
    int32_t gen_cy = gen_first / p.grid.sx, gen_cx = gen_first - gen_cy * p.grid.sx;     // the next sample's cell of the COSINE sampler's grid
    uint32_t q_head = 0, q_count = 0;                   // wave-uniform: the queue is a ring of 64 rays
    // ---- horizon tables (horizon.hip) ----
    // The unit's samples are whole rows cy0 .. cy0 + rows - 1 of the sampler's grid (trace_range makes sure: at most four, inside one word of the
    // table); byte r of a surfel's word is the number of leading cells of row cy0 + r that no triangle of the scene reaches: those samples are
    // unoccluded whatever their jitter and are NOT cast — they go into the sums in the epilogue.  What is left of the rows is a list per surfel,
    // and pass j of the generator makes entry j of every list that has one: row after row, each from its last cell (the most grazing) down to
    // the first blocked one.  The word is fetched again in every pass (L1) rather than held across the walk: this kernel lives at the register cap.
    // (an instantiation of its own — HZMODE — so that the launches without tables compile to the kernel they were before: as a run-time switch the
    // tables' code cost those launches 6 % — 43.8 -> 46.6 ms on BASELINE configs[2] with trace_horizon 1, profiles/r06/horizon/)
    constexpr bool HZ = HZMODE != 0;
    static_assert(!HZ || (SAMPLER == AOB_SAMPLER_COSINE && QUANT == 2), "horizon tables: the COSINE sampler on the quantised 4-ary nodes");
    constexpr bool hz_on = HZ;
    const uint32_t hz_cy0 = (uint32_t)gen_begin / (uint32_t)p.grid.sx;
    const uint32_t hz_rows = (uint32_t)(gen_end - gen_begin) / (uint32_t)p.grid.sx;
    // a surfel's row word -> the rows' remaining cells, cumulated: byte r = cells cast in rows 0 .. r (byte 3: the list's length)
    // (a raw buffer: the table's base in scalar registers, the unit's first word as the scalar offset, the lane's as the vector offset — made where it
    // is used, behind an empty asm, so that no per-lane address is kept, and spilt, across the walk)
    const __amdgpu_buffer_rsrc_t hz_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.horizon, 0, 0xffffffff, 0x00020000);
    const uint32_t hz_unit = (uint32_t)(wave_first * (uint64_t)p.horizon_words + (hz_cy0 >> 2)) * 4u;
    // (up to eight rows: two words of the table.  cum_lo / cum_hi: byte r = cells cast in rows 0 .. r (rows 0 - 3 / 4 - 7); the list's length is the last byte of cum_hi)
    auto hz_cum = [&](uint32_t& skipped, uint32_t& cum_hi) -> uint32_t {
        uint32_t lo = lane;
        asm volatile("" : "+v"(lo));
        const uint32_t sx = (uint32_t)p.grid.sx;
        uint32_t w0 = __builtin_amdgcn_raw_buffer_load_b32(hz_rsrc, lo * (p.horizon_words * 4u), hz_unit, 0) >> ((hz_cy0 & 3u) << 3);
        uint32_t w1 = 0u;
        if (hz_rows > 4u) w1 = __builtin_amdgcn_raw_buffer_load_b32(hz_rsrc, lo * (p.horizon_words * 4u) + 4u, hz_unit, 0);      // (wave-uniform: eight-row units start on a word)
        uint32_t cum = 0u, run = 0u;
        skipped = 0u;
#pragma unroll
        for (uint32_t r = 0; r < 4u; r++) {
            const uint32_t f = r < hz_rows ? min((w0 >> (8u * r)) & 0xffu, sx) : sx;
            if (r < hz_rows) skipped += f;
            run += sx - f;
            cum |= run << (8u * r);
        }
        cum_hi = run * 0x01010101u;
        if (hz_rows > 4u) {
            cum_hi = 0u;
#pragma unroll
            for (uint32_t r = 0; r < 4u; r++) {
                const uint32_t f = r + 4u < hz_rows ? min((w1 >> (8u * r)) & 0xffu, sx) : sx;
                if (r + 4u < hz_rows) skipped += f;
                run += sx - f;
                cum_hi |= run << (8u * r);
            }
        }
        return cum;
    };
    int32_t hz_passes = 0;
    // (a unit of up to four rows — every unit of the default schedules — keeps its list's cumulated counts in ONE register across the walk; the
    // wider units fetch and make them again in every pass.  AOB_HZ_HOLD 0: every unit does — for A/B runs.)
#ifndef AOB_HZ_HOLD
#define AOB_HZ_HOLD 1
#endif
    uint32_t hz_held = 0u;
    if (hz_on) {
        uint32_t sk;
        uint32_t ch = 0u;
        if (valid) hz_held = hz_cum(sk, ch);
        int32_t len = valid ? (int32_t)(ch >> 24) : 0;
        for (int off = 32; off > 0; off >>= 1) len = max(len, __shfl_xor(len, off));
        hz_passes = __builtin_amdgcn_readfirstlane(len);
    }
    // (the generator's pass counter runs over [gen_begin, gen_pass_end): the unit's samples, or the passes its longest list takes)
    const int32_t gen_pass_end = hz_on ? gen_begin + hz_passes : gen_end;

    // ---- the ray a lane is tracing ----
    f3 org, dir, invdir, oxinvdir;
    org = dir = invdir = oxinvdir = make_f3(0.f, 0.f, 0.f);
    float w = 0.0f, closest_t = 0.0f;
    uint3 rot = make_uint3(0u, 0u, 0u);        // QUANT 1: 16 for axes the ray travels down, else 0
    uint3 nsel = make_uint3(0u, 0u, 0u), fsel = make_uint3(0u, 0u, 0u);   // QUANT 2: v_perm_b32 selectors of the near / far plane
    uint32_t slot = 0;                         // the ray's surfel
    uint32_t addr = kIdle;
    uint32_t pending = 0u;                     // CUT: the entries of the tile's cut the ray entered and has not walked yet (bit n - 1 - e: entry e)
    const uint32_t leaf_base = QUANT == 2 ? 0x80000000u : p.leaf_base;      // QUANT 2: a compile-time constant, the votes are sign tests
    // Any hit: the lane's stack top is kept as a BYTE offset into lds_stack (entry e of this lane lives at e * 256 + lane * 4): a
    // push or a pop is the LDS access itself and one add, no address arithmetic.  Closest hit keeps the entry index (its ranked
    // placement of up to three children computes entry numbers anyway; with byte offsets it spilt five more registers and
    // lost 40 %).  Entries from LDS_N up live in the HBM overflow array.
    constexpr bool BYTE_SP = ANYHIT != 0;
    constexpr int kSpUnit = BYTE_SP ? 256 : 1;
    constexpr int kLdsTop = LDS_N * kSpUnit;
    const int sp_first = BYTE_SP ? (int)lane * 4 + 256 : 1;          // entry 1 (entry 0 is the sentinel)
    int spa = sp_first;
    auto lds_entry = [&](int at) -> uint32_t& { return BYTE_SP ? *(uint32_t*)((char*)lds_stack + at) : lds_stack[at * kWave + lane]; };
    auto gs_entry = [&](int at) -> uint32_t* { return gs_slot((BYTE_SP ? at >> 8 : at) - LDS_N); };
#if AOB_EXP_VALU || AOB_EXP_LOAD || AOB_EXP_VALU_SLOW
    uint32_t exp_acc = 0;
#endif

    auto push = [&](uint32_t v) {
        if (spa < kLdsTop) lds_entry(spa) = v;
        else *gs_entry(spa) = v;
        spa += kSpUnit;
    };
    auto pop = [&]() -> uint32_t {
        spa -= kSpUnit;
        uint32_t v;
        if (spa < kLdsTop) { v = lds_entry(spa); asm volatile("" : "+v"(v)); }
        else { v = *gs_entry(spa); asm volatile("" : "+v"(v)); }
        return v;
    };
    // ProcessIntersectionsComputeShader.hlsl:69-90: the ray's score goes to its surfel's sums
    auto finish = [&]() {
        if (ANYHIT) {
            // r is 0 or 1: the sample adds its weight to the weight sum and, if unoccluded, to the total
            // ((1.0f * w) * 255.0f == w * 255.0f exactly).  Unit weights: the weight sum is the sample count (epilogue).
            if (UNITW) {
                if (CUT) { if (addr == kInvalid) atomicAdd(&miss_count[slot >> 1], 1u << ((slot & 1u) << 4)); }
                else if (addr == kInvalid) atomicAdd(&miss_count[slot], 1u);
            } else {
                const long long cW = ACCUM == AOB_ACCUMULATE_QUANTISED ? (long long)ftoi(w * 255.0f) : f2fix(w);
                atomicAdd(&acc[slot][1], (acc_t)cW);
                if (addr == kInvalid) atomicAdd(&acc[slot][0], (acc_t)cW);
            }
        } else {
            // (a hit moved closest_t below tmax)
            const float r = (closest_t < p.tmax && closest_t < p.falloff) ? closest_t / p.falloff : 1.0f;
            const float rw = UNITW ? r * 1.0f : r * w;
            atomicAdd(&acc[slot][0], (acc_t)(ACCUM == AOB_ACCUMULATE_QUANTISED ? (long long)ftoi(rw * 255.0f) : f2fix(rw)));
            if (!UNITW) atomicAdd(&acc[slot][1], (acc_t)(ACCUM == AOB_ACCUMULATE_QUANTISED ? (long long)ftoi(w * 255.0f) : f2fix(w)));
        }
        addr = kIdle;
    };

    // ---- one visit of a quantised 4-ary node: four slab tests, the next node and the deferred children ----
    auto visit4 = [&](const uint4& c0, const uint4& c1, const uint4& c2, const uint4& c3, bool shallow) {
            auto slab = [&](const uint4& c, float& tn) -> bool {
                if (ANYHIT && SAMPLER == AOB_SAMPLER_COSINE) return slab4q_anyhit(c, invdir, oxinvdir, nsel, fsel, tn);
                if (ANYHIT) {
                    // (the table sampler's instantiations hold a weight and two sums more and spill inside the loop: the far selectors are made here, as below)
                    constexpr uint32_t kSwapA = 0x0701000cu ^ 0x0703020cu;
                    return slab4q_anyhit(c, invdir, oxinvdir, nsel, make_uint3(nsel.x ^ kSwapA, nsel.y ^ kSwapA, nsel.z ^ kSwapA), tn);
                }
                // closest hit: intersect_bvh2_lds.cl:54-63 on the plane floats — near clamped at 0, far at the closest hit so far.
                // (Written out here: as a function shared with the any-hit form the register allocator spilt three more values, +40 %.)
                constexpr uint32_t kHalf = 0x3f000000u;
                // (the far plane's selector is the near one's with the two half-words swapped: made here, by one v_xor_b32 each, instead of held in
                // three more registers across the walk — this instantiation is the one that spills inside its loop, tools/scratch_report.py)
                constexpr uint32_t kSwap = 0x0701000cu ^ 0x0703020cu;
                const v2f fx = { __uint_as_float(__builtin_amdgcn_perm(kHalf, c.x, nsel.x)), __uint_as_float(__builtin_amdgcn_perm(kHalf, c.x, nsel.x ^ kSwap)) };
                const v2f fy = { __uint_as_float(__builtin_amdgcn_perm(kHalf, c.y, nsel.y)), __uint_as_float(__builtin_amdgcn_perm(kHalf, c.y, nsel.y ^ kSwap)) };
                const v2f fz = { __uint_as_float(__builtin_amdgcn_perm(kHalf, c.z, nsel.z)), __uint_as_float(__builtin_amdgcn_perm(kHalf, c.z, nsel.z ^ kSwap)) };
                const v2f tx = __builtin_elementwise_fma(fx, (v2f){ invdir.x, invdir.x }, (v2f){ oxinvdir.x, oxinvdir.x });
                const v2f ty = __builtin_elementwise_fma(fy, (v2f){ invdir.y, invdir.y }, (v2f){ oxinvdir.y, oxinvdir.y });
                const v2f tz = __builtin_elementwise_fma(fz, (v2f){ invdir.z, invdir.z }, (v2f){ oxinvdir.z, oxinvdir.z });
                tn = fmaxf(fmaxf(fmaxf(tx.x, ty.x), tz.x), 0.0f);
                const float tf = fminf(fminf(fminf(tx.y, ty.y), tz.y), closest_t);
                return tn <= tf;
            };
            float k0, k1, k2, k3;
            // an empty slot holds an inverted box (min plane above max plane on every axis, bvh_build.hip k_emit_nodes4q): its
            // near distance exceeds its far distance whatever the ray, so it fails the test by itself — no compare of the ref
            const bool h0 = slab(c0, k0), h1 = slab(c1, k1), h2 = slab(c2, k2), h3 = slab(c3, k3);
            if (ANYHIT && AOB_VISIT_BRANCHLESS && shallow) {
                // any-hit: the answer does not depend on the visiting order — follow the first hit child, defer the rest.
                // Every lane of the wavefront has four free entries in LDS (`shallow`): all four refs are written at the stack top, each
                // advancing it only if its box was hit, and the walk goes on with what a pop returns — the first hit child (or, with no
                // hit at all, the entry deferred before).  Same nodes in the same order as the conditional pushes below, no branches.
                lds_entry(spa) = c3.w; spa += h3 ? 256 : 0;
                lds_entry(spa) = c2.w; spa += h2 ? 256 : 0;
                lds_entry(spa) = c1.w; spa += h1 ? 256 : 0;
                lds_entry(spa) = c0.w; spa += h0 ? 256 : 0;
                spa -= 256;
                addr = lds_entry(spa);
            } else if (ANYHIT && !(h0 || h1 || h2 || h3)) {
                addr = pop();
            } else if (ANYHIT) {
                if (h3 && (h0 || h1 || h2)) push(c3.w);
                if (h2 && (h0 || h1)) push(c2.w);
                if (h1 && h0) push(c1.w);
                addr = h0 ? c0.w : (h1 ? c1.w : (h2 ? c2.w : c3.w));
            } else if (!ANYHIT && AOB_VISIT_BRANCHLESS && shallow) {
                // closest hit: nearest child first, the others deferred farthest first, ties by child index (the 4-ary form of
                // intersect_bvh2_lds.cl:123-153; the oracle's traverse4q order) — without a branch.  Children that were not hit count
                // as nearer than any hit (-inf), so the four ranks are a permutation: the child that j - 1 others come before goes to
                // stack entry top + 3 - (j - 1); the hits land on top .. top + nh - 1 with the nearest uppermost, the others above
                // them, where nothing lives (`shallow`: every lane of the wavefront has four free entries in LDS).  A pop then takes
                // the nearest hit — or, with no hit at all, the entry deferred before.  Same order as the ranked placement below.
                const float e0 = h0 ? k0 : -INFINITY, e1 = h1 ? k1 : -INFINITY, e2 = h2 ? k2 : -INFINITY, e3 = h3 ? k3 : -INFINITY;
                const uint32_t c01 = e0 <= e1 ? 256u : 0u, c02 = e0 <= e2 ? 256u : 0u, c03 = e0 <= e3 ? 256u : 0u,
                               c12 = e1 <= e2 ? 256u : 0u, c13 = e1 <= e3 ? 256u : 0u, c23 = e2 <= e3 ? 256u : 0u;
                const uint32_t base = ((uint32_t)spa * kWave + lane) * 4u;
                *(uint32_t*)((char*)lds_stack + (base + c01 + c02 + c03)) = c0.w;
                *(uint32_t*)((char*)lds_stack + (base + 256u - c01 + c12 + c13)) = c1.w;
                *(uint32_t*)((char*)lds_stack + (base + 512u - c02 - c12 + c23)) = c2.w;
                *(uint32_t*)((char*)lds_stack + (base + 768u - c03 - c13 - c23)) = c3.w;
                spa += (int)h0 + (int)h1 + (int)h2 + (int)h3;
                addr = pop();
            } else if (!(h0 || h1 || h2 || h3)) {
                addr = pop();
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
                        const int e = spa + (nh - 1) - r;
                        if (e < kLdsTop) lds_entry(e) = ref; else *gs_entry(e) = ref;
                    }
                };
                place(h0, r0, c0.w); place(h1, r1, c1.w); place(h2, r2, c2.w); place(h3, r3, c3.w);
                spa += nh - 1;
                addr = (h0 && r0 == 0) ? c0.w : ((h1 && r1 == 0) ? c1.w : ((h2 && r2 == 0) ? c2.w : c3.w));
            }
    };

    for (;;) {
        const uint64_t active_m = __ballot(addr != kIdle);
        const bool more = q_count != 0u || gen_next < gen_pass_end;
        if (active_m == 0ull && !more) break;
        AOB_STAT(0, 1);

        // ---- refill section ----
        if (more && (kWave - __popcll(active_m) >= p.refill_idle || active_m == 0ull)) {
            AOB_STAT(5, 1);
            for (int pass = 0; pass < 2; pass++) {
                const uint64_t idle_m = __ballot(addr == kIdle);
                if (q_count != 0u) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m, 0u));
                    if (addr == kIdle && rank < q_count) {
                        const uint32_t pos = (q_head + rank) & (kWave - 1u);
                        const uint4 e0 = queue[pos * 2], e1 = queue[pos * 2 + 1];
                        slot = e0.w;
                        const float4 a = p.frames[4ull * (wave_first + slot)];
                        dir = make_f3(__uint_as_float(e0.x), __uint_as_float(e0.y), __uint_as_float(e0.z));
                        invdir = make_f3(__uint_as_float(e1.x), __uint_as_float(e1.y), __uint_as_float(e1.z));
                        w = __uint_as_float(e1.w);
                        if (CUT && !UNITW) {
                            // the record's last word is the entry mask: the weight again from the surfel's normal, as the generator made it
                            const float4 a1 = p.frames[4ull * (wave_first + slot) + 1];
                            w = dot(make_f3(a.w, a1.x, a1.y), dir);
                        }
                        org = make_f3(fmaf(dir.x, p.bias, a.x), fmaf(dir.y, p.bias, a.y), fmaf(dir.z, p.bias, a.z));   // GenerateRays:47
                        oxinvdir = make_f3(-org.x * invdir.x, -org.y * invdir.y, -org.z * invdir.z);
                        if (QUANT == 2) {
                            // a node word holds (0x8000 | min) | (0x8000 | max) << 16; bytes {0x3f, hi, lo, 0x00} are the float 1 + q/32768.
                            // The near plane is min for rays that travel up the axis, max for rays that travel down it.
                            constexpr uint32_t kLow = 0x0701000cu, kHigh = 0x0703020cu;
                            nsel = make_uint3((e1.x >> 31) ? kHigh : kLow, (e1.y >> 31) ? kHigh : kLow, (e1.z >> 31) ? kHigh : kLow);
                            if (ANYHIT && SAMPLER == AOB_SAMPLER_COSINE) fsel = make_uint3(nsel.x ^ (kLow ^ kHigh), nsel.y ^ (kLow ^ kHigh), nsel.z ^ (kLow ^ kHigh));
                        }
                        if (QUANT) {
                            rot = make_uint3((e1.x >> 31) << 4, (e1.y >> 31) << 4, (e1.z >> 31) << 4);
                            // fold the de-quantisation into the ray: (origin + q*scale - o) * invdir = q*(scale*invdir) + (origin*invdir + oxinvdir)
                            oxinvdir = make_f3(fmaf(q_origin.x, invdir.x, oxinvdir.x), fmaf(q_origin.y, invdir.y, oxinvdir.y), fmaf(q_origin.z, invdir.z, oxinvdir.z));
                            invdir = make_f3(q_scale.x * invdir.x, q_scale.y * invdir.y, q_scale.z * invdir.z);
                        }
                        if (QUANT == 2 && ANYHIT) {
                            // any hit: closest_t stays tmax, so walk the boxes in units of tmax — the clamp of the near
                            // distance to [0, 1] is then an instruction modifier and the far one needs no clamp at all
                            invdir = make_f3(invdir.x * p.inv_tmax, invdir.y * p.inv_tmax, invdir.z * p.inv_tmax);
                            oxinvdir = make_f3(oxinvdir.x * p.inv_tmax, oxinvdir.y * p.inv_tmax, oxinvdir.z * p.inv_tmax);
                        }
                        if (!ANYHIT) closest_t = p.tmax;            // any hit: the bound never moves off tmax, no register for it
                        spa = sp_first;
                        addr = 0;
                        if (cut_on) {
                            // the walk starts below the entries the ray entered, the first of the list first.  Bit 0 of the mask is the
                            // list's last entry.
                            uint32_t m = e1.w;
#if AOB_TRACE_STATS
                            { unsigned long long bits_ = __popc(m); for (int off = 32; off > 0; off >>= 1) bits_ += __shfl_xor(bits_, off); if (lane == (uint32_t)__builtin_ctzll(__ballot(true))) st_[9] += bits_; }
#endif
#if AOB_CUT_LAZY
                            // (the mask stays with the ray; "nothing left on the stack" sends it to the next entry, below the refill section)
                            pending = m;
                            addr = kInvalid;
#else
                            while (m) {
                                const uint32_t lb = (uint32_t)__builtin_ctz(m);
                                push(cut_refs[31u - lb - cut_shift]);
                                m &= m - 1u;
                            }
                            addr = pop();
                            if (addr == kInvalid) finish();          // no box entered: the ray misses (so does every ray of an empty scene)
#endif
                        } else
                        if (p.node_count == 0) { addr = kInvalid; finish(); }      // empty scene: every ray misses
                    }
                    const uint32_t taken = min((uint32_t)__popcll(idle_m), q_count);
                    AOB_STAT(6, taken);
                    q_head = (q_head + taken) & (kWave - 1u);
                    q_count -= taken;
                }
                // rays left in the queue: every idle lane was served
                if (pass == 1 || q_count != 0u || gen_next >= gen_pass_end) break;
                // the queue is empty: sample gen_next of all 64 surfels, one per lane
                int32_t i = GEN_DOWN ? gen_end - 1 - (gen_next - gen_begin) : gen_next;
                int32_t cell_x = gen_cx, cell_y = gen_cy;
                bool make = valid;
                if (hz_on) {
                    // entry j of this surfel's list: the row the cumulated counts put it in, that row's cells from the last one down
                    const uint32_t j = (uint32_t)(gen_next - gen_begin);
                    uint32_t sk;
                    uint32_t ch = 0u;
                    uint32_t cum = hz_held;
                    ch = (hz_held >> 24) * 0x01000000u;
                    if (!AOB_HZ_HOLD || hz_rows > 4u) cum = valid ? hz_cum(sk, ch) : 0u;
                    make = j < (ch >> 24);
                    // the row: how many of the seven cumulated counts the entry's number has reached
                    const uint32_t c0 = cum & 0xffu, c1 = (cum >> 8) & 0xffu, c2 = (cum >> 16) & 0xffu, c3 = cum >> 24;
                    uint32_t r = (j >= c0 ? 1u : 0u) + (j >= c1 ? 1u : 0u) + (j >= c2 ? 1u : 0u);
                    uint32_t before = r == 0u ? 0u : (r == 1u ? c0 : (r == 2u ? c1 : c2));
                    if (hz_rows > 4u && j >= c3) {          // (wave-uniform test first: four-row units never come here)
                        const uint32_t d0 = ch & 0xffu, d1 = (ch >> 8) & 0xffu, d2 = (ch >> 16) & 0xffu;
                        const uint32_t r2 = (j >= d0 ? 1u : 0u) + (j >= d1 ? 1u : 0u) + (j >= d2 ? 1u : 0u);
                        r = 4u + r2;
                        before = r2 == 0u ? c3 : (r2 == 1u ? d0 : (r2 == 2u ? d1 : d2));
                    }
                    cell_y = (int32_t)(hz_cy0 + r);
                    cell_x = p.grid.sx - 1 - (int32_t)(j - before);
                    i = cell_y * p.grid.sx + cell_x;
                }
                gen_next++;
                AOB_STAT(7, 1);
                if (GEN_DOWN) { if (gen_cx-- == 0) { gen_cx = p.grid.sx - 1; gen_cy--; } }
                else if (++gen_cx == p.grid.sx) { gen_cx = 0; gen_cy++; }
                const uint64_t make_m = hz_on ? __ballot(make) : valid_m;
                if (HZ) cast_acc += (uint32_t)__popcll(make_m);
                if (make) {
                    const float4* fr = p.frames + 4ull * (wave_first + lane);
                    const float4 a = fr[0], b = fr[1], c = fr[2];
                    const f3 N = make_f3(a.w, b.x, b.y), X = make_f3(b.z, b.w, c.x), Z = make_f3(c.y, c.z, c.w);
                    f3 h;
                    if (SAMPLER == AOB_SAMPLER_REFERENCE_COMPAT) {
                        const float* t = p.hemisphere + 3ull * gen_key + 3 * i;
                        h = make_f3(t[0], t[1], t[2]);
                    } else {
                        h = cosine_sample_cell(gen_key, i, cell_x, cell_y, p.grid);
                    }
                    const f3 d = make_f3(fmaf(h.z, Z.x, fmaf(h.y, N.x, h.x * X.x)),
                                         fmaf(h.z, Z.y, fmaf(h.y, N.y, h.x * X.y)),
                                         fmaf(h.z, Z.z, fmaf(h.y, N.z, h.x * X.z)));                       // GenerateRays:46
                    const float gw = SAMPLER == AOB_SAMPLER_REFERENCE_COMPAT ? dot(N, d) : 1.0f;
                    const uint32_t pos = (q_head + __builtin_amdgcn_mbcnt_hi((uint32_t)(make_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)make_m, 0u))) & (kWave - 1u);
                    const f3 gi = make_f3(safe_inv1(d.x), safe_inv1(d.y), safe_inv1(d.z));
                    queue[pos * 2] = make_uint4(__float_as_uint(d.x), __float_as_uint(d.y), __float_as_uint(d.z), lane);
                    uint32_t entry_mask = 0u;
                    if (cut_on) {
                        AOB_STAT(8, cut_count);
                        // the new ray against its tile's cut, now, while every lane has a ray: the ray as the pop sets it up (same
                        // expressions, same values), each entry's planes by scalar loads, the walk's own slab test
                        const f3 go = make_f3(fmaf(d.x, p.bias, a.x), fmaf(d.y, p.bias, a.y), fmaf(d.z, p.bias, a.z));
                        f3 ox = make_f3(-go.x * gi.x, -go.y * gi.y, -go.z * gi.z);
                        ox = make_f3(fmaf(q_origin.x, gi.x, ox.x), fmaf(q_origin.y, gi.y, ox.y), fmaf(q_origin.z, gi.z, ox.z));
                        f3 iv = make_f3(q_scale.x * gi.x, q_scale.y * gi.y, q_scale.z * gi.z);
                        if (ANYHIT) {
                            iv = make_f3(iv.x * p.inv_tmax, iv.y * p.inv_tmax, iv.z * p.inv_tmax);
                            ox = make_f3(ox.x * p.inv_tmax, ox.y * p.inv_tmax, ox.z * p.inv_tmax);
                        }
                        constexpr uint32_t kLow = 0x0701000cu, kHigh = 0x0703020cu;
                        const uint3 ns = make_uint3((__float_as_uint(gi.x) >> 31) ? kHigh : kLow, (__float_as_uint(gi.y) >> 31) ? kHigh : kLow, (__float_as_uint(gi.z) >> 31) ? kHigh : kLow);
                        const uint3 fs = make_uint3(ns.x ^ (kLow ^ kHigh), ns.y ^ (kLow ^ kHigh), ns.z ^ (kLow ^ kHigh));
                        // mask = 2 * mask + (the ray enters the box): the compare's carry goes in with one v_addc_co_u32
                        auto shift_in = [&](const uint4& bx) {
                            float tn;
                            const bool h = ANYHIT ? slab4q_anyhit(bx, iv, ox, ns, fs, tn) : slab4q_closest(bx, iv, ox, ns, fs, p.tmax);
                            asm("v_addc_co_u32 %0, vcc, %0, %0, %1" : "+v"(entry_mask) : "s"(__ballot(h)) : "vcc");
                        };
                        {
                            // four entries per pass (the list is padded to a multiple of four with empty slots, which no ray enters), the
                            // next four fetched (scalar loads) while these are tested; the fetch past the list's end reads the next record
                            typedef const __attribute__((address_space(4))) uint32_t* const_words;
                            const uint32_t* eg = tile_cut + kCutMax;
                            asm volatile("" : "+s"(eg));
                            const_words ew = (const_words)eg;
                            uint32_t w0 = ew[0], w1 = ew[1], w2 = ew[2], w3 = ew[3], w4 = ew[4], w5 = ew[5], w6 = ew[6], w7 = ew[7], w8 = ew[8], w9 = ew[9], w10 = ew[10], w11 = ew[11];
                            for (uint32_t e = 0; e < cut_count; e += 4u) {
                                eg += 12;
                                asm volatile("" : "+s"(eg));
                                ew = (const_words)eg;
                                const uint32_t x0 = ew[0], x1 = ew[1], x2 = ew[2], x3 = ew[3], x4 = ew[4], x5 = ew[5], x6 = ew[6], x7 = ew[7], x8 = ew[8], x9 = ew[9], x10 = ew[10], x11 = ew[11];
                                shift_in(make_uint4(w0, w1, w2, 0u));
                                shift_in(make_uint4(w3, w4, w5, 0u));
                                shift_in(make_uint4(w6, w7, w8, 0u));
                                shift_in(make_uint4(w9, w10, w11, 0u));
                                w0 = x0; w1 = x1; w2 = x2; w3 = x3; w4 = x4; w5 = x5; w6 = x6; w7 = x7; w8 = x8; w9 = x9; w10 = x10; w11 = x11;
                            }
                        }
                    }
                    queue[pos * 2 + 1] = make_uint4(__float_as_uint(gi.x), __float_as_uint(gi.y), __float_as_uint(gi.z),
                                                    cut_on ? entry_mask : __float_as_uint(gw));
                }
                q_count = hz_on ? (uint32_t)__popcll(make_m) : n_valid;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the queue writes land before other lanes read them
                __builtin_amdgcn_wave_barrier();
            }
            // ---- scalar steps ----
            // The rays that have just started are at the root (node 0 is nobody's child), all at once, and the rays of one refill
            // — one sample of neighbouring surfels — walk the top of the tree together.  While enough lanes stand on ONE
            // node, that node is fetched by scalar loads and its four boxes are tested from scalar registers, here, outside the
            // walk: those visits never reach the vector memory path, which is what bounds the walk.
            if (SCALAR_STEPS && p.node_count) {
                // (the node is read through the constant address space — nothing writes the nodes while this kernel runs —
                // which is what lets the compiler use scalar loads; its address passes an empty asm so that the loads stay here)
                typedef const __attribute__((address_space(4))) uint32_t* const_words;
                uint32_t a0 = 0u;
                int steps = 24;
                if (CUT) {
                    // (the rays start below their tile's cut: the first step is where the first of them starts, if enough others do)
                    const uint64_t in_m = __ballot(addr < leaf_base);
                    if (in_m == 0ull) steps = 0;
                    else {
                        a0 = (uint32_t)__builtin_amdgcn_readlane((int)addr, (int)__builtin_ctzll(in_m));
                        if ((uint32_t)__popcll(__ballot(addr == a0)) < kScalarMin) steps = 0;
                    }
                }
                for (int step = 0; step < steps; step++) {
                    const uint4* ng = p.nodes4q + 4ull * a0;
                    asm volatile("" : "+s"(ng));
                    const const_words nw = (const_words)ng;
                    const uint4 n0 = make_uint4(nw[0], nw[1], nw[2], nw[3]), n1 = make_uint4(nw[4], nw[5], nw[6], nw[7]),
                                n2 = make_uint4(nw[8], nw[9], nw[10], nw[11]), n3 = make_uint4(nw[12], nw[13], nw[14], nw[15]);
                    const bool here = addr == a0;
                    if (here) {
                        // (the conditional pushes: the branch-free form, with its four LDS writes per lane, is 1.6 % slower here)
                        visit4(n0, n1, n2, n3, false);
                        if (addr == kInvalid) finish();      // nothing left to visit: the ray ends here, before the leaf section sees it
                    }
                    // the next node: where the first lane of the group went, if enough of the others went with it
                    const uint64_t on_m = __ballot(here && addr < leaf_base);
                    if ((uint32_t)__popcll(on_m) < kScalarMin) break;
                    a0 = (uint32_t)__builtin_amdgcn_readlane((int)addr, (int)__builtin_ctzll(on_m));
                    if ((uint32_t)__popcll(__ballot(addr == a0)) < kScalarMin) break;
                }
            }
        }

        // ---- CUT: a ray whose stack has run dry goes on below the next entry of its tile's cut that it entered, shallowest first (the big
        // boxes far from the tile: 51.9 -> 50.3 ms on BASELINE configs[2] against deepest first) ----
        if (CUT && AOB_CUT_LAZY) {
            if (addr == kInvalid && pending != 0u) {
                if (ANYHIT) {
                    const uint32_t hb = (uint32_t)__clz((int)pending);
                    addr = cut_refs[hb - cut_shift];
                    pending &= ~(0x80000000u >> hb);
                } else {
                    // closest hit: the deepest entry first (bit 0 of the mask is the list's last entry)
                    const uint32_t lb = (uint32_t)__builtin_ctz(pending);
                    addr = cut_refs[cut_count - 1u - lb];
                    pending &= pending - 1u;
                }
                spa = sp_first;                     // (the pop that found the sentinel left the stack top below entry 1)
            }
        }

        // ---- leaf section: :162-182 (closest) / :323-339 (any hit); the triangle is inlined in the node ----
        const bool at_leaf = addr >= leaf_base && addr < kIdle;
        const bool inner = addr < leaf_base;
        const uint64_t inner_m = __ballot(inner);
        // (run it, too, when fewer than 16 lanes are left walking: the parked lanes are then worth more than a fuller leaf section —
        // 69.6 -> 68.6 ms on BASELINE configs[2]; thresholds 8 / 12 / 16 / 20 / 24: 69.6 / 69.0 / 68.6 / 68.7 / 69.7)
        // (any hit; closest-hit scoring — configs[4] — loses 1.5 % with it and keeps "nobody left walking")
        if (__popcll(__ballot(at_leaf)) >= p.leaf_wait || (uint32_t)__popcll(inner_m) < (ANYHIT ? 16u : 1u)) {
            { const uint64_t leaf_m_ = __ballot(at_leaf); AOB_STAT(3, 1); AOB_STAT(4, __popcll(leaf_m_)); (void)leaf_m_; }
            if ((AOB_LEAF_HELPERS == 1 || (AOB_LEAF_HELPERS == 2 && ANYHIT)) && QUANT == 2) {
                // ---- helper lanes (round 5) ----
                // A leaf section runs with ~20 of 64 lanes parked, and a parked lane's PAIR costs two Möller–Trumbore passes (with an IEEE division
                // each) that the other 44 lanes sit out.  Here the pair with rank j among the parked pairs hands its second triangle to the lane with
                // rank j among the lanes that are NOT parked at a leaf (walking, idle or ended: their own state stays in its registers): the helper
                // pulls the ray (origin, direction, bound, ref) across with ds_bpermute, both test ONE triangle in the same pass, the parked lane
                // pulls the helper's distance back and keeps the smaller — the same two tests on the same numbers, so the same answer, bit for bit.
                // Who serves whom: every pair pushes its lane number to lane `rank` (ds_permute), every free lane likewise, and each side pulls the
                // other's number from there.  Pairs beyond the free lanes (more than half of the wavefront parked) test their second triangle
                // themselves, afterwards.  Twelve cross-lane moves for one pass of ~70 vector instructions.
                const uint32_t lref = addr - leaf_base;
                const bool pairL = at_leaf && (lref & 0x40000000u) != 0u;
                const uint64_t leaf_m = __ballot(at_leaf), pair_m = __ballot(pairL), free_m = ~leaf_m;
                const uint32_t n_pairs = (uint32_t)__popcll(pair_m), n_free = (uint32_t)__popcll(free_m);
                const uint32_t n_help = min(n_pairs, n_free);
                const uint32_t prank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pair_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pair_m, 0u));
                const uint32_t frank = __builtin_amdgcn_mbcnt_hi((uint32_t)(free_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)free_m, 0u));
                const bool helper = !at_leaf && frank < n_help, served = pairL && prank < n_help;
                uint32_t from = lane, back = lane;
                if (n_help) {           // (wave-uniform)
                    // (lane 63 is nobody's rank unless every lane is a pair / free — and then nobody writes a dummy)
                    const int pair_lane = __builtin_amdgcn_ds_permute((int)((pairL ? prank : 63u) << 2), (int)lane);
                    const int free_lane = __builtin_amdgcn_ds_permute((int)((!at_leaf ? frank : 63u) << 2), (int)lane);
                    const int a = __builtin_amdgcn_ds_bpermute((int)((helper ? frank : lane) << 2), pair_lane);
                    const int h = __builtin_amdgcn_ds_bpermute((int)((served ? prank : lane) << 2), free_lane);
                    if (helper) from = (uint32_t)a;
                    if (served) back = (uint32_t)h;
                }
                auto pull = [&](float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute((int)(from << 2), (int)__float_as_uint(v))); };
                f3 to = org, td = dir;
                uint32_t tref = lref;
                float tbound = ANYHIT ? p.tmax : closest_t;
                if (n_help) {
                    to = make_f3(pull(org.x), pull(org.y), pull(org.z));
                    td = make_f3(pull(dir.x), pull(dir.y), pull(dir.z));
                    tref = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from << 2), (int)lref);
                    if (!ANYHIT) tbound = pull(closest_t);
                }
                float t = tbound;
                auto masked_out = [&](uint32_t shape) -> bool {
                    return shape == p.self_shape ? p.self_hidden != 0u : ((p.hidden_from_others[shape >> 5] >> (shape & 31u)) & 1u) != 0u;
                };
                if (at_leaf || helper) {
                    const float4* np = p.leaf_tris + 3ull * ((tref & 0x3fffffffu) + (helper ? 1u : 0u));
                    const float4 n0 = np[0], n1 = np[1], n2 = np[2];
                    t = intersect_triangle(to, td, make_f3(n0.x, n0.y, n0.z), make_f3(n1.x, n1.y, n1.z), make_f3(n2.x, n2.y, n2.z), tbound);
                    // the ray mask of intersect_bvh2_lds.cl:163-166, by section — the SERVED ray's section: a launch traces one section's surfels,
                    // so self_shape is the same for every ray of the wavefront
                    if (MASKED && masked_out(__float_as_uint(n0.w))) t = tbound;
                }
                if (n_help) {
                    const float t2 = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute((int)(back << 2), (int)__float_as_uint(t)));
                    if (served) t = fminf(t, t2);
                }
                if (n_pairs > n_help) {          // (wave-uniform: more pairs than free lanes)
                    if (pairL && !served) {
                        const float4* np = p.leaf_tris + 3ull * ((lref & 0x3fffffffu) + 1u);
                        const float4 m0 = np[0], m1 = np[1], m2 = np[2];
                        float t2 = intersect_triangle(org, dir, make_f3(m0.x, m0.y, m0.z), make_f3(m1.x, m1.y, m1.z), make_f3(m2.x, m2.y, m2.z), tbound);
                        if (MASKED && masked_out(__float_as_uint(m0.w))) t2 = tbound;
                        t = fminf(t, t2);
                    }
                }
                if (at_leaf) {
                    const bool closer = t < tbound;
                    if (!ANYHIT && closer) closest_t = t;
                    addr = (ANYHIT && closer) ? kEndHit : pop();
                }
            } else
            if (at_leaf) {
                // QUANT 2: the ref may name a PAIR of triangles, p and p + 1 (AOB_PAIR_FLAG): both are fetched in one go
                const float bound = ANYHIT ? p.tmax : closest_t;
                const uint32_t lref = addr - leaf_base;
                const bool pair = QUANT == 2 && (lref & 0x40000000u) != 0u;
                const float4* np = QUANT ? p.leaf_tris + 3ull * (lref & 0x3fffffffu) : p.nodes + 4ull * addr;
                float t;
                // Any hit: the pair's second triangle is fetched WITH the first (round 4: with the quantisation frame in scalar registers
                // the eighteen vertex registers fit; one memory round trip per leaf section instead of two: 46.7 -> 46.1 ms on BASELINE
                // configs[2], the reference's constants 42.5 -> 41.7).  Closest hit — more live state — loses 2.6 % with it and fetches it
                // after the first test, as before.
                constexpr bool BOTH = AOB_LEAF_BOTH && ANYHIT && QUANT == 2;
                float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0, q2 = q0;
                if (BOTH && pair) { q0 = np[3]; q1 = np[4]; q2 = np[5]; }
                {
                    const float4 n0 = np[0], n1 = np[1], n2 = np[2];
                    t = intersect_triangle(org, dir, make_f3(n0.x, n0.y, n0.z), make_f3(n1.x, n1.y, n1.z),
                                           make_f3(n2.x, n2.y, n2.z), bound);
                    if (MASKED) {
                        // the ray mask of intersect_bvh2_lds.cl:163-166, by section: skip the triangles this ray must not see
                        const uint32_t shape = __float_as_uint(QUANT ? n0.w : n1.w);
                        const bool skip = shape == p.self_shape ? p.self_hidden != 0u
                                                                : ((p.hidden_from_others[shape >> 5] >> (shape & 31u)) & 1u) != 0u;
                        if (skip) t = bound;
                    }
                }
                if (QUANT == 2 && pair) {
                    // the second triangle against the same bound: min(t, t2) is what testing them one after the other gives.
                    float4 m0 = q0, m1 = q1, m2 = q2;
                    if (!BOTH) { m0 = np[3]; m1 = np[4]; m2 = np[5]; }
                    float t2 = intersect_triangle(org, dir, make_f3(m0.x, m0.y, m0.z), make_f3(m1.x, m1.y, m1.z),
                                                  make_f3(m2.x, m2.y, m2.z), bound);
                    if (MASKED) {
                        const uint32_t shape = __float_as_uint(m0.w);
                        const bool skip = shape == p.self_shape ? p.self_hidden != 0u
                                                                : ((p.hidden_from_others[shape >> 5] >> (shape & 31u)) & 1u) != 0u;
                        if (skip) t2 = bound;
                    }
                    t = fminf(t, t2);
                }
                const bool closer = t < bound;
                if (!ANYHIT && closer) closest_t = t;
                addr = (ANYHIT && closer) ? kEndHit : pop();
            }
        }

        // ---- inner section: intersect_bvh2_lds.cl:108-160 ----
        if (inner_m) { AOB_STAT(1, 1); AOB_STAT(2, __popcll(inner_m)); if (!more) AOB_STAT(10, (1ull << 40) + (unsigned long long)__popcll(inner_m)); }
        if (inner) {
          if (QUANT == 2) {
            // ---- 4-ary node: four slab tests per visit, children ranked only when the order matters ----
            // buffer loads: a 32-bit byte offset off an SGPR resource (one v_lshlrev) instead of a 64-bit address per lane
            const uint32_t noff = addr << 6;
#ifndef AOB_EXP_AUX
#define AOB_EXP_AUX 0
#endif
            const uint4 c0 = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(nodes_rsrc, noff, 0, AOB_EXP_AUX)),
                        c1 = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(nodes_rsrc, noff + 16u, 0, AOB_EXP_AUX)),
                        c2 = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(nodes_rsrc, noff + 32u, 0, AOB_EXP_AUX)),
                        c3 = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(nodes_rsrc, noff + 48u, 0, AOB_EXP_AUX));
#if AOB_EXP_LOAD
            for (int k = 0; k < AOB_EXP_LOAD; k++) exp_acc ^= __builtin_amdgcn_raw_buffer_load_b32(nodes_rsrc, noff + 12u + 16u * (uint32_t)k, 0, 0);
#endif
#if AOB_EXP_VALU
            for (int k = 0; k < AOB_EXP_VALU; k++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(exp_acc) : "v"(addr));
#endif
#if AOB_EXP_VALU_SLOW
            for (int k = 0; k < AOB_EXP_VALU_SLOW; k++) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(exp_acc) : "v"(addr), "v"(nsel.x));
#endif
            // (wave-uniform: no lane that visits a node now has fewer than four free stack entries in LDS)
            const bool shallow = AOB_VISIT_BRANCHLESS && __ballot(spa >= kLdsTop - 3 * kSpUnit) == 0ull;
#if AOB_TRACE_STATS
            if (!shallow && lane == (uint32_t)__builtin_ctzll(inner_m)) atomicAdd(&g_trace_stats[11], 1ull);
#endif
            visit4(c0, c1, c2, c3, shallow);
          } else {
            const float bound = ANYHIT ? p.tmax : closest_t;
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
                                  fmaf((float)(lz >> 16), invdir.z, oxinvdir.z)), bound);
                s1x = fmaxf(fmaxf(fmaxf(fmaf((float)(rx & 0xffffu), invdir.x, oxinvdir.x), fmaf((float)(ry & 0xffffu), invdir.y, oxinvdir.y)),
                                  fmaf((float)(rz & 0xffffu), invdir.z, oxinvdir.z)), 0.0f);
                s1y = fminf(fminf(fminf(fmaf((float)(rx >> 16), invdir.x, oxinvdir.x), fmaf((float)(ry >> 16), invdir.y, oxinvdir.y)),
                                  fmaf((float)(rz >> 16), invdir.z, oxinvdir.z)), bound);
                al = b.z; ar = b.w;
            } else {
                const float4* np = p.nodes + 4ull * addr;
                const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
                intersect_bbox(make_f3(n0.x, n0.y, n0.z), make_f3(n1.x, n1.y, n1.z), invdir, oxinvdir, bound, s0x, s0y);
                intersect_bbox(make_f3(n2.x, n2.y, n2.z), make_f3(n3.x, n3.y, n3.z), invdir, oxinvdir, bound, s1x, s1y);
                al = __float_as_uint(n0.w); ar = __float_as_uint(n2.w);
            }
            const bool c0 = s0x <= s0y, c1 = s1x <= s1y;
            const bool right_first = (c1 && (s0x > s1x)) || !c0;
            if (c0 && c1) push(right_first ? al : ar);
            addr = right_first ? ar : al;
            if (!(c0 || c1)) {
                addr = pop();
            }
          }
        }
        // kInvalid or kEndHit: the one place a ray ends (CUT: unless it has entries of its tile's cut left to walk)
        if (addr > kIdle && !(CUT && AOB_CUT_LAZY && addr == kInvalid && pending != 0u)) finish();
    }
#if AOB_EXP_VALU || AOB_EXP_LOAD || AOB_EXP_VALU_SLOW
    asm volatile("" :: "v"(exp_acc));
#endif
#if AOB_TRACE_STATS
    if (lane == 0) { for (int i_ = 0; i_ < 12; i_++) { atomicAdd(&g_trace_stats[i_], st_[i_]); st_[i_] = 0; } }
#endif
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's LDS atomics have landed

    // ProcessIntersectionsComputeShader.hlsl:89-90: one surfel per cell per launch.  A workgroup that traced all
    // samples of its tile owns the cells: a plain read-modify-write replaces the 2*spp InterlockedAdds.  Parts of a
    // tile add atomically, exactly in any order: to integer cells, to float cells that hold whole numbers (unit weights, any
    // hit: misses and samples counted), or — fractions — to 64-bit accumulators that are rounded once after the launch.
    if (valid) {
        const float4 d = p.frames[4ull * (wave_first + lane) + 3];
        long long sT = COUNT32 ? (long long)(CUT ? (miss_count[lane >> 1] >> ((lane & 1u) << 4)) & 0xffffu : miss_count[lane]) * kUnit : (long long)acc[lane][0];
        // (the samples the horizon table resolved: unoccluded, weight 1 — what finish() adds for a ray that ends without a hit)
        if (hz_on) { uint32_t sk, ch; (void)hz_cum(sk, ch); sT += (long long)sk * kUnit; }
        const long long sW = UNITW ? (long long)(gen_end - gen_begin) * kUnit : (long long)acc[lane][TOTAL_ONLY ? 0 : 1];
        AobResult* cell = p.results + __float_as_uint(d.x);
        if (p.parts > 1u) {
            if (ACCUM == AOB_ACCUMULATE_QUANTISED) {
                atomicAdd((int32_t*)&cell->total_bits, (int32_t)sT);
                atomicAdd((int32_t*)&cell->weight_bits, (int32_t)sW);
            } else if (COUNT32) {
                // whole numbers (unit weights, samples counted): float adds of them are exact in any order
                unsafeAtomicAdd((float*)&cell->total_bits, (float)sT * 2.3283064365386962890625e-10f);
                unsafeAtomicAdd((float*)&cell->weight_bits, (float)sW * 2.3283064365386962890625e-10f);
            } else {
                // fractions: a float cell is the exact sum rounded ONCE, so the parts add their Q32.32 sums as integers and
                // k_acc_to_cells rounds the total after the launch — the cell a single part would have written
                unsigned long long* a = p.acc_cells + 2ull * __float_as_uint(d.x);
                atomicAdd(a, (unsigned long long)sT);
                atomicAdd(a + 1, (unsigned long long)sW);
            }
        } else {
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
  }   // next unit
    if (HZMODE != 0 && p.cast_total && lane == 0) atomicAdd(p.cast_total, (unsigned long long)cast_acc);
}

// After a launch whose parts met in acc_cells: every surfel of the launch owns one cell (sections in one launch share no texel);
// its exact sums are rounded to float once, added to the cell as a single part would have added them, and cleared for the next launch.
__global__ __launch_bounds__(256)
void k_acc_to_cells(const float4* __restrict__ frames, uint64_t begin, uint64_t end, unsigned long long* __restrict__ acc, AobResult* __restrict__ results)
{
    const uint64_t i = begin + (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= end) return;
    const float4 d = frames[4ull * i + 3];
    if (__float_as_uint(d.z) == 0u) return;              // no tangent frame: no rays, nothing was added
    const uint32_t c = __float_as_uint(d.x);
    const long long sT = (long long)acc[2ull * c], sW = (long long)acc[2ull * c + 1];
    acc[2ull * c] = 0ull; acc[2ull * c + 1] = 0ull;
    AobResult r = results[c];
    r.total_bits = __float_as_uint(__uint_as_float(r.total_bits) + (float)sT * 2.3283064365386962890625e-10f);
    r.weight_bits = __float_as_uint(__uint_as_float(r.weight_bits) + (float)sW * 2.3283064365386962890625e-10f);
    results[c] = r;
}

// Stack entries per lane kept in LDS (deeper ones go to an HBM array): 5 KB per wavefront = 2 KB of queued rays + the
// per-surfel sums + 256 B per entry.  (4-ary walk, 1 M-triangle terrain: 96 % of the pushes land in slots 1..7.)
constexpr int lds_stack_entries(int sampler, int anyhit, int accum)
{
    const int sums = sampler == AOB_SAMPLER_COSINE ? (anyhit ? 1 : (accum == AOB_ACCUMULATE_QUANTISED ? 1 : 2))      // 4-byte counts / totals
                                                   : (accum == AOB_ACCUMULATE_QUANTISED ? 2 : 4);                   // total + weight
    return (3072 - 256 * sums) / 256;       // 11, 10 or 8
}

template <int SAMPLER, int ANYHIT, int ACCUM, int MASKED>
static void launch_stack(int quant, dim3 grid, hipStream_t s, const TraceParams& p)
{
    constexpr int N = lds_stack_entries(SAMPLER, ANYHIT, ACCUM);
    constexpr bool CAN_CUT = AOB_TRACE_CUT != 0;
    // (a cut's 32 refs: with unit weights and any-hit scoring the 16-bit miss counters make their room, otherwise a stack entry does)
    constexpr int NC = (SAMPLER == AOB_SAMPLER_COSINE && ANYHIT) ? N : N - 1;
    // (horizon tables — p.horizon — are the COSINE sampler's, on the quantised 4-ary nodes: instantiations of their own)
    constexpr int HZ = SAMPLER == AOB_SAMPLER_COSINE ? 1 : 0;
    // (28 instead of 32 wavefronts per CU would leave each 5.7 KB of LDS — two more stack entries per lane, AOB_HZ_STACK_MORE 2: measured slower, 33.5 -> 34.2 ms)
#ifndef AOB_HZ_STACK_MORE
#define AOB_HZ_STACK_MORE 0
#endif
    constexpr int NH = HZ ? AOB_HZ_STACK_MORE : 0;
    if (quant == 2 && CAN_CUT && p.tile_cuts && p.horizon) hipLaunchKernelGGL((k_trace<SAMPLER, ANYHIT, ACCUM, NC + NH, 2, MASKED, CAN_CUT ? 1 : 0, HZ>), grid, dim3(kWave), 0, s, p);
    else if (quant == 2 && CAN_CUT && p.tile_cuts) hipLaunchKernelGGL((k_trace<SAMPLER, ANYHIT, ACCUM, NC, 2, MASKED, CAN_CUT ? 1 : 0, 0>), grid, dim3(kWave), 0, s, p);
    else if (quant == 2 && p.horizon) hipLaunchKernelGGL((k_trace<SAMPLER, ANYHIT, ACCUM, N + NH, 2, MASKED, 0, HZ>), grid, dim3(kWave), 0, s, p);
    else if (quant == 2) hipLaunchKernelGGL((k_trace<SAMPLER, ANYHIT, ACCUM, N, 2, MASKED, 0, 0>), grid, dim3(kWave), 0, s, p);
    else if (quant == 1) hipLaunchKernelGGL((k_trace<SAMPLER, ANYHIT, ACCUM, N, 1, MASKED, 0, 0>), grid, dim3(kWave), 0, s, p);
    else hipLaunchKernelGGL((k_trace<SAMPLER, ANYHIT, ACCUM, N, 0, MASKED, 0, 0>), grid, dim3(kWave), 0, s, p);
}
template <int SAMPLER, int ANYHIT, int MASKED>
static void launch_acc(int accum, int quant, dim3 grid, hipStream_t s, const TraceParams& p)
{
    if (accum == AOB_ACCUMULATE_QUANTISED) launch_stack<SAMPLER, ANYHIT, AOB_ACCUMULATE_QUANTISED, MASKED>(quant, grid, s, p);
    else launch_stack<SAMPLER, ANYHIT, AOB_ACCUMULATE_FLOAT, MASKED>(quant, grid, s, p);
}
template <int SAMPLER, int MASKED>
static void launch_hit(int anyhit, int accum, int quant, dim3 grid, hipStream_t s, const TraceParams& p)
{
    if (anyhit) launch_acc<SAMPLER, 1, MASKED>(accum, quant, grid, s, p);
    else launch_acc<SAMPLER, 0, MASKED>(accum, quant, grid, s, p);
}
template <int SAMPLER>
static void launch_masked(int masked, int anyhit, int accum, int quant, dim3 grid, hipStream_t s, const TraceParams& p)
{
    if (masked) launch_hit<SAMPLER, 1>(anyhit, accum, quant, grid, s, p);
    else launch_hit<SAMPLER, 0>(anyhit, accum, quant, grid, s, p);
}

// The azimuth wedges of a launch's sample parts (k_tile_cuts): part q of n covers the samples [spp q / n, spp (q + 1) / n), i.e.
// the rows cy0..cy1 of the COSINE sampler's sx x sy grid, whose azimuth is 2 pi (cy + jitter) / sy.  Coarse parts first, then the
// fine ones (the units of the last tiles of every queue).  A part that reaches the unstratified samples beyond the grid, or
// spans half the circle or more, gets no azimuth bound; more parts or boundaries than the record holds: no culling at all.
static CutWedges cut_wedges(int32_t spp, const CosineGrid& g, uint32_t parts, uint32_t parts_fine)
{
    CutWedges w;
    std::memset(&w, 0, sizeof w);
    if (!AOB_CUT_WEDGES || parts + parts_fine > (uint32_t)kCutWedges || g.sx < 1 || g.sy < 1) return w;
    const double pi = 3.14159265358979323846;
    double line_angle[kCutLines];
    auto line_of = [&](double frac) -> int {           // the line through the origin at azimuth 2 pi frac (mod pi)
        double a = fmod(frac, 0.5);
        if (a > 0.5 - 1e-12) a = 0.0;
        for (int l = 0; l < w.lines; l++) if (fabs(line_angle[l] - a) < 1e-12) return l;
        if (w.lines == kCutLines) return -1;
        line_angle[w.lines] = a;
        w.line_s[w.lines] = (float)sin(2.0 * pi * a); w.line_c[w.lines] = (float)cos(2.0 * pi * a);
        return w.lines++;
    };
    uint32_t index = 0;
    for (int pass = 0; pass < 2; pass++) {
        const uint32_t n = pass ? parts_fine : parts;
        for (uint32_t q = 0; q < n; q++, index++) {
            const int64_t i0 = (int64_t)spp * q / n, i1 = (int64_t)spp * (q + 1) / n;
            if (i1 <= i0) continue;
            const int64_t cy0 = i0 / g.sx, cy1 = (i1 - 1) / g.sx;
            if (cy1 >= g.sy) continue;                                  // unstratified samples: any azimuth
            const double f0 = (double)cy0 / g.sy, f1 = (double)(cy1 + 1) / g.sy;
            if (f1 - f0 > 0.5 - 1e-9) continue;
            // the cone lies on the positive side of the lower boundary's normal (-sin, cos)(f0) and on the negative side of the
            // upper boundary's; a boundary in the second half of the circle is its line's normal negated
            const int l0 = line_of(f0), l1 = line_of(f1);
            if (l0 < 0 || l1 < 0) { std::memset(&w, 0, sizeof w); return w; }
            const bool lo_flip = fmod(f0, 1.0) >= 0.5 - 1e-12, hi_flip = fmod(f1, 1.0) >= 0.5 - 1e-12 && fmod(f1, 1.0) < 1.0 - 1e-12;
            (lo_flip ? w.need_neg[l0] : w.need_pos[l0]) |= 1u << index;
            (hi_flip ? w.need_pos[l1] : w.need_neg[l1]) |= 1u << index;
        }
    }
    w.count = (int32_t)(parts + parts_fine);
    return w;
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

// a schedule field of AobOptions: 0 means the default aob_default_options writes
static int knob(int32_t value, int fallback) { return value > 0 ? value : fallback; }

// The tile cuts of a material's trace launches: 128 words per 64-surfel tile and record (a record per sample part with the COSINE sampler).
// They are an accelerator, not a result: allocated on their own (not in the bake's arena), sized by what the rasterisation covered, when the
// rasterisation ends (rasterize_material -> reserve_tile_cuts) — so a rank that rasterised a strip holds a strip's worth — and again by a
// launch whose schedule wants more.  Without the memory the launch walks from the root: same cells, slower — and NOT silently:
// AobStats::trace_cuts becomes 2 and aob_last_error() holds a warning.
// (tests: aob_debug_fail_tile_cuts(1) makes every allocation of tile cuts fail from then on — the fallback must bake the same cells and say so)
static bool g_fail_tile_cuts = false;
static bool ensure_tile_cuts(AobBake* b, DeviceMaterial& m, uint64_t words)
{
    if (words <= m.tile_cut_capacity && !g_fail_tile_cuts) return true;
    // (a request that failed is not tried again launch after launch — each try is a stream synchronisation and a multi-GB allocation that fails again —
    // until the material is rasterised anew or a launch asks for less)
    if (m.tile_cut_failed && words >= m.tile_cut_failed) return false;
    (void)hipStreamSynchronize(b->stream);
    if (m.tile_cuts) pool_free(m.tile_cuts, sizeof(uint32_t) * m.tile_cut_capacity, b->device);
    m.tile_cuts = nullptr; m.tile_cut_capacity = 0;
    if (!g_fail_tile_cuts && pool_alloc((void**)&m.tile_cuts, sizeof(uint32_t) * words, b->device) == hipSuccess) { m.tile_cut_capacity = words; return true; }
    (void)hipGetLastError();
    m.tile_cuts = nullptr;
    m.tile_cut_failed = words;
    m.cuts_state = 2u;
    set_error("warning: %.1f MiB for the tile cuts of a %dx%d material could not be allocated: its rays walk from the root (same results, slower; AobStats::trace_cuts = 2)",
              (double)(sizeof(uint32_t) * words) / 1048576.0, m.width, m.height);
    return false;
}

static bool cuts_apply(const AobBake* b)
{
    const AobOptions& o = b->opt;
    return AOB_TRACE_CUT && o.bvh_quantised && o.bvh_width == 4 && o.trace_cut != 1;
}

// the sample parts per tile of a launch (and whether the last tiles of a queue are cut finer): trace_range's schedule
static uint32_t schedule_parts(const AobOptions& o, bool tables = true)
{
    uint32_t parts = (uint32_t)max(1, min(o.trace_parts > 0 ? o.trace_parts : 4, o.sample_count / (o.trace_part_min > 0 ? o.trace_part_min : 16)));
    // horizon tables (horizon.hip): a unit's samples are at most four whole rows of the sampler's grid — one word of a surfel's table.  A 32 x 32 grid
    // (1024 spp) in four parts would be eight rows each: eight parts then.
    if (tables && horizon_applies(o)) {
        CosineGrid g;
        cosine_grid(o.sample_count, g);
        const uint32_t most = g.sx <= 16 ? 8u : 4u;          // rows a unit's list may span (trace.hip hz_cum)
        if ((uint32_t)g.sy % parts == 0u && (uint32_t)g.sy / parts > most && g.sy % 4 == 0 && g.sy / 4 <= kCutWedges) parts = (uint32_t)g.sy / 4u;
    }
    return parts;
}

void reserve_tile_cuts(AobBake* b, int32_t material)
{
    if (!cuts_apply(b)) return;
    DeviceMaterial& m = b->materials[material];
    if (m.surfel_count == 0) return;
    // (the records of the schedule with horizon tables and of the one without — a launch may turn the tables down, horizon.hip — whichever needs more)
    uint64_t records = 1;
    for (int with = 0; with < 2; with++) {
        const uint32_t parts = schedule_parts(b->opt, with != 0);
        const bool fine = parts > 1u && (uint32_t)b->opt.sample_count / (2u * parts) >= 16u && 3u * parts <= (uint32_t)kCutWedges;
        records = std::max<uint64_t>(records, b->opt.sampler == AOB_SAMPLER_COSINE ? (fine ? 3ull : 1ull) * parts : 1ull);       // coarse parts + twice as many fine ones
    }
    const uint64_t tiles = m.surfel_count / 64 + m.section_offsets.size() + 1;
    (void)ensure_tile_cuts(b, m, (tiles * records + 1) * (uint64_t)kCutWords);
}

// The overflow stack of the walk (entries beyond those a lane keeps in LDS): sized when the tree is built (build_bvh -> reserve_gstack) for
// the deepest walk any mode of this bake makes, so that no trace launch carries an allocation; a failure is an error, not a fallback.
int reserve_gstack(AobBake* b, uint64_t words)
{
    if (words <= b->gstack_words) return AOB_OK;
    AOB_HIP(hipStreamSynchronize(b->stream));
    if (b->gstack) AOB_HIP(hipFree(b->gstack));
    b->gstack = nullptr; b->gstack_words = 0;
    AOB_HIP(hipMalloc((void**)&b->gstack, sizeof(uint32_t) * words));
    b->gstack_words = words;
    return AOB_OK;
}

static uint32_t stack_overflow_entries(const AobBake* b, bool cut)
{
    const AobOptions& o = b->opt;
    // a lane's stack holds the sentinel plus at most one deferred sibling per tree level
    // (the 4-ary walk defers up to three children per 4-ary level)
    const uint32_t need = ((o.bvh_quantised && o.bvh_width == 4) ? 3u * ((b->bvh_depth + 2) / 2) + 2u : b->bvh_depth + 2) + (cut && !AOB_CUT_LAZY ? (uint32_t)kCutMax : 0u);
    // 8, 10 or 11 stack entries per lane live in LDS (2 KB + 1 KB of sums, 2.5 KB + 512 B of totals, or 2.75 KB + 256 B of counters, + 2 KB of queued
    // rays per wavefront = 32 wavefronts per CU); a launch below tile cuts gives one of them to the cut's refs unless the 16-bit miss counters make
    // their room (COSINE sampler, any hit); deeper entries — rare — go to an HBM array.
    const uint32_t in_lds = (uint32_t)lds_stack_entries(o.sampler, o.use_any_hit, o.accumulate) - (cut && !(o.sampler == AOB_SAMPLER_COSINE && o.use_any_hit) ? 1u : 0u);
    return need > in_lds ? need - in_lds : 0u;
}

int reserve_trace_stack(AobBake* b)
{
    if (b->node_count == 0) return AOB_OK;
    // (with and without cuts: a launch that cannot have its cuts keeps one more entry in LDS, never fewer)
    const uint32_t overflow = max(stack_overflow_entries(b, cuts_apply(b)), stack_overflow_entries(b, false));
    if (!overflow) return AOB_OK;
    return reserve_gstack(b, (uint64_t)(32u * (uint32_t)b->cu_count) * kWave * overflow);
}

int trace_range(AobBake* b, int32_t material, uint64_t begin, uint64_t end)
{
    DeviceMaterial& m = b->materials[material];
    const AobOptions& o = b->opt;
    if (end > m.surfel_count) end = m.surfel_count;
    hipStream_t s = b->stream;
    AOB_HIP(hipEventRecord(b->ev_trace[0], s));
    if (begin < end) {
        const bool cut = cuts_apply(b) && b->node_count > 0;
        const int cut_entries = o.trace_cut >= 4 ? min(o.trace_cut, kCutMax) : kCutMax;
        if (!cut && m.cuts_state != 2u) m.cuts_state = 0u;
        TraceParams p;
        if (!o.bvh_quantised) AOB_TRY(ensure_nodes(b));
        else if (o.bvh_width != 4) AOB_TRY(ensure_nodes2q(b));
        p.nodes = (const float4*)b->nodes;
        p.nodesq = b->nodes2q; p.nodes4q = b->nodes4q; p.leaf_tris = b->leaf_tris; p.quant_frame = b->quant_frame;
        const int quant = o.bvh_quantised ? (o.bvh_width == 4 ? 2 : 1) : 0;
        p.frames = m.frames; p.results = m.results;
        p.hemisphere = b->hemisphere;
        p.node_count = (uint32_t)b->node_count;
        p.leaf_base = b->triangle_count ? (uint32_t)(b->triangle_count - 1) : 0u;
        if (quant == 2) p.leaf_base = 0x80000000u;          // 4-ary refs: leaf = 0x80000000 | triangle, inner = node index
        if (quant == 2 && b->triangle_count >= (1ull << 26)) { set_error("the quantised 4-ary walk addresses its nodes with 32-bit byte offsets: at most 2^26 triangles (use bvh_width 2)"); return AOB_ERR_UNSUPPORTED; }
        p.spp = o.sample_count;
        p.seed_hash = host_pcg(o.seed);
        p.tmax = o.ray_tmax; p.falloff = o.falloff_distance;
        if (o.material_ranges && m.ao_max_range > 0.0f) {          // MaterialObject::AOMaxRange (AobMaterialDesc::ao_max_range)
            if (m.ao_max_range < p.tmax) p.tmax = m.ao_max_range;
            p.falloff = m.ao_max_range;
        }
        if (!o.use_any_hit && o.clamp_tmax_to_falloff && p.falloff < p.tmax) p.tmax = p.falloff;
        p.inv_tmax = 1.0f / p.tmax;
        p.bias = o.ray_bias;
        p.hidden_from_others = b->hidden_from_others;
        // Measured (profiles/r02/trace_knob_sweep.jsonl, shapes.jsonl): with unit weights and any-hit scoring the walk is fastest
        // when 42 lanes wait for a refill (fresh rays of one sample start together and share the top of the tree: configs[2]
        // 76.2 -> 74.1 ms, configs[1] +5 %); closest-hit and weighted scoring keep 24 (configs[4] loses 4.5 % at 42)
        // (round 3, with a material's sections in one launch: closest-hit scoring — no scalar steps, so nothing to gain from rays that
        // start together — is best at 16: 4 / 8 / 12 / 16 / 20 / 24 / 32 / 42 -> 187.6 / 180.7 / 179.4 / 180.0 / 182.2 / 184.3 / 196.5 / 193.0 ms on
        // configs[4]'s scene at 16 spp, 116.9 / 108.8 / 104.8 / 104.1 / 104.8 / 106.1 / 115.4 / 110.2 on the terrain; profiles/r03/roofs/closest_hit_refill.jsonl)
        // (round 4: rays that start below their tile's cut have no top of the tree to share — 16 idle lanes, and 20 parked at a leaf: refill
        // 8 / 16 / 24 / 32 / 48 -> 53.1 / 52.1 / 52.8 / 53.7 / 56.4 ms, leaf wait 16 / 20 / 28 -> 52.6 / 52.1 / 52.5 on BASELINE configs[2];
        // profiles/r04/cuts/)
        // (the table sampler below its tiles' cuts — the reference's own constants on the terrain: refill 8 / 16 / 24 / 32 -> 43.7 / 43.1 / 42.7 / 42.8 ms,
        // leaf wait 16 / 20 / 28 -> 43.5 / 43.1 / 42.6; without cuts 46.6)
        // (round 5, closest-hit rays below their tiles' cuts: refill 8 / 12 / 16 / 24 / 32 -> 68.9 / 69.9 / 70.8 / 73.5 / 78.5 ms on the terrain at 256 spp with falloff scoring,
        // 40.7 / 40.9 / 41.4 / 42.9 / 45.5 on configs[4]'s scene at 2048^2 x 64 spp; profiles/r05/closest_cuts/)
        p.refill_idle = knob(o.trace_refill_idle, o.use_any_hit ? (o.sampler == AOB_SAMPLER_COSINE ? (cut ? 16 : 42) : 24) : (cut ? 8 : 16));
        // (the leaf section: 16 parked lanes, 24 with any-hit scoring — 12 / 16 / 20 / 24 / 28 -> 59.5 / 58.3 / 57.9 / 57.7 / 57.7 ms on
        // BASELINE configs[2]; the scalar steps: since the vector-memory path is what binds the walk (profiles/r03/roofs/) a step pays
        // for 24 lanes on one node — 16 / 24 / 32 / 36 / 40 -> 56.9 / 56.6 / 56.9 / 57.2 / 57.7 ms — where round 2's kernel wanted 40)
        p.leaf_wait = knob(o.trace_leaf_wait, o.use_any_hit ? (cut && o.sampler == AOB_SAMPLER_COSINE ? 20 : 24) : 16);
        // (the table sampler's rays of one refill point everywhere — a sample index means a different direction for every surfel — and
        // part ways a level or two below the root: there a step pays from 8 lanes, 2 / 4 / 6 / 8 / 12 / 24 -> 49.1 / 48.2 / 47.9 / 47.8 /
        // 48.0 / 49.6 ms on the reference's own constants; profiles/r03/roofs/knobs_other_workloads.jsonl)
        p.scalar_min = (uint32_t)min(knob(o.trace_scalar_min, o.sampler == AOB_SAMPLER_COSINE ? 24 : 8), max(1, p.refill_idle));
        cosine_grid(o.sample_count, p.grid);
        // One launch per section where two sections of the material cover a common texel: their contributions are added in
        // section order (BakeEngine.cpp:473,659), and a launch owns the cells it writes.  Where no texel is shared (the rasteriser
        // checks: sections_overlap) and no ray asks which section it came from (no visibility flags), the sections are one
        // list of surfels to the kernel and go in ONE launch — the 101 charts of BASELINE configs[4] are 101 launches of ~1.1
        // units per resident wavefront otherwise, each ending in a tail as long as itself.
        const bool fused = m.section_offsets.size() > 2 && !m.sections_overlap && !b->any_mesh_flags;
        const size_t launches = fused ? 1 : m.section_offsets.size() - 1;
        for (size_t si = 0; si < launches; si++) {
            uint64_t sb = fused ? begin : (m.section_offsets[si] > begin ? m.section_offsets[si] : begin);
            uint64_t se = fused ? end : (m.section_offsets[si + 1] < end ? m.section_offsets[si + 1] : end);
            if (sb >= se) continue;
            p.begin = sb; p.end = se;
            p.ntiles = (uint32_t)((se - sb + kWave - 1) / kWave);
            // Sample parts: a wavefront lives ~11 ms on the headline workload and the launch ends with a tail of idle
            // CUs about that long; four parts per tile shorten both fourfold (a part is at least 16 samples: fewer cost more in
            // prologues and result atomics than they balance).  Only where adding the parts is exact.
            // Adding the parts is exact with integer cells and with float cells that hold whole numbers (unit weights, any hit); float
            // cells that hold fractions go through 64-bit accumulators and are rounded once after the launch (k_acc_to_cells).
            const bool split_direct = o.accumulate == AOB_ACCUMULATE_QUANTISED || (o.sampler == AOB_SAMPLER_COSINE && o.use_any_hit);
            // (horizon tables under trace_horizon 0: a large launch first asks a sample of its tiles whether they pay — before the schedule, which depends on the answer)
            if (horizon_applies(b) && quant == 2 && b->node_count > 0 && o.trace_horizon == 0 && m.horizon_verdict == 0u) { (void)horizon_build(b, m, sb, se, p.tmax, true); m.horizon_declined = false; }
            p.parts = schedule_parts(o, m.horizon_verdict != 2u);          // (a material whose sampled tiles turned the tables down keeps the schedule without them)
            const bool through_acc = !split_direct && p.parts > 1u;
            if (through_acc) {
                const uint64_t cells = (uint64_t)m.width * m.height;
                if (cells > b->acc_cells_count) {               // (aob_bake_create sizes them: only schedules set after creation come here)
                    AOB_HIP(hipStreamSynchronize(s));
                    release_device(b, b->acc_cells);
                    b->acc_cells = nullptr; b->acc_cells_count = 0;
                    AOB_HIP(hipMalloc((void**)&b->acc_cells, 16ull * cells));
                    AOB_HIP(hipMemsetAsync(b->acc_cells, 0, 16ull * cells, s));
                    b->acc_cells_count = cells;
                }
            }
            p.acc_cells = b->acc_cells;
            // chunks: bands of 8 tile rows when every texel of the rasterised rows has exactly one surfel (then the trace
            // order is the plain tile grid), else 2048 tiles
            p.band_row = 0;
            if (m.section_offsets.size() == 2 && m.surfel_count == (uint64_t)m.width * m.window_rows && m.width % 64 == 0 &&
                sb % ((uint64_t)m.width * 8u) == 0 && o.trace_bands != 2)
                p.band_row = (uint32_t)m.width / 8u;
            p.chunk = p.band_row ? p.band_row * 8u : 2048u;
            const uint32_t groups = p.ntiles / (p.chunk * 8u);
            p.xcd_chunk_tiles = groups * p.chunk;
            p.full_tiles = groups * p.chunk * 8u;
            p.rest_per_xcd = (p.ntiles - p.full_tiles + 7u) / 8u;
            // the last eighth of every queue in parts half as long (where a part keeps at least 16 samples): the launch ends
            // when the slowest wavefront finishes its last unit
            const uint32_t queue_tiles = p.xcd_chunk_tiles + p.rest_per_xcd;
            p.parts_fine = p.parts * 2u;
            // (... and where the parts' records still fit the tile cuts' wedge table)
            const uint32_t fine_tiles = (p.parts > 1u && (uint32_t)o.sample_count / p.parts_fine >= 16u && p.parts + p.parts_fine <= (uint32_t)kCutWedges) ? queue_tiles / (uint32_t)knob(o.trace_fine_div, 8) : 0u;
            p.coarse_tiles = queue_tiles - fine_tiles;
            p.coarse_units = p.coarse_tiles * p.parts;
            p.units_per_xcd = p.coarse_units + fine_tiles * p.parts_fine;
            AOB_HIP(hipMemsetAsync(b->trace_unit_next, 0, 8 * sizeof(uint32_t), s));
            p.unit_next = b->trace_unit_next;
            p.tile_cuts = nullptr; p.cut_records = 1u;
            m.tile_cut_tiles = 0; m.tile_cut_records = 0;
            AOB_HIP(hipEventRecord(b->ev_part[0], s));
            // (16-bit miss counters: a part of at most 65 535 samples)
            if (cut && (uint32_t)o.sample_count / p.parts <= 65535u - 1u) {
                // the cuts of this launch's tiles (a wavefront per tile; kCutMax entries of four words each)
                const CutWedges wedges = o.sampler == AOB_SAMPLER_COSINE ? cut_wedges(o.sample_count, p.grid, p.parts, fine_tiles ? p.parts_fine : 0u)
                                                                          : cut_wedges(0, p.grid, 99u, 0u);      // (the table's directions know no parts: one record)
                p.cut_records = wedges.count > 0 ? (uint32_t)wedges.count : 1u;
                const uint64_t words = ((uint64_t)p.ntiles * p.cut_records + 1u) * kCutWords;      // (one record of slack: the entry phase fetches 48 bytes past a list)
                // (reserved when the rasterisation ended: only a schedule set afterwards allocates here; without the memory the launch walks
                // from the root and the bake says so — ensure_tile_cuts)
                if (ensure_tile_cuts(b, m, words)) {
                    // rays of bounded reach (closest-hit scoring with tmax clamped to the falloff distance, a material range): a slot farther from the
                    // tile than any of its rays goes never joins the cut (k_tile_cuts).  One per cent and the bias on top of the span: directions are
                    // unit vectors to within the rounding of the surfel's frame (an fp16 normal: 5e-4).
                    const float reach = (p.tmax + fabsf(p.bias)) * 1.01f;
                    hipLaunchKernelGGL(k_tile_cuts, dim3(p.ntiles), dim3(kWave), 0, s, m.frames, sb, se, b->nodes4q, b->quant_frame, p.node_count,
                                       (float)AOB_CUT_SHRINK_PCT * 0.01f, cut_entries, reach * reach, wedges, m.tile_cuts);
                    p.tile_cuts = m.tile_cuts;
                    m.tile_cut_tiles = p.ntiles; m.tile_cut_records = p.cut_records;
                    if (m.cuts_state != 2u) m.cuts_state = 1u;
                } else p.cut_records = 1u;
            }
            AOB_HIP(hipEventRecord(b->ev_part[1], s));
            // ---- horizon tables (horizon.hip): where the unit's samples are whole rows of the sampler's grid, at most four of them in one word of the table ----
            p.horizon = nullptr; p.horizon_words = 0u; p.cast_total = nullptr;
            if (horizon_applies(b) && quant == 2 && b->node_count > 0 && !(o.trace_horizon == 0 && m.horizon_verdict == 2u)) {
                const uint32_t per = (uint32_t)o.sample_count / p.parts, sx = (uint32_t)p.grid.sx;
                const uint32_t rows = per / sx;
                // (whole rows, 1 / 2 / 4 of them inside one word of the table, or 8 — two words — where a list of 8 rows still fits its byte counts: <= 16 columns)
                const bool whole = per * p.parts == (uint32_t)o.sample_count && rows * sx == per && (rows == 1u || rows == 2u || rows == 4u || (rows == 8u && sx <= 16u)) && !(fine_tiles && rows == 1u);
                if (whole) {
                    if (horizon_build(b, m, sb, se, p.tmax)) {
                        p.horizon = (const uint32_t*)m.horizon; p.horizon_words = horizon_stride(b) / 4u;
                        p.cast_total = b->horizon_counters + 16 + material;
                        m.horizon_asked += (se - sb) * (uint64_t)o.sample_count;
                        if (m.horizon_state != 2u) m.horizon_state = 1u;
                    } else if (m.horizon_declined) {
                        m.horizon_declined = false;               // (a sample of the tiles said the tables would not pay here: no tables, by design)
                    } else {
                        m.horizon_state = 2u;
                        set_error("warning: the horizon tables of a %dx%d material could not be allocated: every sample is cast (same results, slower; AobStats::trace_horizon = 2)", m.width, m.height);
                    }
                }
            }
            AOB_HIP(hipEventRecord(b->ev_part[2], s));
            // as many wavefronts as the chip holds (5 KB of LDS and <= 64 VGPRs each: 32 per CU; under horizon tables 72 registers: 28), fewer for small jobs
            const uint32_t resident = (p.horizon ? 4u * (uint32_t)AOB_TRACE_WAVES_HZ : 32u) * (uint32_t)b->cu_count;
            dim3 grid(min(resident, (p.units_per_xcd * 8u + 7u) / 8u * 8u));
            // the HBM overflow stack (reserved when the tree was built; a launch whose schedule or mode was set later re-sizes it here)
            const uint32_t overflow = stack_overflow_entries(b, p.tile_cuts != nullptr);
            if (overflow) AOB_TRY(reserve_gstack(b, (uint64_t)grid.x * kWave * overflow));
            p.gstack = b->gstack;
            p.self_shape = m.section_mesh[si] + 1u;
            p.self_hidden = (b->meshes[m.section_mesh[si]].flags & AOB_MESH_HIDDEN_FROM_SELF) ? 1u : 0u;
            if (o.sampler == AOB_SAMPLER_REFERENCE_COMPAT) launch_masked<AOB_SAMPLER_REFERENCE_COMPAT>(b->any_mesh_flags, o.use_any_hit, o.accumulate, quant, grid, s, p);
            else launch_masked<AOB_SAMPLER_COSINE>(b->any_mesh_flags, o.use_any_hit, o.accumulate, quant, grid, s, p);
            if (through_acc)
                hipLaunchKernelGGL(k_acc_to_cells, dim3((uint32_t)((se - sb + 255) / 256)), dim3(256), 0, s, m.frames, sb, se, b->acc_cells, m.results);
            AOB_HIP(hipEventRecord(b->ev_part[3], s));
            b->parts_timed = true;
        }
        AOB_HIP(hipGetLastError());
    }
    AOB_HIP(hipEventRecord(b->ev_trace[1], s));
    b->trace_timed = true;
    return AOB_OK;
}

} // namespace aob

extern "C" void aob_debug_fail_tile_cuts(int on) { aob::g_fail_tile_cuts = on != 0; }

#if AOB_TRACE_STATS
// diagnostic builds only: read and clear the loop counters
extern "C" int aob_debug_trace_stats(unsigned long long* out)
{
    unsigned long long zero[16] = { 0 };
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(aob::g_trace_stats), sizeof zero) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(aob::g_trace_stats), zero, sizeof zero) != hipSuccess) return -1;
    return 0;
}
#endif

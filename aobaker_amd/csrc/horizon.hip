// horizon.hip — per-surfel HORIZON TABLES: which strata of the COSINE sampler's grid cannot hold an occluded ray (gfx950).
//
// The COSINE sampler (aob_math.h cosine_sample_cell) is stratified: sample i = (cx, cy) of an sx x sy grid draws u1 in
// [cx/sx, (cx+1)/sx) — u1 = sin^2(angle to the normal) — and u2 in [cy/sy, (cy+1)/sy) — the azimuth in the surfel's tangent frame
// (GenerateRaysComputeShader.hlsl:34-46).  A hit point p of such a ray lies at P + s d with s >= ray_bias, so in the frame's (dual)
// coordinates (alpha, beta, gamma) of p - P the azimuth is atan2(gamma, alpha) = 2 pi u2 and beta / rho = sqrt((1 - u1) / u1): a piece of
// geometry that stays below beta_max and beyond rho_min as seen from P can only be hit by rays with u1 >= rho_min^2 / (rho_min^2 +
// beta_max^2), inside the arc of azimuths it spans.  k_horizon turns that into a table per surfel — for every azimuth row cy the number of
// leading columns cx that NO geometry of the scene reaches — and k_trace (trace.hip) does not generate the samples of those cells: they
// are unoccluded whatever their jitter, and score as the oracle's rays would have.  The table is an accelerator like the tile cuts: it
// moves work and never changes a result cell; every bound below is conservative with margins far above the rounding involved
// (kMarginU, kMarginPhi, kMarginF), and the parity suite bakes with it by default.
//
// One wavefront per tile of 64 surfels (the trace order's 8x8-texel tiles), three phases:
//   A  lane = child box: a walk of the quantised 4-ary tree from the tile (depth first, the four children of up to 16 nodes per round).  A box nearer
//      than r_near — or a leaf that looks big from the tile — is opened down to its triangles (the NEAR list).  Every other box is measured against the
//      tile in the tile's own frame (pole = the surfels' mean normal): its top — the largest sin(elevation) any point of it can have over any origin of
//      the tile, the box grown by the tile's extent — and its arc of azimuths.  H[32] is the tile's horizon: the sin(elevation) below which a direction
//      may be blocked for SOME origin of the tile.  A box wholly under H is dropped (what it could block is blocked already); a leaf in the middle
//      distance is put aside and measured exactly, a wavefront's worth at a time (its vertices against the boxes of 8 groups of 8 origins), and raises H;
//      a small box or a far leaf raises H by its own top; anything else is opened.
//   B  lane = surfel: the near triangles against each surfel's own tangent frame, into its local table (LDS).
//   C  lane = surfel: the boundary of {sin E <= H[az]} as a closed polyline of great-circle arcs (a latitude's chord lies poleward of it:
//      southern latitudes are raised so that the chord stays above), each arc into the local table: from a blocked direction the way to
//      the surfel's pole along its meridian leaves the blocked set through that boundary, inside the same azimuth row.
// The prototypes that chose this shape: tools/measure_horizon_proto2.py (numpy, the oracle's rays as the check), profiles/r06/horizon/.
#include "aob_internal.h"
#include "horizon_kernel.h"

namespace aob {

// ---- host ----

// The tables cost a fixed time per surfel (k_horizon: 4.7 ms for the 65 536 tiles of a 2048^2 atlas over the 1 M-triangle terrain) and save time per SAMPLE: measured on
// MI355X, BASELINE configs[2] (256 spp) traces 12 % faster with them, the terrain with closest-hit scoring likewise, configs[1] (64 spp) a third SLOWER
// (profiles/r06/horizon/).  trace_horizon 0 — the default — therefore means: from 128 samples per surfel on.
bool horizon_applies(const AobOptions& o)
{
    if (o.trace_horizon == 1 || o.sampler != AOB_SAMPLER_COSINE || !o.bvh_quantised || o.bvh_width != 4) return false;
    if (o.trace_horizon == 0 && o.sample_count < 128) return false;
    CosineGrid g;
    cosine_grid(o.sample_count, g);
    return g.sx >= 4 && g.sy >= 4 && g.sx <= 32 && g.sy <= 32 && g.sx * g.sy == o.sample_count;      // (a grid with unstratified samples beyond it keeps them all)
}
bool horizon_applies(const AobBake* b) { return horizon_applies(b->opt); }

uint32_t horizon_stride(const AobBake* b)
{
    CosineGrid g;
    cosine_grid(b->opt.sample_count, g);
    return g.sy <= 16 ? 16u : 32u;
}

// The tables of the surfels [begin, end) of a material (a trace launch's range), into m.horizon.  false: no memory, or — horizon_declined — not worth
// building here: the launch traces every sample.  verdict_only: allocate, and settle whether they are worth building (below), nothing more.
bool horizon_build(AobBake* b, DeviceMaterial& m, uint64_t begin, uint64_t end, float tmax, bool verdict_only)
{
    const AobOptions& o = b->opt;
    const uint32_t stride = horizon_stride(b);
    const uint64_t bytes = (uint64_t)stride * m.surfel_count;
    if (bytes > m.horizon_capacity) {
        (void)hipStreamSynchronize(b->stream);
        pool_free(m.horizon, m.horizon_capacity, b->device);
        m.horizon = nullptr; m.horizon_capacity = 0;
        const uint64_t want = (uint64_t)stride * (m.surfel_capacity > m.surfel_count ? m.surfel_capacity : m.surfel_count);
        if (pool_alloc((void**)&m.horizon, want, b->device) != hipSuccess) { (void)hipGetLastError(); m.horizon = nullptr; return false; }
        m.horizon_capacity = want;
    }
    HorizonParams p;
    std::memset(&p, 0, sizeof p);
    CosineGrid g;
    cosine_grid(o.sample_count, g);
    p.frames = m.frames; p.begin = begin; p.end = end;
    p.nodes4q = b->nodes4q; p.leaf_tris = b->leaf_tris; p.quant_frame = b->quant_frame; p.node_count = (uint32_t)b->node_count;
    p.K = g.sy; p.C = g.sx;
    p.bias = o.ray_bias > 0.0f ? o.ray_bias : 0.0f;
    p.reach = tmax < 1e30f ? (tmax + fabsf(o.ray_bias)) * 1.01f : INFINITY;
    #ifndef AOB_HZ_MID
#define AOB_HZ_NEAR 0.5f
#define AOB_HZ_MID 3.0f
#define AOB_HZ_SMALL 0.14f
#define AOB_HZ_BIG 0.6f
#endif
    p.near_factor = AOB_HZ_NEAR; p.mid_factor = AOB_HZ_MID; p.small2 = AOB_HZ_SMALL * AOB_HZ_SMALL; p.big2 = AOB_HZ_BIG;
    p.out = m.horizon; p.stride = stride;
    p.tile_stride = 1u;
    // (diagnostics only on request — aob_bake_horizon_counters(.., 2): nine contended global atomics per tile cost more than the kernel itself, 5.6 of 7.2 ms)
    p.counters = b->horizon_diag ? b->horizon_counters : nullptr;
    const uint32_t tiles = (uint32_t)((end - begin + kHzWave - 1) / kHzWave);
    auto launch = [&](uint32_t blocks) {
#ifndef AOB_HZ_DYN_LDS
#define AOB_HZ_DYN_LDS 0             // measurement builds: unused dynamic LDS per workgroup, to lower the wavefronts a CU holds
#endif
        if (g.sx <= 16 && g.sy <= 16) hipLaunchKernelGGL((k_horizon<16, true>), dim3(blocks), dim3(kHzWave), AOB_HZ_DYN_LDS, b->stream, p);
        else hipLaunchKernelGGL((k_horizon<32, false>), dim3(blocks), dim3(kHzWave), AOB_HZ_DYN_LDS, b->stream, p);
    };
    // ---- are they worth it HERE?  The tables cost time per surfel and save time per sample they resolve: on BASELINE configs[2] they resolve 58 % of the
    // samples and the trace gains 8 %; on configs[4]'s rocks — every surfel hemmed in by its neighbours within the rays' 1.25 — they resolve 0.4 % and
    // cost 3 %.  With trace_horizon 0 a large launch (4096 tiles and more) measures a sample of its tiles first — every 64th — and goes on only if the
    // sample promises at least kHzWorth samples resolved per surfel; the verdict is kept for the handle (its scene and options do not change).
    // trace_horizon 2 builds them unasked.
    constexpr double kHzWorth = 80.0;
    if (o.trace_horizon == 0 && m.horizon_verdict == 0 && tiles >= 4096u) {
        unsigned long long* cnt = b->horizon_counters + 12;          // [12] free cells, [13] surfels of the sample
        if (hipMemsetAsync(cnt, 0, 2 * sizeof(unsigned long long), b->stream) != hipSuccess) return false;
        p.tile_stride = 64u; p.skipped = cnt;
        launch(tiles / 64u);
        unsigned long long got[2] = { 0, 0 };
        if (hipMemcpyAsync(got, cnt, sizeof got, hipMemcpyDeviceToHost, b->stream) != hipSuccess || hipStreamSynchronize(b->stream) != hipSuccess) return false;
        p.tile_stride = 1u; p.skipped = nullptr;
        const double per_surfel = got[1] ? (double)got[0] / (double)got[1] : 0.0;      // samples resolved per surfel
        m.horizon_verdict = per_surfel >= kHzWorth ? 1u : 2u;
        m.horizon_sampled = (float)per_surfel;
    }
    if (o.trace_horizon == 0 && m.horizon_verdict == 2u) { m.horizon_declined = true; return false; }
    if (verdict_only) return true;          // (trace_range asks before it makes its schedule; the tables themselves when it launches)
    launch(tiles);
    return hipGetLastError() == hipSuccess;
}

} // namespace aob

// raster.hip — UV-space surfel rasterisation + texel extraction on the GPU (gfx950).
//
// Replaces, per material and per section (reference BakeEngine.cpp:509-588):
//   * RasterizeVertexShader.hlsl:26-35 / RasterizePixelShader.hlsl:17-23 drawn
//     once per stencil layer (LayerDSS, BakeEngine.cpp:130-140,539-545),
//   * three full-surface read-backs per layer (:548-564), and
//   * the single-threaded CPU extraction loop (:566-588).
// The layer peel keeps, for every covered texel, exactly the LAST fragment
// drawn, so one pass suffices here: k_cover takes the
// atomicMax of (triangle index + 1) per texel, an exclusive scan compacts the
// covered texels in the reference's row-major order, and k_emit_surfels
// interpolates the winner's attributes straight into the 40-byte
// TexelRasterization record (BakeEngine.cpp:98-116).
//
// Coverage is exact integer arithmetic on 24.8 fixed-point coordinates with
// the D3D11 top-left rule, so GPU and oracle agree texel for texel.
#include "aob_internal.h"
#include "aob_math.h"

namespace aob {

static constexpr int kBlock = 256;
static constexpr int kScanItems = 8;                       // texels per thread in the scan kernels
static constexpr int kScanTile = kBlock * kScanItems;      // 2048 texels per block
static constexpr float kCoordLimit = 67108864.0f;          // 2^26 sub-pixels

struct MeshXform { float m[16]; };

__device__ __forceinline__ int64_t edge_fn(int64_t ax, int64_t ay, int64_t bx, int64_t by, int64_t px, int64_t py)
{
    return (bx - ax) * (py - ay) - (by - ay) * (px - ax);
}
__device__ __forceinline__ bool edge_top_left(int64_t ax, int64_t ay, int64_t bx, int64_t by)
{
    int64_t dx = bx - ax, dy = by - ay;
    return dy < 0 || (dy == 0 && dx > 0);
}

// RasterizeVertexShader.hlsl:29 + viewport transform: pixel = (u*W, v*H); 8 sub-pixel bits.
__device__ __forceinline__ bool snap_uv(const float* uv, int32_t W, int32_t H, int64_t& X, int64_t& Y)
{
    float sx = uv[0] * (float)W * 256.0f, sy = uv[1] * (float)H * 256.0f;
    if (!(fabsf(sx) < kCoordLimit) || !(fabsf(sy) < kCoordLimit)) return false;
    X = (int64_t)rintf(sx); Y = (int64_t)rintf(sy);
    return true;
}

// One lane per triangle.  A triangle whose bounding box holds at most kSmallBox texels — nearly all of them once a mesh
// has about as many triangles as the atlas has texels (the 1 M-triangle terrain on 2048^2: 4 texels each) — is rasterised
// by its own lane; the others are handed round the wavefront one at a time, all 64 lanes striding over the box.
static constexpr int kSmallBox = 16;

struct CoverTri { int64_t X[3], Y[3]; int64_t px0, py0; int64_t b0, b1, b2; uint64_t bw, npix; };

// AOB_RASTER_CONSERVATIVE: a fragment whose triangle covers the texel centre carries this bit in its cover key, so it
// beats (atomicMax) every fragment that merely overlaps the texel's square
static constexpr uint32_t kCentreBit = 0x80000000u;

template <bool CONSERVATIVE>
__device__ __forceinline__ bool cover_setup(const float* __restrict__ uvs, const uint32_t* __restrict__ indices, uint32_t t,
                                            int32_t W, int32_t H, int32_t row_begin, int32_t row_end, CoverTri& c)
{
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; k++) ok = snap_uv(uvs + 2ull * indices[3ull * t + k], W, H, c.X[k], c.Y[k]) && ok;
    if (!ok) return false;
    int64_t area = edge_fn(c.X[0], c.Y[0], c.X[1], c.Y[1], c.X[2], c.Y[2]);
    if (area == 0) return false;
    if (area < 0) { int64_t tx = c.X[1], ty = c.Y[1]; c.X[1] = c.X[2]; c.Y[1] = c.Y[2]; c.X[2] = tx; c.Y[2] = ty; }
    int64_t minx = min(min(c.X[0], c.X[1]), c.X[2]), maxx = max(max(c.X[0], c.X[1]), c.X[2]);
    int64_t miny = min(min(c.Y[0], c.Y[1]), c.Y[2]), maxy = max(max(c.Y[0], c.Y[1]), c.Y[2]);
    // texels whose centre lies in the bounding box — or, CONSERVATIVE, whose open square overlaps it
    int64_t px0 = max(CONSERVATIVE ? minx >> 8 : (minx - 128 + 255) >> 8, (int64_t)0);
    int64_t px1 = min(CONSERVATIVE ? (maxx - 1) >> 8 : (maxx - 128) >> 8, (int64_t)W - 1);
    // only the texel rows [row_begin, row_end) of the atlas are wanted (a rank's strip; the whole atlas by default)
    int64_t py0 = max(CONSERVATIVE ? miny >> 8 : (miny - 128 + 255) >> 8, (int64_t)row_begin);
    int64_t py1 = min(CONSERVATIVE ? (maxy - 1) >> 8 : (maxy - 128) >> 8, (int64_t)row_end - 1);
    if (px1 < px0 || py1 < py0) return false;
    c.b0 = edge_top_left(c.X[0], c.Y[0], c.X[1], c.Y[1]) ? 0 : -1;
    c.b1 = edge_top_left(c.X[1], c.Y[1], c.X[2], c.Y[2]) ? 0 : -1;
    c.b2 = edge_top_left(c.X[2], c.Y[2], c.X[0], c.Y[0]) ? 0 : -1;
    c.px0 = px0; c.py0 = py0;
    c.bw = (uint64_t)(px1 - px0 + 1);
    c.npix = c.bw * (uint64_t)(py1 - py0 + 1);
    return true;
}

// the largest value of the edge function a -> b over the texel square around its centre value e: e + 128 (|dx| + |dy|)
__device__ __forceinline__ int64_t edge_reach(int64_t ax, int64_t ay, int64_t bx, int64_t by)
{
    const int64_t dx = bx - ax, dy = by - ay;
    return 128 * ((dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy));
}

// AOB_LAYERS_ALL (PEEL): one pass per layer of the reference's stencil peel.  The pass keeps, per texel, the FIRST triangle
// (in index order) after the one the previous pass kept: cover holds 0xffffffff - t, so atomicMax finds the smallest t
// and 0 still means "no fragment"; prev holds the previous layer's t + 1 (0: none yet).
template <bool CONSERVATIVE, bool PEEL>
__device__ __forceinline__ void cover_texel(const CoverTri& c, uint64_t p, uint32_t t, int32_t W, uint32_t* __restrict__ cover,
                                            const uint32_t* __restrict__ prev)
{
    const int64_t py = c.py0 + (int64_t)(p / c.bw), px = c.px0 + (int64_t)(p % c.bw);
    const int64_t cx = px * 256 + 128, cy = py * 256 + 128;
    const int64_t e0 = edge_fn(c.X[0], c.Y[0], c.X[1], c.Y[1], cx, cy), e1 = edge_fn(c.X[1], c.Y[1], c.X[2], c.Y[2], cx, cy),
                  e2 = edge_fn(c.X[2], c.Y[2], c.X[0], c.Y[0], cx, cy);
    const bool centre = e0 + c.b0 >= 0 && e1 + c.b1 >= 0 && e2 + c.b2 >= 0;
    if (PEEL) {
        if (centre && t + 1 > prev[(uint64_t)py * W + px]) atomicMax(&cover[(uint64_t)py * W + px], 0xffffffffu - t);
        return;
    }
    if (!CONSERVATIVE) {
        if (centre) atomicMax(&cover[(uint64_t)py * W + px], t + 1);       // last fragment drawn wins
        return;
    }
    // the square and the triangle share area iff no edge (nor, by the loop bounds, a side of the square) separates them
    const bool touch = e0 + edge_reach(c.X[0], c.Y[0], c.X[1], c.Y[1]) > 0 && e1 + edge_reach(c.X[1], c.Y[1], c.X[2], c.Y[2]) > 0 &&
                       e2 + edge_reach(c.X[2], c.Y[2], c.X[0], c.Y[0]) > 0;
    if (centre) atomicMax(&cover[(uint64_t)py * W + px], kCentreBit | (t + 1));
    else if (touch) atomicMax(&cover[(uint64_t)py * W + px], t + 1);
}

__device__ __forceinline__ int64_t bcast64(int64_t v, int src)
{
    const uint32_t lo = __shfl((uint32_t)(uint64_t)v, src), hi = __shfl((uint32_t)((uint64_t)v >> 32), src);
    return (int64_t)((uint64_t)hi << 32 | lo);
}

template <bool CONSERVATIVE, bool PEEL>
__global__ __launch_bounds__(kBlock)
void k_cover(const float* __restrict__ uvs, const uint32_t* __restrict__ indices, uint32_t ntri,
             int32_t W, int32_t H, int32_t row_begin, int32_t row_end, uint32_t* __restrict__ cover, const uint32_t* __restrict__ prev)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    CoverTri c;
    const bool live = t < ntri && cover_setup<CONSERVATIVE>(uvs, indices, t, W, H, row_begin, row_end, c);
    if (live && c.npix <= (uint64_t)kSmallBox)
        for (uint64_t p = 0; p < c.npix; p++) cover_texel<CONSERVATIVE, PEEL>(c, p, t, W, cover, prev);
    // the big ones, one after the other, by the whole wavefront
    uint64_t big = __ballot(live && c.npix > (uint64_t)kSmallBox);
    while (big) {
        const int src = __ffsll((unsigned long long)big) - 1;
        big &= big - 1;
        CoverTri g;
#pragma unroll
        for (int k = 0; k < 3; k++) { g.X[k] = bcast64(c.X[k], src); g.Y[k] = bcast64(c.Y[k], src); }
        g.px0 = bcast64(c.px0, src); g.py0 = bcast64(c.py0, src);
        g.b0 = bcast64(c.b0, src); g.b1 = bcast64(c.b1, src); g.b2 = bcast64(c.b2, src);
        g.bw = (uint64_t)bcast64((int64_t)c.bw, src); g.npix = (uint64_t)bcast64((int64_t)c.npix, src);
        const uint32_t gt = __shfl(t, src);
        for (uint64_t p = lane; p < g.npix; p += 64) cover_texel<CONSERVATIVE, PEEL>(g, p, gt, W, cover, prev);
    }
}

// ---- row-major exclusive scan of (cover != 0) -----------------------------------

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total_out)
{
    __shared__ uint32_t wave_sums[kBlock / 64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t y = __shfl_up(inc, off);
        if (lane >= (uint32_t)off) inc += y;
    }
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) { if ((uint32_t)w < wave) base += wave_sums[w]; total += wave_sums[w]; }
    __syncthreads();
    *total_out = total;
    return base + inc - v;
}

__global__ __launch_bounds__(kBlock)
void k_scan_count(const uint32_t* __restrict__ cover, uint64_t cells, uint32_t* __restrict__ block_sums,
                  uint8_t* __restrict__ seen, uint32_t* __restrict__ overlap)
{
    uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
    uint32_t c = 0;
    bool again = false;
#pragma unroll
    for (int k = 0; k < kScanItems; k++)
        if (base + k < cells && cover[base + k] != 0u) {
            c++;
            // a texel an earlier section (or layer) of this rasterisation covered too: the sections overlap (trace.hip
            // keeps their launches apart then)
            if (seen[base + k]) again = true; else seen[base + k] = 1;
        }
    if (again) *overlap = 1u;
    uint32_t total;
    block_exclusive_scan(c, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// single block: exclusive scan of the per-tile counts, total into block_sums[nblocks]
__global__ __launch_bounds__(kBlock)
void k_scan_partials(uint32_t* block_sums, uint32_t nblocks)
{
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += kBlock) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < nblocks ? block_sums[i] : 0u;
        uint32_t total;
        uint32_t ex = block_exclusive_scan(v, &total);
        uint32_t carry = carry_s;
        if (i < nblocks) block_sums[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[nblocks] = carry_s;
}

// RasterizeVertexShader.hlsl:30-32 (row vector times model matrix; the normal uses the model
// matrix itself and is normalised per vertex, not per fragment)
__device__ __forceinline__ f3 xform(const MeshXform& x, const float* p, bool point)
{
    f3 r; float t;
    t = p[0] * x.m[0]; t = fmaf(p[1], x.m[4], t); t = fmaf(p[2], x.m[8], t);  r.x = point ? t + x.m[12] : t;
    t = p[0] * x.m[1]; t = fmaf(p[1], x.m[5], t); t = fmaf(p[2], x.m[9], t);  r.y = point ? t + x.m[13] : t;
    t = p[0] * x.m[2]; t = fmaf(p[1], x.m[6], t); t = fmaf(p[2], x.m[10], t); r.z = point ? t + x.m[14] : t;
    return r;
}

// AOB_RASTER_CONSERVATIVE: barycentric coordinates of the point of the triangle nearest to (cx, cy), for a texel whose
// centre the triangle does not cover: the point lies on one of the edges the centre is on the wrong side of (the
// first of them on ties).  Integer dot products; the edge parameter in float, the squared distances in double.
__device__ __noinline__ void nearest_point(const int64_t* X, const int64_t* Y, int64_t cx, int64_t cy, float& l0, float& l1, float& l2)
{
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
        if (d2 < best) { best = d2; l[a] = 1.0f - t; l[bb] = t; l[cc] = 0.0f; }      // l[k] weighs vertex k
    }
    l0 = l[0]; l1 = l[1]; l2 = l[2];
}

struct RasterVertex { f3 wpos, wnrm; float u, v; };

template <bool CONSERVATIVE, bool PEEL>
__global__ __launch_bounds__(kBlock)
void k_emit_surfels(uint32_t* __restrict__ cover, uint64_t cells, const uint32_t* __restrict__ block_offsets,
                    const float* __restrict__ positions, const float* __restrict__ normals,
                    const float* __restrict__ uvs, const uint32_t* __restrict__ indices,
                    MeshXform xf, int32_t W, int32_t H, int32_t row0, int fp16_mode,
                    AobSurfel* __restrict__ surfels, uint32_t* __restrict__ section_index, uint32_t* __restrict__ prev)
{
    // the block's 2048 texels in eight stripes of 256: consecutive lanes take consecutive texels, so the 40-byte records of a
    // wavefront land next to each other (one texel run per lane: 3x the time, the stores 320 B apart)
    uint32_t offset = block_offsets[blockIdx.x];
#pragma unroll 1
    for (int k = 0; k < kScanItems; k++) {
        const uint64_t cell = (uint64_t)blockIdx.x * kScanTile + (uint64_t)k * kBlock + threadIdx.x;
        const uint32_t cv = cell < cells ? cover[cell] : 0u;
        uint32_t total;
        const uint32_t local = offset + block_exclusive_scan(cv != 0u ? 1u : 0u, &total);
        offset += total;
        if (!cv) continue;
        const uint32_t t = PEEL ? 0xffffffffu - cv : (cv & ~kCentreBit) - 1;
        const int32_t py = row0 + (int32_t)(cell / (uint64_t)W), px = (int32_t)(cell % (uint64_t)W);    // `cover` starts at texel row row0
        RasterVertex rv[3];
        int64_t X[3], Y[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            uint32_t vi = indices[3ull * t + j];
            const float* uv = uvs + 2ull * vi;
            snap_uv(uv, W, H, X[j], Y[j]);
            rv[j].wpos = xform(xf, positions + 3ull * vi, true);
            rv[j].wnrm = normalize(xform(xf, normals + 3ull * vi, false));
            rv[j].u = uv[0]; rv[j].v = uv[1];
        }
        int64_t area = edge_fn(X[0], Y[0], X[1], Y[1], X[2], Y[2]);
        if (area < 0) {
            int64_t tx = X[1], ty = Y[1]; X[1] = X[2]; Y[1] = Y[2]; X[2] = tx; Y[2] = ty;
            RasterVertex tv = rv[1]; rv[1] = rv[2]; rv[2] = tv;
            area = -area;
        }
        const int64_t cx = (int64_t)px * 256 + 128, cy = (int64_t)py * 256 + 128;
        const float fa = (float)area;
        float l0 = (float)edge_fn(X[1], Y[1], X[2], Y[2], cx, cy) / fa;
        float l1 = (float)edge_fn(X[2], Y[2], X[0], Y[0], cx, cy) / fa;
        float l2 = (float)edge_fn(X[0], Y[0], X[1], Y[1], cx, cy) / fa;
        if (CONSERVATIVE && !(cv & kCentreBit)) nearest_point(X, Y, cx, cy, l0, l1, l2);
#define AOB_LERP(f) fmaf(l2, rv[2].f, fmaf(l1, rv[1].f, l0 * rv[0].f))
        AobSurfel s;
        s.texel_x = px; s.texel_y = py;
        s.position[0] = AOB_LERP(wpos.x); s.position[1] = AOB_LERP(wpos.y); s.position[2] = AOB_LERP(wpos.z);
        // RasterizePixelShader.hlsl:20-21: U rides with the fp32 position; the normal and V pass
        // through the RGBA16F target (BakeEngine.cpp:446,584-585)
        s.texcoord[0] = AOB_LERP(u);
        s.texcoord[1] = quantise_fp16(AOB_LERP(v), fp16_mode);
        s.normal[0] = quantise_fp16(AOB_LERP(wnrm.x), fp16_mode);
        s.normal[1] = quantise_fp16(AOB_LERP(wnrm.y), fp16_mode);
        s.normal[2] = quantise_fp16(AOB_LERP(wnrm.z), fp16_mode);
#undef AOB_LERP
        surfels[local] = s;
        section_index[local] = local;
        cover[cell] = local + 1;            // from here on: row-major rank + 1 (read by the tile-order pass)
        if (PEEL) prev[cell] = t + 1;       // the next layer starts after this triangle
    }
}

// ---- trace order: 8x8 texel tiles -------------------------------------------------------------
// The extraction order above is the reference's (row-major; its chunk-local sample indexing
// depends on it).  The trace kernel instead wants the 64 surfels of a wavefront to be spatial
// neighbours — an 8x8 texel tile rather than a 64x1 strip (+11 % rays/s, better L1 reuse) — so a
// second scan enumerates the atlas tile by tile (tiles row-major, texels row-major inside a tile)
// and emits perm[j] = row-major rank of the j-th covered texel in that order.
__device__ __forceinline__ bool tile_cell(uint64_t e, int32_t W, int32_t H, uint32_t tiles_x, uint64_t& cell)
{
    const uint64_t tile = e >> 6; const uint32_t in = (uint32_t)e & 63u;
    const uint32_t ty = (uint32_t)(tile / tiles_x), tx = (uint32_t)(tile - (uint64_t)ty * tiles_x);
    const uint32_t y = ty * 8u + (in >> 3), x = tx * 8u + (in & 7u);
    cell = (uint64_t)y * (uint32_t)W + x;
    return x < (uint32_t)W && y < (uint32_t)H;
}

__global__ __launch_bounds__(kBlock)
void k_tile_count(const uint32_t* __restrict__ rank, int32_t W, int32_t H, uint32_t tiles_x, uint64_t total_e,
                  uint32_t* __restrict__ block_sums)
{
    uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        uint64_t cell;
        if (base + k < total_e && tile_cell(base + k, W, H, tiles_x, cell)) c += rank[cell] != 0u;
    }
    uint32_t total;
    block_exclusive_scan(c, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(kBlock)
void k_tile_emit(const uint32_t* __restrict__ rank, int32_t W, int32_t H, uint32_t tiles_x, uint64_t total_e,
                 const uint32_t* __restrict__ block_offsets, uint32_t* __restrict__ perm)
{
    uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanItems;
    uint32_t rv[kScanItems];
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        uint64_t cell;
        rv[k] = (base + k < total_e && tile_cell(base + k, W, H, tiles_x, cell)) ? rank[cell] : 0u;
        c += rv[k] != 0u;
    }
    uint32_t total;
    uint32_t j = block_exclusive_scan(c, &total) + block_offsets[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kScanItems; k++) if (rv[k]) perm[j++] = rv[k] - 1u;
}

// number of covered texels before each tile row (for tile-row aligned multi-GPU shards)
__global__ __launch_bounds__(kBlock)
void k_tile_row_prefix(const uint32_t* __restrict__ rank, int32_t W, int32_t H, uint32_t tiles_x, uint32_t tiles_y,
                       const uint32_t* __restrict__ block_offsets, uint32_t total, uint32_t* __restrict__ row_prefix)
{
    uint32_t ty = blockIdx.x * kBlock + threadIdx.x;
    if (ty > tiles_y) return;
    if (ty == tiles_y) { row_prefix[ty] = total; return; }
    const uint64_t e0 = (uint64_t)ty * tiles_x * 64u;
    const uint64_t blk = e0 / kScanTile;
    uint32_t c = block_offsets[blk];
    for (uint64_t e = blk * kScanTile; e < e0; e++) {
        uint64_t cell;
        if (tile_cell(e, W, H, tiles_x, cell)) c += rank[cell] != 0u;
    }
    row_prefix[ty] = c;
}

// Per-surfel trace record: everything the trace kernel needs when a lane picks up a block of
// samples of an arbitrary surfel of its wave — position, normal, the tangent frame of
// GenerateRaysComputeShader.hlsl:34-41, the result cell and the base of the surfel's run in the
// compat hemisphere table (Hemisphere[DTid.x * SampleCount + i] with DTid.x chunk-local, :45).
// A surfel with a non-finite frame or an out-of-range texel is flagged invalid: all its weights
// would be NaN, and (int)NaN == 0 in HLSL, so it contributes nothing.
__global__ __launch_bounds__(kBlock)
void k_surfel_frames(const AobSurfel* __restrict__ surfels, const uint32_t* __restrict__ section_index,
                     const uint32_t* __restrict__ perm, uint64_t count,
                     int32_t W, int32_t H, uint32_t chunk_max, uint32_t spp, float4* __restrict__ frames)
{
    uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;      // position in trace (tile) order
    if (j >= count) return;
    const uint64_t k = perm[j];                                    // row-major rank of that surfel
    AobSurfel sf = surfels[k];
    const f3 P = make_f3(sf.position[0], sf.position[1], sf.position[2]);
    const f3 N = make_f3(sf.normal[0], sf.normal[1], sf.normal[2]);
    f3 X = make_f3(1.0f, 0.0f, 0.0f), Z = make_f3(0.0f, 0.0f, 1.0f);
    if (N.y < 0.9999f && N.y > -0.9999f) {
        X = normalize(make_f3(-N.z, 0.0f, N.x));          // cross(N, (0,1,0))
        Z = normalize(cross(X, N));
    }
    bool valid = sf.texel_x >= 0 && sf.texel_y >= 0 && sf.texel_x < W && sf.texel_y < H &&
                 finite3(P) && finite3(N) && finite3(X) && finite3(Z);
    uint32_t texel = (uint32_t)sf.texel_y * (uint32_t)W + (uint32_t)sf.texel_x;
    uint32_t table_base = (section_index[k] % chunk_max) * spp;
    float4* out = frames + 4ull * j;
    out[0] = make_float4(P.x, P.y, P.z, N.x);
    out[1] = make_float4(N.y, N.z, X.x, X.y);
    out[2] = make_float4(X.z, Z.x, Z.y, Z.z);
    out[3] = make_float4(__uint_as_float(texel), __uint_as_float(table_base), __uint_as_float(valid ? 1u : 0u), 0.0f);
}

static int ensure_surfel_capacity(AobBake* b, DeviceMaterial& m, uint64_t need)
{
    if (need <= m.surfel_capacity) return AOB_OK;
    uint64_t cap = m.surfel_capacity ? m.surfel_capacity : (uint64_t)m.width * m.height;
    while (cap < need) cap *= 2;
    AobSurfel* ns = nullptr; uint32_t* ni = nullptr; float4* nf = nullptr; uint32_t* np = nullptr;
    {
        // all four or none: a failed allocation must not leak the ones before it
        hipError_t e = hipMalloc((void**)&ns, sizeof(AobSurfel) * cap);
        if (e == hipSuccess) e = hipMalloc((void**)&ni, sizeof(uint32_t) * cap);
        if (e == hipSuccess) e = hipMalloc((void**)&nf, sizeof(float4) * 4 * cap);
        if (e == hipSuccess) e = hipMalloc((void**)&np, sizeof(uint32_t) * cap);
        if (e != hipSuccess) {
            hipFree(ns); hipFree(ni); hipFree(nf); hipFree(np);
            set_error("growing the surfel buffers to %llu entries failed: %s", (unsigned long long)cap, hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? AOB_ERR_OUT_OF_MEMORY : AOB_ERR_DEVICE;
        }
    }
    {
        // the sections rasterised so far move over; the old buffers may still be read by work in flight, so the stream drains
        // before they are released.  A failure here frees the new buffers like a failed allocation does.
        hipError_t e = hipSuccess;
        if (m.surfel_count) {
            e = hipMemcpyAsync(ns, m.surfels, sizeof(AobSurfel) * m.surfel_count, hipMemcpyDeviceToDevice, b->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(ni, m.section_index, sizeof(uint32_t) * m.surfel_count, hipMemcpyDeviceToDevice, b->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(nf, m.frames, sizeof(float4) * 4 * m.surfel_count, hipMemcpyDeviceToDevice, b->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(np, m.perm, sizeof(uint32_t) * m.surfel_count, hipMemcpyDeviceToDevice, b->stream);      // (aob_bake_copy_horizon hands the tables out in surfel order)
        }
        if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
        if (e != hipSuccess) {
            hipFree(ns); hipFree(ni); hipFree(nf); hipFree(np);
            set_error("moving the surfel buffers failed: %s", hipGetErrorString(e));
            return AOB_ERR_DEVICE;
        }
    }
    release_device(b, m.surfels); release_device(b, m.section_index); release_device(b, m.frames); release_device(b, m.perm);   // (slices of the arena stay)
    m.surfels = ns; m.section_index = ni; m.frames = nf; m.perm = np; m.surfel_capacity = cap;
    return AOB_OK;
}

int rasterize_material(AobBake* b, int32_t material, int32_t row_begin, int32_t row_end)
{
    DeviceMaterial& m = b->materials[material];
    m.window_rows = row_end - row_begin; m.window_row0 = row_begin;
    hipStream_t s = b->stream;
    const uint64_t cells = (uint64_t)m.width * m.height;
    m.surfel_count = 0;
    m.section_offsets.clear(); m.section_mesh.clear();
    m.section_offsets.push_back(0);
    m.tile_prefix_valid = false;          // tile_row_prefix_host is written by the first section that has surfels
    AOB_HIP(hipMemsetAsync(m.results, 0, sizeof(AobResult) * cells, s));
    m.sections_overlap = false;
    // Everything after the coverage pass works on the strip of texel rows alone: the strip starts on a tile row, so it is an
    // atlas of its own — W x window_rows texels at cover + row_begin * W — and a rank of the N-GPU run does not walk the
    // other ranks' rows.
    const int32_t hw = row_end - row_begin;
    const uint64_t cell0 = (uint64_t)row_begin * m.width, cells_w = (uint64_t)hw * m.width;
    uint32_t* const cover_w = m.cover + cell0;
    const uint32_t nblocks = max(1u, (uint32_t)((cells_w + kScanTile - 1) / kScanTile));     // (an empty strip still runs the scans, on nothing)
    AOB_HIP(hipMemsetAsync(m.seen + cell0, 0, cells_w, s));
    AOB_HIP(hipMemsetAsync(m.block_sums + nblocks + 1, 0, sizeof(uint32_t), s));
    const bool conservative = b->opt.raster_mode == AOB_RASTER_CONSERVATIVE;
    const bool peel = b->opt.raster_layers == AOB_LAYERS_ALL;
    if (peel && !m.prev_layer) AOB_HIP(hipMalloc((void**)&m.prev_layer, sizeof(uint32_t) * cells));
    for (size_t mi = 0; mi < b->meshes.size(); mi++) {
        const DeviceMesh& mesh = b->meshes[mi];
        if (mesh.material != material) continue;                       // BakeEngine.cpp:523
        if (peel) AOB_HIP(hipMemsetAsync(m.prev_layer + cell0, 0, sizeof(uint32_t) * cells_w, s));
        // one pass per section — or, AOB_LAYERS_ALL, one per layer of the section's stencil peel (BakeEngine.cpp:531-532),
        // each layer a section of its own from here on
        for (int layer = 0; layer < (peel ? 255 : 1); layer++) {
        AOB_HIP(hipMemsetAsync(cover_w, 0, sizeof(uint32_t) * cells_w, s));
        if (mesh.triangle_count) {
            uint32_t grid = (mesh.triangle_count + kBlock - 1) / kBlock;
            if (peel)
                hipLaunchKernelGGL((k_cover<false, true>), dim3(grid), dim3(kBlock), 0, s, mesh.uvs, mesh.indices, mesh.triangle_count,
                                   m.width, m.height, row_begin, row_end, m.cover, m.prev_layer);
            else if (conservative)
                hipLaunchKernelGGL((k_cover<true, false>), dim3(grid), dim3(kBlock), 0, s, mesh.uvs, mesh.indices, mesh.triangle_count,
                                   m.width, m.height, row_begin, row_end, m.cover, (const uint32_t*)nullptr);
            else
                hipLaunchKernelGGL((k_cover<false, false>), dim3(grid), dim3(kBlock), 0, s, mesh.uvs, mesh.indices, mesh.triangle_count,
                                   m.width, m.height, row_begin, row_end, m.cover, (const uint32_t*)nullptr);
        }
        hipLaunchKernelGGL(k_scan_count, dim3(nblocks), dim3(kBlock), 0, s, cover_w, cells_w, m.block_sums, m.seen + cell0, m.block_sums + nblocks + 1);
        hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(kBlock), 0, s, m.block_sums, nblocks);
        // the section's surfel count decides the launch geometry of everything downstream (with it: whether any texel so far has
        // surfels of two sections)
        AOB_HIP(hipMemcpyAsync(&b->pinned[2], m.block_sums + nblocks, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));     // ([1] is the tree depth's)
        AOB_HIP(hipStreamSynchronize(s));
        const uint64_t count = b->pinned[2];
        m.sections_overlap = b->pinned[3] != 0u;
        if (peel && layer > 0 && count == 0) break;                    // the peel is through (the first layer stands for the section even when empty)
        AOB_TRY(ensure_surfel_capacity(b, m, m.surfel_count + count));
        if (count) {
            MeshXform xf;
            for (int k = 0; k < 16; k++) xf.m[k] = mesh.model[k];
            if (peel)
                hipLaunchKernelGGL((k_emit_surfels<false, true>), dim3(nblocks), dim3(kBlock), 0, s, cover_w, cells_w, m.block_sums,
                                   mesh.positions, mesh.normals, mesh.uvs, mesh.indices, xf, m.width, m.height, row_begin,
                                   b->opt.normal_fp16, m.surfels + m.surfel_count, m.section_index + m.surfel_count, m.prev_layer + cell0);
            else if (conservative)
                hipLaunchKernelGGL((k_emit_surfels<true, false>), dim3(nblocks), dim3(kBlock), 0, s, cover_w, cells_w, m.block_sums,
                                   mesh.positions, mesh.normals, mesh.uvs, mesh.indices, xf, m.width, m.height, row_begin,
                                   b->opt.normal_fp16, m.surfels + m.surfel_count, m.section_index + m.surfel_count, (uint32_t*)nullptr);
            else
                hipLaunchKernelGGL((k_emit_surfels<false, false>), dim3(nblocks), dim3(kBlock), 0, s, cover_w, cells_w, m.block_sums,
                                   mesh.positions, mesh.normals, mesh.uvs, mesh.indices, xf, m.width, m.height, row_begin,
                                   b->opt.normal_fp16, m.surfels + m.surfel_count, m.section_index + m.surfel_count, (uint32_t*)nullptr);
            // trace order: 8x8 tiles (perm), then the per-surfel trace records in that order
            const uint32_t tiles_x = (uint32_t)(m.width + 7) / 8, tiles_y = (uint32_t)(hw + 7) / 8;      // of the strip
            const uint64_t total_e = (uint64_t)tiles_x * tiles_y * 64u;
            const uint32_t tblocks = (uint32_t)((total_e + kScanTile - 1) / kScanTile);
            hipLaunchKernelGGL(k_tile_count, dim3(tblocks), dim3(kBlock), 0, s, cover_w, m.width, hw, tiles_x, total_e, m.tile_sums);
            hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(kBlock), 0, s, m.tile_sums, tblocks);
            hipLaunchKernelGGL(k_tile_emit, dim3(tblocks), dim3(kBlock), 0, s, cover_w, m.width, hw, tiles_x, total_e, m.tile_sums,
                               m.perm + m.surfel_count);
            if (m.section_offsets.size() == 1) {
                hipLaunchKernelGGL(k_tile_row_prefix, dim3((tiles_y + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, s, cover_w, m.width, hw,
                                   tiles_x, tiles_y, m.tile_sums, (uint32_t)count, m.tile_row_prefix);
                AOB_HIP(hipMemcpyAsync(m.tile_row_prefix_host, m.tile_row_prefix, sizeof(uint32_t) * (tiles_y + 1), hipMemcpyDeviceToHost, s));
                m.tile_prefix_valid = true;
            }
            hipLaunchKernelGGL(k_surfel_frames, dim3((uint32_t)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                               m.surfels + m.surfel_count, m.section_index + m.surfel_count, m.perm + m.surfel_count, count, m.width, m.height,
                               (uint32_t)chunk_texel_max(b->opt), (uint32_t)b->opt.sample_count,
                               m.frames + 4ull * m.surfel_count);
        }
        m.surfel_count += count;
        m.section_offsets.push_back(m.surfel_count);
        m.section_mesh.push_back((uint32_t)mi);
        }   // layers
    }
    AOB_HIP(hipGetLastError());
    m.rasterized = true;
    // the tile cuts of this material's trace launches, sized by what was covered (a strip of rows: a strip's worth) — here, so that no trace
    // launch carries an allocation; without the memory the launches walk from the root and the bake says so (trace.hip ensure_tile_cuts)
    m.cuts_state = 0u; m.horizon_state = 0u; m.tile_cut_failed = 0;      // (per material: another material's rasterisation does not erase what this one's launches reported)
    m.horizon_asked = 0;
    (void)hipMemsetAsync(b->horizon_counters + 16 + material, 0, sizeof(unsigned long long), b->stream);
    reserve_tile_cuts(b, material);
    return AOB_OK;
}

} // namespace aob

// horizon_kernel.h — the device side of horizon.hip (see there), apart so that tests/horizon_emu/horizon_emu.cpp can run it on the CPU, a thread per lane.
#pragma once
#include "aob_math.h"

#include <cmath>
#include <cstring>

namespace aob {

static constexpr int kHzWave = 64;
#ifndef AOB_HZ_GROUPS
#define AOB_HZ_GROUPS 8
#endif
static constexpr int kHzGroups = AOB_HZ_GROUPS;              // groups of 64 / kHzGroups consecutive origins a middle-distance leaf is measured against (16: tighter, twice the work)
static constexpr int kHzSectors = 32;            // sectors of the tile's horizon H
#ifndef AOB_HZ_OPEN_CAP
#define AOB_HZ_OPEN_CAP 512
#endif
#ifndef AOB_HZ_NEAR_CAP
#define AOB_HZ_NEAR_CAP 384
#endif
// (nothing here is part of the arithmetic contract of DESIGN.md §4 — the tables are bounds with margins, not results — so multiply-adds may fuse: a tenth of the kernel's instructions)
#if defined(__clang__) && !defined(AOB_HZ_NO_CONTRACT)
#pragma clang fp contract(fast)
#endif
#ifndef AOB_HZ_LANE_CHILD
#define AOB_HZ_LANE_CHILD 1        // phase A: 1 a lane a child box (16 nodes a round), 0 a lane a node (64 a round, its children one after the other)
#endif
#ifndef AOB_HZ_LEAFQ
#define AOB_HZ_LEAFQ (AOB_HZ_LANE_CHILD ? 128 : 512)
#endif
#ifndef AOB_HZ_WAVES
#define AOB_HZ_WAVES 0
#endif
#if AOB_HZ_WAVES
#define AOB_HZ_BOUNDS __launch_bounds__(kHzWave, AOB_HZ_WAVES)
#else
#define AOB_HZ_BOUNDS __launch_bounds__(kHzWave)
#endif
static constexpr int kHzLeafQ = AOB_HZ_LEAFQ;        // middle-distance leaves a round can put aside (LDS); one more raises H by its box, as a far leaf does
static constexpr int kHzOpenCap = AOB_HZ_OPEN_CAP;   // open nodes of the walk (LDS); a node that does not fit raises H by its own box instead
static constexpr int kHzNearCap = AOB_HZ_NEAR_CAP;           // near triangles per tile; more: the tile gets no culling (every cell blocked)
static constexpr int kHzMaxRounds = 1024;
static constexpr uint32_t kHzEmptyPlanes = 0x8000ffffu;
static constexpr float kMarginU = 2e-3f;         // on u1 (a column of a 32-column grid is 0.031 wide)
static constexpr float kMarginPhi = 1e-3f;       // on azimuths, in turns (0.36 degrees)
static constexpr float kMarginF = 1e-3f;         // on sines / cosines of elevations

#ifdef AOB_HORIZON_EMU                       // tests/horizon_emu/horizon_emu.cpp: the kernel on the CPU, a thread per lane
#define HZ_CLOCK() 0ull
#define HZ_SCALAR_POINTER(p) ((void)0)
typedef const float* hz_const_floats;
#else
#define HZ_CLOCK() ((unsigned long long)__builtin_readcyclecounter())
#define HZ_SCALAR_POINTER(p) asm volatile("" : "+s"(p))
typedef const __attribute__((address_space(4))) float* hz_const_floats;
#endif
#ifndef AOB_HZ_SKIP
#define AOB_HZ_SKIP 0                        // measurement builds (tools/build_variants.py): bit 0 no near phase, bit 1 no arcs, bit 2 no middle-leaf measuring, bit 3 no walk at all
#endif
#define HZ_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)

struct HorizonParams {
    const float4* frames;          // 4 x float4 per surfel (k_surfel_frames), trace order
    uint64_t begin, end;           // the launch's surfels; tile t = [begin + 64 t, begin + 64 t + 64)
    const uint4* nodes4q;
    const float4* leaf_tris;
    const float* quant_frame;
    uint32_t node_count;
    int32_t K, C;                  // azimuth rows (= the grid's sy) and columns (= sx) of the local tables; both <= 32
    float bias, reach;             // ray_bias; the farthest a hit can be from its surfel (tmax + bias, a little more), INFINITY: unbounded
    float near_factor, mid_factor; // r_near, r_mid in diagonals of the tile's box
    float small2;                  // a box whose angular radius squared is below this is not opened
    float big2;                    // a leaf whose box diagonal squared exceeds this times its squared gap to the tile is a near leaf
    uint8_t* out;                  // stride bytes per surfel: free columns of row cy in byte cy
    uint32_t stride;
    uint32_t tile_stride;          // the launch's block i measures tile i * tile_stride (1: every tile; more: a sample of them)
    unsigned long long* skipped;   // (or null) += the free cells of the launch's tables, += 64 per tile at [1]: what a sample of tiles promises
    unsigned long long* counters;  // diagnostics (or null): rounds, box tests, near triangles, mid leaves, tiles given up, tiles, clock ticks of phases A / B / C
#ifdef AOB_HORIZON_EMU
    float* dbg;                    // per tile 32 floats: the tile's horizon after phase A; then per tile 64 x 32 words: the local tables after phase B
    uint32_t* dbg_tables;
#endif
};

__device__ __forceinline__ float hz_uni(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
__device__ __forceinline__ float hz_rcp(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float hz_rsq(float v) { return __builtin_amdgcn_rsqf(v); }
// sqrt from above / from below (v_sqrt_f32 is good to 1 ulp; the file's correctly rounded sqrtf is a dozen instructions)
__device__ __forceinline__ float hz_sqrt_up(float v) { return __builtin_amdgcn_sqrtf(v) * 1.0000004f; }
__device__ __forceinline__ float hz_sqrt_down(float v) { return __builtin_amdgcn_sqrtf(v) * 0.9999996f; }

// atan2(y, x) in turns, (-0.5, 0.5]; |error| < 4e-6 turns (Abramowitz & Stegun 4.4.49 on the octant's ratio)
__device__ __forceinline__ float hz_atan2_turns(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float a = mx > 0.0f ? mn * hz_rcp(mx) : 0.0f;
    const float s = a * a;
    float r = a * (0.9998660f + s * (-0.3302995f + s * (0.1801410f + s * (-0.0851330f + s * 0.0208351f))));
    r *= 0.15915494f;
    if (ay > ax) r = 0.25f - r;
    if (x < 0.0f) r = 0.5f - r;
    if (y < 0.0f) r = -r;
    return r;
}

// asin(x) / 2 pi from above, x in [0, 1]: asin x <= x + (pi/2 - 1) x^3
__device__ __forceinline__ float hz_asin_turns_up(float x)
{
    x = fminf(x, 1.0f);
    return (x + 0.5707964f * x * x * x) * 0.15915494f + 1e-6f;
}

// the sectors [lo, hi] (turns, hi >= lo) of a K-sector circle touch, as a bit mask
__device__ __forceinline__ uint32_t hz_sector_mask(float lo, float hi, bool full, int K, float margin)
{
    const uint32_t all = K >= 32 ? 0xffffffffu : (1u << K) - 1u;
    const float fK = (float)K;
    const int a = (int)floorf((lo - margin) * fK), b = (int)floorf((hi + margin) * fK);
    const int cnt = b - a + 1;
    if (full || cnt >= K || !(hi >= lo)) return all;
    int s = a;
    if (s < 0) s += K;
    if (s < 0) s += K;
    if (s >= K) s -= K;
    if (s >= K) s -= K;
    if (s < 0 || s >= K) return all;                 // (angles beyond two turns: never, but never wrong)
    const uint32_t base = (1u << cnt) - 1u;          // cnt <= K - 1 <= 31
    uint32_t m = base << s;
    if (s) m |= base >> (K - s);
    return m & all;
}

// first sector and sector count of [lo, hi] on the tile's 32-sector horizon (count 32: all)
__device__ __forceinline__ uint32_t hz_sector_range32(float lo, float hi, bool full)
{
    const int a = (int)floorf((lo - kMarginPhi) * 32.0f), b = (int)floorf((hi + kMarginPhi) * 32.0f);
    int cnt = b - a + 1;
    if (full || cnt >= 32 || !(hi >= lo)) return 32u << 8;
    return (uint32_t)(a & 31) | ((uint32_t)cnt << 8);
}

__device__ __forceinline__ float hz_dot(const f3& a, const f3& b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }   // (aob_math.h's dot() keeps the contract's unfused products)
__device__ __forceinline__ float hz_wave_min(float v) { for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off)); return v; }
__device__ __forceinline__ float hz_wave_max(float v) { for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off)); return v; }
__device__ __forceinline__ float hz_wave_sum(float v) { for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off); return v; }

struct HzLeaf { uint32_t ref, key; };                 // key: first sector | sectors << 5 | margin (sin, 1 / 16384ths, rounded up) << 11
struct HzWalk { uint32_t open_ref[kHzOpenCap]; uint32_t open_key[kHzOpenCap]; HzLeaf leafq[kHzLeafQ]; };
// phase A's lists are dead when phases B and C start: the local tables and phase C's arcs live in their storage (10.1 -> 8.1 KB of LDS per tile for 16-row
// tables: five wavefronts per SIMD instead of four — the kernel waits on its own round trips, and its time follows the wavefronts a SIMD holds: 9.9 / 7.7 / 6.0 ms at 2 / 3 / 4)
template <int CMAX, bool T16> struct HzLate { float arcs[kHzWave][12]; uint32_t table[CMAX * kHzWave / (T16 ? 2 : 1)]; };
template <int CMAX, bool T16> union HzScratch { HzWalk walk; HzLate<CMAX, T16> late; };

template <int CMAX, bool T16>
__global__ AOB_HZ_BOUNDS
void k_horizon(HorizonParams p)
{
    __shared__ HzScratch<CMAX, T16> scratch;         // phase A's lists, then the local tables and phase C's arcs
    uint32_t* const open_ref = scratch.walk.open_ref;   // phase A: the nodes to open (bit 31: a near node), last in first out
    uint32_t* const open_key = scratch.walk.open_key;   // ... and what the box they came from measured: top (16 bits, rounded up) | first sector << 16 | sectors << 21
    HzLeaf* const leafq = scratch.walk.leafq;        // a round's middle-distance leaves
    float (* const arcs)[12] = scratch.late.arcs;         // phase C: b, n, n x a, b x n of every arc (a = the arc before's b)
    __shared__ uint32_t near_tri[kHzNearCap];
    __shared__ uint32_t H[kHzSectors];               // bits of (sin E + 2): positive floats order like integers
    __shared__ float grp[kHzGroups][8];              // the boxes of the tile's groups of consecutive origins, in the tile's frame: lo xyz, hi xyz, valid
    uint32_t* const table = scratch.late.table;      // (phases B and C) local tables: [column][lane] = rows blocked FROM that column (cumulated at the end); T16: 16 rows, two lanes a word

    const uint32_t lane = threadIdx.x;
    const uint64_t k = p.begin + (uint64_t)blockIdx.x * p.tile_stride * kHzWave + lane;
    const int K = p.K, C = p.C;
    if (AOB_HZ_SKIP & 16) return;

    // ---- the surfel and its frame's dual basis ----
    f3 P = make_f3(0.f, 0.f, 0.f), Xd = P, Nd = P, Zd = P, Nw = P;
    bool framed = false;
    if (k < p.end) {
        const float4 a = p.frames[4ull * k], b = p.frames[4ull * k + 1], c = p.frames[4ull * k + 2], d = p.frames[4ull * k + 3];
        const bool valid = __float_as_uint(d.z) != 0u;
        P = make_f3(a.x, a.y, a.z);
        const f3 N = make_f3(a.w, b.x, b.y), X = make_f3(b.z, b.w, c.x), Z = make_f3(c.y, c.z, c.w);
        const f3 nz = cross(N, Z), zx = cross(Z, X), xn = cross(X, N);
        const float det = hz_dot(X, nz);
        const float nn = hz_dot(N, N);
        if (valid && fabsf(det) > 1e-12f && det == det && nn > 1e-20f) {
            const float id = 1.0f / det;
            Xd = make_f3(nz.x * id, nz.y * id, nz.z * id);
            Nd = make_f3(zx.x * id, zx.y * id, zx.z * id);
            Zd = make_f3(xn.x * id, xn.y * id, xn.z * id);
            const float il = 1.0f / sqrtf(nn);
            Nw = make_f3(N.x * il, N.y * il, N.z * il);
            // (the bounds below want the duals of a frame that is orthonormal up to rounding and to the 0.0001 by which an axis frame's normal may lean:
            // anything else — a frame the sampler itself would make garbage of — gets no culling)
            const float lx = hz_dot(Xd, Xd), lz = hz_dot(Zd, Zd), ln = hz_dot(Nd, Nd) * nn;
            framed = finite3(Xd) && finite3(Nd) && finite3(Zd) && lx > 0.9f && lx < 1.1f && lz > 0.9f && lz < 1.1f && ln > 0.9f && ln < 1.1f;
        }
    }
    const uint64_t framed_m = __ballot(framed);
    if (AOB_HZ_SKIP & 32) { if (framed && k == 0xffffffffffull) p.out[0] = 1; return; }
    auto write_out = [&](const uint32_t* fr) {          // fr: stride / 4 words of free counts
        if (k < p.end) {
            uint32_t* dst = (uint32_t*)(p.out + k * (uint64_t)p.stride);
            for (uint32_t w = 0; w < p.stride / 4u; w++) dst[w] = fr[w];
        }
    };
    uint32_t zero8[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (framed_m == 0ull || p.node_count == 0u) {
        // no frame: nothing to cull for.  (No scene: every ray misses — but the kernel that traces them says so just as fast.)
        write_out(zero8);
        return;
    }

    // ---- the tile: its box, its frame (pole = the mean normal), its extent in that frame, the boxes of its 16 groups of four origins ----
    const float big = 3.0e38f;
    f3 tlo = make_f3(hz_wave_min(framed ? P.x : big), hz_wave_min(framed ? P.y : big), hz_wave_min(framed ? P.z : big));
    f3 thi = make_f3(hz_wave_max(framed ? P.x : -big), hz_wave_max(framed ? P.y : -big), hz_wave_max(framed ? P.z : -big));
    f3 Nt = make_f3(hz_wave_sum(framed ? Nw.x : 0.f), hz_wave_sum(framed ? Nw.y : 0.f), hz_wave_sum(framed ? Nw.z : 0.f));
    {
        const float l2 = hz_dot(Nt, Nt);
        if (l2 > 1e-12f) { const float il = 1.0f / sqrtf(l2); Nt = make_f3(Nt.x * il, Nt.y * il, Nt.z * il); }
        else Nt = make_f3(0.f, 1.f, 0.f);
    }
    f3 Xt, Zt;
    if (Nt.y < 0.9999f && Nt.y > -0.9999f) {
        Xt = normalize(make_f3(-Nt.z, 0.0f, Nt.x));
        Zt = normalize(cross(Xt, Nt));
    } else {
        Nt = make_f3(0.f, Nt.y > 0.f ? 1.f : -1.f, 0.f);
        Xt = make_f3(1.f, 0.f, 0.f);
        Zt = cross(Xt, Nt);
    }
    tlo = make_f3(hz_uni(tlo.x), hz_uni(tlo.y), hz_uni(tlo.z)); thi = make_f3(hz_uni(thi.x), hz_uni(thi.y), hz_uni(thi.z));
    Xt = make_f3(hz_uni(Xt.x), hz_uni(Xt.y), hz_uni(Xt.z)); Nt = make_f3(hz_uni(Nt.x), hz_uni(Nt.y), hz_uni(Nt.z)); Zt = make_f3(hz_uni(Zt.x), hz_uni(Zt.y), hz_uni(Zt.z));
    const f3 fP = make_f3(hz_dot(Xt, P), hz_dot(Nt, P), hz_dot(Zt, P));          // the origin in the tile's frame
    f3 flo = make_f3(hz_wave_min(framed ? fP.x : big), hz_wave_min(framed ? fP.y : big), hz_wave_min(framed ? fP.z : big));
    f3 fhi = make_f3(hz_wave_max(framed ? fP.x : -big), hz_wave_max(framed ? fP.y : -big), hz_wave_max(framed ? fP.z : -big));
    {
        // (a hair of slack on the extents: the products above are rounded)
        const float slack = 1e-6f * (fabsf(flo.x) + fabsf(fhi.x) + fabsf(flo.y) + fabsf(fhi.y) + fabsf(flo.z) + fabsf(fhi.z)) + 1e-30f;
        flo = make_f3(hz_uni(flo.x - slack), hz_uni(flo.y - slack), hz_uni(flo.z - slack));
        fhi = make_f3(hz_uni(fhi.x + slack), hz_uni(fhi.y + slack), hz_uni(fhi.z + slack));
        float glo[3] = { framed ? fP.x - slack : big, framed ? fP.y - slack : big, framed ? fP.z - slack : big };
        float ghi[3] = { framed ? fP.x + slack : -big, framed ? fP.y + slack : -big, framed ? fP.z + slack : -big };
#pragma unroll
        for (int q = 0; q < 3; q++) {
            for (int off = 1; off < kHzWave / kHzGroups; off <<= 1) { glo[q] = fminf(glo[q], __shfl_xor(glo[q], off)); ghi[q] = fmaxf(ghi[q], __shfl_xor(ghi[q], off)); }
        }
        constexpr uint32_t per = kHzWave / kHzGroups;
        if ((lane & (per - 1u)) == 0u) {
            float* g = grp[lane / per];
            g[0] = glo[0]; g[1] = glo[1]; g[2] = glo[2]; g[3] = ghi[0]; g[4] = ghi[1]; g[5] = ghi[2];
            g[6] = ((framed_m >> lane) & ((1ull << per) - 1ull)) ? 1.0f : 0.0f; g[7] = 0.0f;
        }
    }
    const f3 fc = make_f3(0.5f * (flo.x + fhi.x), 0.5f * (flo.y + fhi.y), 0.5f * (flo.z + fhi.z));
    const f3 te = make_f3(0.5f * (fhi.x - flo.x), 0.5f * (fhi.y - flo.y), 0.5f * (fhi.z - flo.z));
    const f3 aXt = make_f3(fabsf(Xt.x), fabsf(Xt.y), fabsf(Xt.z)), aNt = make_f3(fabsf(Nt.x), fabsf(Nt.y), fabsf(Nt.z)), aZt = make_f3(fabsf(Zt.x), fabsf(Zt.y), fabsf(Zt.z));
    const float diag2 = (thi.x - tlo.x) * (thi.x - tlo.x) + (thi.y - tlo.y) * (thi.y - tlo.y) + (thi.z - tlo.z) * (thi.z - tlo.z);
    const float r_near2 = p.near_factor * p.near_factor * diag2, r_mid2 = p.mid_factor * p.mid_factor * diag2;
    const float reach2 = p.reach * p.reach;                    // (inf * inf = inf)
    const f3 qo = make_f3(p.quant_frame[0], p.quant_frame[1], p.quant_frame[2]), qs = make_f3(p.quant_frame[3], p.quant_frame[4], p.quant_frame[5]);

    if (AOB_HZ_SKIP & 64) { if (r_near2 + r_mid2 + fc.x + te.x + aXt.x + qo.x + qs.x == 12345.678f) p.out[0] = 1; return; }
    if (lane < (uint32_t)kHzSectors) H[lane] = __float_as_uint(1.0f);          // sin E = -1: nothing blocked
    if (lane == 0) { open_ref[0] = 0x80000000u; open_key[0] = 0xffffffffu; }       // the root, as a near node: opened whatever the horizon
    HZ_SYNC();

    // ---- phase A ----
    const unsigned long long clk0 = p.counters ? HZ_CLOCK() : 0ull;
    int open_n = 1, near_n = 0, rounds = 0;
    bool give_up = false;
    unsigned long long n_tests = 0, n_leaves = 0;
    auto key_top = [](uint32_t key) { return (float)(key & 0xffffu) * (1.0f / 32767.0f) - 1.0f; };   // the sin E a key stands for (rounded up when it was made)
    // does [first, first + cnt) hold a sector whose horizon is below `top`?  Up to four sectors are looked at one by one; a wider range is held
    // against the lowest horizon of all sectors as the round found it (an older, lower horizon only keeps more: conservative either way).
    uint32_t h_low = __float_as_uint(1.0f);
    auto over_horizon = [&](float top, uint32_t rng, bool want) -> bool {
        const uint32_t first = rng & 31u, cnt = rng >> 8;
        const uint32_t tb = __float_as_uint(top + 2.0f);
        const uint32_t h0 = H[first], h1 = H[(first + 1u) & 31u], h2 = H[(first + 2u) & 31u], h3 = H[(first + 3u) & 31u];
        const uint32_t few = min(min(h0, cnt > 1u ? h1 : 0xffffffffu), min(cnt > 2u ? h2 : 0xffffffffu, cnt > 3u ? h3 : 0xffffffffu));
        return want && (cnt <= 4u ? few : h_low) < tb;
    };
    // raise the horizon of [first, first + cnt) to `top`: a narrow range sector by sector (most ranges are one to three sectors wide), a wide one —
    // a box close to the tile — by the first 32 lanes together, one such box after the other
    auto raise = [&](float top, uint32_t rng, bool want) {
        const uint32_t first = rng & 31u, cnt = rng >> 8;
        const uint32_t tb = __float_as_uint(fminf(top, 1.0f) + 2.0f);
        const bool narrow = want && cnt <= 4u;
        if (__ballot(narrow) != 0ull) {
#pragma unroll
            for (uint32_t i = 0; i < 4u; i++) if (narrow && i < cnt) atomicMax(&H[(first + i) & 31u], tb);
        }
        for (uint64_t wide = __ballot(want && cnt > 4u); wide; wide &= wide - 1ull) {
            const int l = (int)__builtin_ctzll(wide);
            const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)first, l), n = (uint32_t)__builtin_amdgcn_readlane((int)cnt, l), t = (uint32_t)__builtin_amdgcn_readlane((int)tb, l);
            if (lane < 32u && ((lane - f) & 31u) < n) atomicMax(&H[lane], t);
        }
    };
    int n_leafq = 0;                 // middle-distance leaves put aside: measured when a wavefront's worth has come together (or the walk is over)
    constexpr int kLeafFlush = kHzLeafQ - kHzWave * (AOB_HZ_LANE_CHILD ? 1 : 4);      // (a round adds at most a leaf a child)
    static_assert(kLeafFlush >= 0, "AOB_HZ_LEAFQ: room for a round's leaves");
    while (open_n > 0 && !(AOB_HZ_SKIP & 8)) {
        if (++rounds > kHzMaxRounds) break;
        {
            uint32_t hv = H[lane & 31u];
            for (int off = 16; off > 0; off >>= 1) hv = min(hv, (uint32_t)__shfl_xor((int)hv, off));
            h_low = (uint32_t)__builtin_amdgcn_readfirstlane((int)hv);
        }
        // ---- a round: the nodes of the list's top, up to 16, four lanes a node, every lane one child box ----
        // (the walk's list is short at its start and at its end: with a lane a NODE and its children one after the other — 64 nodes a round — 36 lanes of 64 had
        // one on the terrain, and every lane went through the ~450 instructions of a box four times a round; a lane a CHILD fills the wavefront from 16 nodes)
        // (as many as the list can take the children of: a node leaves the list and up to four enter it)
        constexpr int kPer = AOB_HZ_LANE_CHILD ? 4 : 1;          // lanes a node
        const int slot = (int)lane / kPer;
        const int take = min(min(kHzWave / kPer, open_n), max(1, (kHzOpenCap - open_n) / 3));
        bool have = slot < take;
        uint32_t node = 0u, key = 0u;
        if (have) { node = open_ref[open_n - 1 - slot]; key = open_key[open_n - 1 - slot]; }
        open_n -= take;
        const bool near_node = (node >> 31) != 0u;
        node &= 0x7fffffffu;
        // (the horizon has risen since the node was put aside: is its box still over it?)
        if (__ballot(have && !near_node) != 0ull) {
            const bool still = over_horizon(key_top(key), ((key >> 16) & 31u) | (((key >> 21) & 63u) << 8), have && !near_node);
            if (have && !near_node && !still) have = false;
        }
        uint4 kid[4];
#pragma unroll
        for (int ci = 0; ci < 4 / kPer; ci++) kid[ci] = have ? p.nodes4q[4ull * node + (kPer == 4 ? (lane & 3u) : (uint32_t)ci)] : make_uint4(kHzEmptyPlanes, kHzEmptyPlanes, kHzEmptyPlanes, 0u);
        for (int ci = 0; ci < 4 / kPer; ci++) {
            const uint4 c = kid[ci];
            bool live = have && !(c.x == kHzEmptyPlanes && c.y == kHzEmptyPlanes && c.z == kHzEmptyPlanes);
            if (__ballot(live) == 0ull) continue;
            n_tests += (unsigned long long)__popcll(__ballot(live));
            // the child's box (stored planes: origin + q * scale, min rounded down, max up)
            const f3 lo = make_f3(fmaf((float)(c.x & 0x7fffu), qs.x, qo.x), fmaf((float)(c.y & 0x7fffu), qs.y, qo.y), fmaf((float)(c.z & 0x7fffu), qs.z, qo.z));
            const f3 hi = make_f3(fmaf((float)((c.x >> 16) & 0x7fffu), qs.x, qo.x), fmaf((float)((c.y >> 16) & 0x7fffu), qs.y, qo.y), fmaf((float)((c.z >> 16) & 0x7fffu), qs.z, qo.z));
            const float gx = fmaxf(fmaxf(lo.x - thi.x, tlo.x - hi.x), 0.0f), gy = fmaxf(fmaxf(lo.y - thi.y, tlo.y - hi.y), 0.0f), gz = fmaxf(fmaxf(lo.z - thi.z, tlo.z - hi.z), 0.0f);
            const float gap2 = gx * gx + gy * gy + gz * gz;
            if (live && gap2 > reach2 * 1.0001f) live = false;          // no ray of the tile reaches it
            const bool leaf = (c.w >> 31) != 0u;
            // near: within r_near of the tile — or a leaf that looks big from the tile (its box's diagonal against the gap): an edge seen under a wide
            // angle bulges far over its ends, and the tile-level bounds below would have to give up on it
            const float ldiag2 = (hi.x - lo.x) * (hi.x - lo.x) + (hi.y - lo.y) * (hi.y - lo.y) + (hi.z - lo.z) * (hi.z - lo.z);
            const bool near = live && (gap2 <= r_near2 || (leaf && ldiag2 > p.big2 * gap2));
            const bool far = live && !near;
            // ---- far boxes: top and arc in the tile's frame, the box grown by the tile's extent ----
            const f3 cw = make_f3(0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)), ew = make_f3(0.5f * (hi.x - lo.x), 0.5f * (hi.y - lo.y), 0.5f * (hi.z - lo.z));
            const f3 v = make_f3(hz_dot(Xt, cw) - fc.x, hz_dot(Nt, cw) - fc.y, hz_dot(Zt, cw) - fc.z);
            const float slack = 2e-6f * (fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(fc.x) + fabsf(fc.y) + fabsf(fc.z));
            const f3 e = make_f3(hz_dot(aXt, ew) + te.x + slack, hz_dot(aNt, ew) + te.y + slack, hz_dot(aZt, ew) + te.z + slack);
            const float a0 = v.x - e.x, a1 = v.x + e.x, b1 = v.y + e.y, g0 = v.z - e.z, g1 = v.z + e.z;
            const float rx = fmaxf(fmaxf(a0, -a1), 0.0f), rz = fmaxf(fmaxf(g0, -g1), 0.0f);
            const float rmin2 = rx * rx + rz * rz;
            const float fx = fmaxf(fabsf(a0), fabsf(a1)), fz = fmaxf(fabsf(g0), fabsf(g1));
            const float r2 = b1 > 0.0f ? rmin2 : fx * fx + fz * fz;
            const float den = b1 * b1 + r2;
            float top = den > 0.0f ? b1 * hz_rsq(den) : 1.0f;
            top = fminf(top + kMarginF, 1.0f);
            const bool around = rmin2 == 0.0f;                          // the grown box stands over (or under) the tile: every azimuth
            // the arc of azimuths: the grown rectangle's bounding circle about its centre (cheaper than its four corners, and the rectangle is nearly a square)
            const float rc2 = v.x * v.x + v.z * v.z, R2c = e.x * e.x + e.z * e.z;
            const float thc = hz_atan2_turns(v.z, v.x);
            const float alpha = (around || rc2 <= R2c) ? 0.5f : hz_asin_turns_up(hz_sqrt_up(R2c * hz_rcp(rc2)) * 1.0001f);
            uint32_t rng = hz_sector_range32(thc - alpha, thc + alpha, around || alpha >= 0.25f);
            const bool over = over_horizon(top, rng, far);
            const float ang2 = hz_dot(e, e) * hz_rcp(fmaxf(hz_dot(v, v), 1e-30f));
            const bool relevant = far && over;
            // (a leaf whose grown box stands over the tile may hold the pole of some origin in its INTERIOR, where no vertex or edge sees it: its box's top, 1, it is)
            const bool mid_leaf = relevant && leaf && gap2 <= r_mid2 && gap2 > 0.0f && !around;
            const bool box_apply = relevant && !mid_leaf && (leaf || ang2 <= p.small2);
            const bool push_far = relevant && !leaf && !box_apply;
            const bool push_near = near && !leaf;
            const bool near_leaf = near && leaf;
            const uint64_t below = (1ull << lane) - 1ull;
            // ---- near leaves: their triangles to the near list ----
            {
                const uint64_t m1 = __ballot(near_leaf), m2 = __ballot(near_leaf && (c.w & 0x40000000u));
                const int n1 = __popcll(m1), n2 = __popcll(m2);
                if (near_n + n1 + n2 > kHzNearCap) give_up = true;
                else {
                    if (near_leaf) near_tri[near_n + __popcll(m1 & below)] = c.w & 0x3fffffffu;
                    if (near_leaf && (c.w & 0x40000000u)) near_tri[near_n + n1 + __popcll(m2 & below)] = (c.w & 0x3fffffffu) + 1u;
                    near_n += n1 + n2;
                }
            }
            if (give_up) break;
            // ---- nodes to open: the far ones first, the near ones on top of them (last in, first out) ----
            {
                const uint32_t tq = (uint32_t)fminf(ceilf((top + 1.0f) * 32767.0f), 65535.0f);
                const uint32_t mykey = tq | ((rng & 31u) << 16) | ((rng >> 8) << 21);
                const uint64_t mf = __ballot(push_far), mn = __ballot(push_near);
                const int nf = __popcll(mf), nn = __popcll(mn);
                const int room = kHzOpenCap - open_n;
                bool spill_far = false;
                if (push_far) {
                    const int r = __popcll(mf & below);
                    if (r < room - nn) { open_ref[open_n + r] = c.w; open_key[open_n + r] = mykey; }
                    else spill_far = true;                                     // no room: the box itself raises the horizon
                }
                const int kept_f = max(0, min(nf, room - nn));
                if (nn > room) give_up = true;
                else if (push_near) { const int r = __popcll(mn & below); open_ref[open_n + kept_f + r] = c.w | 0x80000000u; open_key[open_n + kept_f + r] = 0xffffffffu; }
                open_n += kept_f + (give_up ? 0 : nn);
                if (__ballot(box_apply || spill_far) != 0ull) raise(top, rng, box_apply || spill_far);
            }
            if (give_up) break;
            // ---- middle-distance leaves: put aside, measured together below ----
            {
                const uint64_t ml = __ballot(mid_leaf);
                if (mid_leaf) {
                    // the arc's bulge over its ends: an edge of angular length w rises at most w^2 / 7 (in sine) over the higher end; w <= box diagonal / gap
                    // (1 / cos(w / 2) - 1 <= w^2 / 7 up to w = 1.2)
                    const float w2 = 4.0f * hz_dot(ew, ew) * hz_rcp(gap2) * 1.001f;
                    const float mg = (w2 <= 1.0f ? w2 * (1.0f / 7.0f) : 2.0f) + kMarginF;
                    HzLeaf lf; lf.ref = c.w; lf.key = (rng & 31u) | ((rng >> 8) << 5) | ((uint32_t)fminf(ceilf(mg * 16384.0f), 32768.0f) << 11);
                    if (n_leafq + __popcll(ml & below) < kHzLeafQ) leafq[n_leafq + __popcll(ml & below)] = lf;
                }
                if (n_leafq + __popcll(ml) > kHzLeafQ) {
                    const bool spill_leaf = mid_leaf && n_leafq + __popcll(ml & below) >= kHzLeafQ;
                    raise(top, rng, spill_leaf);
                }
                n_leafq = min(n_leafq + __popcll(ml), kHzLeafQ);
            }
        }
        if (give_up) break;
        // ---- the middle-distance leaves put aside, 16 at a time: four lanes a leaf, each lane its vertices against a quarter of the groups' boxes ----
        // (not round by round — five leaves a round on the terrain, a third of a batch — but when 64 and more wait, or nothing is left to open: until then the horizon
        // is the lower for it and keeps a box or two more; the bounds hold whenever they are applied)
        const bool flush = n_leafq > kLeafFlush || open_n == 0 || rounds == kHzMaxRounds;
        if (flush) n_leaves += (unsigned long long)n_leafq;
        if (n_leafq && flush && !(AOB_HZ_SKIP & 4)) {
            HZ_SYNC();
            for (int base = 0; base < n_leafq; base += 16) {
                const int j = base + (int)(lane >> 2), g0i = (int)(lane & 3u) * (kHzGroups / 4);
                const bool on = j < n_leafq;
                float el = -2.0f;
                HzLeaf q; q.ref = 0u; q.key = 0u;
                if (on) q = leafq[j];
                if (on) {
                    const uint32_t tri = q.ref & 0x3fffffffu;
                    const int nv = (q.ref & 0x40000000u) ? 6 : 3;
                    for (int vi = 0; vi < nv; vi++) {
                        const float4 w = p.leaf_tris[3ull * tri + vi];
                        const f3 f = make_f3(hz_dot(Xt, make_f3(w.x, w.y, w.z)), hz_dot(Nt, make_f3(w.x, w.y, w.z)), hz_dot(Zt, make_f3(w.x, w.y, w.z)));
                        const float sl = 2e-6f * (fabsf(f.x) + fabsf(f.y) + fabsf(f.z));
#pragma unroll
                        for (int gi = 0; gi < kHzGroups / 4; gi++) {
                            const float* gb = grp[g0i + gi];
                            const float lx = f.x - gb[3] - sl, hx = f.x - gb[0] + sl, lz = f.z - gb[5] - sl, hz = f.z - gb[2] + sl;
                            const float bh = f.y - gb[1] + sl;               // the highest the vertex stands over an origin of the group
                            const float da = fmaxf(fmaxf(lx, -hx), 0.0f), dg = fmaxf(fmaxf(lz, -hz), 0.0f);
                            const float fa = fmaxf(fabsf(lx), fabsf(hx)), fg = fmaxf(fabsf(lz), fabsf(hz));
                            const float rr = bh > 0.0f ? da * da + dg * dg : fa * fa + fg * fg;
                            const float dd = bh * bh + rr;
                            const float e1 = dd > 0.0f ? bh * hz_rsq(dd) : 1.0f;
                            if (gb[6] != 0.0f) el = fmaxf(el, e1);
                        }
                    }
                }
                el = fmaxf(el, __shfl_xor(el, 1)); el = fmaxf(el, __shfl_xor(el, 2));
                const bool lead = on && (lane & 3u) == 0u && el > -1.5f;
                if (__ballot(lead) != 0ull) raise(fminf(el + (float)(q.key >> 11) * (1.0f / 16384.0f), 1.0f), (q.key & 31u) | (((q.key >> 5) & 63u) << 8), lead);
            }
        }
        if (flush) n_leafq = 0;
        HZ_SYNC();
    }
    if (!give_up && open_n > 0) {
        // the budget of rounds ran out: what is left raises the horizon by the box it was put aside with (a near node has none: no culling)
        for (int base = 0; base < open_n; base += kHzWave) {
            const int i = base + (int)lane;
            const bool on = i < open_n;
            const uint32_t ref = on ? open_ref[i] : 0u, key = on ? open_key[i] : 0u;
            if (on && (ref >> 31)) give_up = true;
            const bool far = on && !(ref >> 31);
            if (__ballot(far) != 0ull) raise(key_top(key), ((key >> 16) & 31u) | (((key >> 21) & 63u) << 8), far);
        }
        give_up = __ballot(give_up) != 0ull;
    }
    const unsigned long long clk1 = p.counters ? HZ_CLOCK() : 0ull;
    if (p.counters && lane == 0) {
        atomicAdd(&p.counters[0], (unsigned long long)rounds); atomicAdd(&p.counters[1], n_tests); atomicAdd(&p.counters[2], (unsigned long long)near_n);
        atomicAdd(&p.counters[3], n_leaves); atomicAdd(&p.counters[4], give_up ? 1ull : 0ull); atomicAdd(&p.counters[5], 1ull);
    }
    if (give_up) { write_out(zero8); return; }
    HZ_SYNC();
    for (int c = 0; c < C; c++) { if (!T16 || lane < 32u) table[c * (kHzWave / (T16 ? 2 : 1)) + (int)lane] = 0u; }      // (in the walk's storage: it is done with)
    HZ_SYNC();
#ifdef AOB_HORIZON_EMU
    if (p.dbg && lane < 32u) p.dbg[blockIdx.x * 32u + lane] = __uint_as_float(H[lane]) - 2.0f;
#endif

    const uint32_t allK = K >= 32 ? 0xffffffffu : (1u << K) - 1u;
    const float fC = (float)C;
    const float inv_bias = p.bias > 0.0f ? 1.0f / p.bias : 0.0f;
    auto tab_or = [&](int c, uint32_t mask) {
        if (T16) atomicOr(&table[c * (kHzWave / 2) + (int)(lane >> 1)], mask << ((lane & 1u) << 4));
        else atomicOr(&table[c * kHzWave + (int)lane], mask);
    };
    auto tab_get = [&](int c) -> uint32_t {
        return T16 ? (table[c * (kHzWave / 2) + (int)(lane >> 1)] >> ((lane & 1u) << 4)) & 0xffffu : table[c * kHzWave + (int)lane];
    };
    // block the rows of `mask` from the column u falls in
    auto block = [&](float u, uint32_t mask, bool want) {
        const int c0 = (int)fminf(fmaxf(floorf((u - kMarginU) * fC), 0.0f), fC);
        if (want && c0 < C) tab_or(c0, mask);
    };

    // ---- phase B: the near triangles, every surfel in its own frame ----
    for (int i = 0; i < ((AOB_HZ_SKIP & 1) ? 0 : near_n); i++) {
        // (a wave-uniform address through the constant address space: the vertices come in by scalar loads)
        const uint32_t tri = (uint32_t)__builtin_amdgcn_readfirstlane((int)near_tri[i]);
        const float4* tp = p.leaf_tris + 3ull * tri;
        HZ_SCALAR_POINTER(tp);
        const hz_const_floats tw = (hz_const_floats)tp;
        float al[3], be[3], ga[3];
#pragma unroll
        for (int vi = 0; vi < 3; vi++) {
            const f3 d = make_f3(tw[4 * vi] - P.x, tw[4 * vi + 1] - P.y, tw[4 * vi + 2] - P.z);
            al[vi] = hz_dot(Xd, d); be[vi] = hz_dot(Nd, d); ga[vi] = hz_dot(Zd, d);
        }
        const float bmax = fmaxf(fmaxf(be[0], be[1]), be[2]);
        const bool want = framed && bmax > -1e-7f;
        if (__ballot(want) == 0ull) continue;
        const float bm = fmaxf(bmax, 0.0f) * 1.00001f + 1e-12f;
        const float ca = (al[0] + al[1] + al[2]) * (1.0f / 3.0f), cg = (ga[0] + ga[1] + ga[2]) * (1.0f / 3.0f);
        float R2 = 0.0f;
#pragma unroll
        for (int vi = 0; vi < 3; vi++) { const float dx = al[vi] - ca, dz = ga[vi] - cg; R2 = fmaxf(R2, dx * dx + dz * dz); }
        R2 = R2 * 1.0001f + 1e-30f;
        const float rc2 = ca * ca + cg * cg;
        const bool full = rc2 <= R2;
        // rho_min from below: the projected triangle's box, and its bounding circle
        const float mnx = fminf(fminf(al[0], al[1]), al[2]), mxx = fmaxf(fmaxf(al[0], al[1]), al[2]);
        const float mnz = fminf(fminf(ga[0], ga[1]), ga[2]), mxz = fmaxf(fmaxf(ga[0], ga[1]), ga[2]);
        const float bx = fmaxf(fmaxf(mnx, -mxx), 0.0f), bz = fmaxf(fmaxf(mnz, -mxz), 0.0f);
        const float R = hz_sqrt_up(R2), rc = hz_sqrt_down(rc2);
        const float cd = fmaxf(rc - R, 0.0f);
        const float d2 = fmaxf(bx * bx + bz * bz, cd * cd) * 0.9999f;
        const float ua = d2 * hz_rcp(d2 + bm * bm) * 0.99999f;
        const float hb = fminf(bm * inv_bias, 1.0f);              // a hit lies at least `bias` along its ray: cos(theta) <= beta_max / bias
        const float ub = p.bias > 0.0f ? 1.0f - hb * hb : 0.0f;
        const float u = fmaxf(ua, ub);
        const float th = hz_atan2_turns(cg, ca);
        const float alpha = full ? 0.5f : hz_asin_turns_up(R * hz_rcp(rc) * 1.0001f);
        block(u, hz_sector_mask(th - alpha, th + alpha, full || alpha >= 0.25f, K, kMarginPhi), want);
    }

    const unsigned long long clk2 = p.counters ? HZ_CLOCK() : 0ull;
#ifdef AOB_HORIZON_EMU
    HZ_SYNC();
    if (p.dbg_tables) for (int c = 0; c < C; c++) p.dbg_tables[((uint64_t)blockIdx.x * 64u + lane) * 32u + c] = tab_get(c);
#endif
    // ---- phase C: the horizon's boundary as arcs, every arc into the local tables ----
    {
        // lane = arc: even arcs run along sector s = lane / 2 at its (raised) latitude, odd arcs join it to the next sector's along the meridian between them
        const int s = (int)(lane >> 1), sn = (s + 1) & 31;
        auto lat = [&](int sec, float& sE, float& cE) {
            float h = __uint_as_float(H[sec]) - 2.0f;
            h = fminf(fmaxf(h, -1.0f), 1.0f);
            float cs = sqrtf(fmaxf(1.0f - h * h, 0.0f));
            if (h < 0.0f) {
                // a southern latitude's chord dips below it: raise the latitude until the chord's middle touches it (tan E' = tan E cos(half the sector))
                const float t = cs > 1e-6f ? (h / cs) * 0.9951847f : -1e6f;       // cos(pi / 32)
                const float il = hz_rsq(1.0f + t * t);
                h = t * il; cs = il;
            }
            // ... and a margin's worth higher
            const float sd = 2e-3f, cdm = 0.999998f;
            const float h2 = h * cdm + cs * sd, c2 = cs * cdm - h * sd;
            if (c2 <= 0.0f || h2 >= 1.0f) { sE = 1.0f; cE = 0.0f; } else { sE = h2; cE = c2; }
        };
        auto point = [&](int sec, float az_turns) -> f3 {
            float sE, cE;
            lat(sec, sE, cE);
            float sa, ca;
            sincos2pi(az_turns - floorf(az_turns), sa, ca);
            const f3 t = make_f3(cE * ca, sE, cE * sa);
            return make_f3(t.x * Xt.x + t.y * Nt.x + t.z * Zt.x, t.x * Xt.y + t.y * Nt.y + t.z * Zt.y, t.x * Xt.z + t.y * Nt.z + t.z * Zt.z);
        };
        const float az0 = (float)s * (1.0f / 32.0f), az1 = (float)(s + 1) * (1.0f / 32.0f);
        const f3 a = (lane & 1u) ? point(s, az1) : point(s, az0);
        const f3 b = (lane & 1u) ? point(sn, az1) : point(s, az1);
        f3 n = cross(a, b);
        const float nl2 = hz_dot(n, n);
        float ok = 0.0f;
        if (nl2 > 1e-14f) { const float il = hz_rsq(nl2); n = make_f3(n.x * il, n.y * il, n.z * il); ok = 1.0f; }
        const f3 na = cross(n, a), bn = cross(b, n);
        float* A = arcs[lane];
        A[0] = b.x; A[1] = b.y; A[2] = b.z; A[3] = n.x; A[4] = n.y; A[5] = n.z; A[6] = na.x; A[7] = na.y; A[8] = na.z; A[9] = bn.x; A[10] = bn.y; A[11] = ok > 0.0f ? bn.z : 0.0f;
        if (ok == 0.0f) { A[3] = 0.0f; A[4] = 0.0f; A[5] = 0.0f; }
        HZ_SYNC();
        // cos(theta) and azimuth of a world direction in the surfel's frame
        auto local = [&](const f3& d, float& cs, float& th) {
            const float x = hz_dot(Xd, d), y = hz_dot(Nd, d), z = hz_dot(Zd, d);
            const float l2 = x * x + y * y + z * z;
            cs = l2 > 0.0f ? y * hz_rsq(l2) : 1.0f;
            th = hz_atan2_turns(z, x);
        };
        // The surfel's own pole first: if the normal itself points into the blocked set, no walk along a meridian ever leaves it — every row is
        // blocked from its first column.  (Its sector's horizon and both neighbours'; next to the tile's pole, where sectors meet, every sector's.)
        {
            const float sN = hz_dot(Nt, Nw), xN = hz_dot(Xt, Nw), zN = hz_dot(Zt, Nw);
            float th = hz_atan2_turns(zN, xN);
            th -= floorf(th);
            const int sk = (int)(th * 32.0f) & 31;
            uint32_t hb = max(max(H[sk], H[(sk + 1) & 31]), H[(sk + 31) & 31]);
            if (sN > 0.99f) for (int q = 0; q < kHzSectors; q++) hb = max(hb, H[q]);
            const float hN = __uint_as_float(hb) - 2.0f;
            if (framed && sN <= hN + 4.0f * kMarginF + 2e-3f) tab_or(0, allK);
        }
        float cs_a, th_a;
        {
            const float* L = arcs[kHzWave - 1];
            local(make_f3(L[0], L[1], L[2]), cs_a, th_a);            // arc 0 starts where arc 63 ends
        }
        for (int i = 0; i < ((AOB_HZ_SKIP & 2) ? 0 : kHzWave); i++) {
            const float* L = arcs[i];
            const f3 bb = make_f3(L[0], L[1], L[2]), nn = make_f3(L[3], L[4], L[5]);
            float cs_b, th_b;
            local(bb, cs_b, th_b);
            float f = fmaxf(cs_a, cs_b);
            const bool arc = nn.x != 0.0f || nn.y != 0.0f || nn.z != 0.0f;
            if (arc) {
                // the arc's point nearest the surfel's pole: the pole's foot on the arc's plane, if it falls between the ends
                const float dn = hz_dot(Nw, nn);
                const f3 q = make_f3(Nw.x - dn * nn.x, Nw.y - dn * nn.y, Nw.z - dn * nn.z);
                const float ql2 = hz_dot(q, q);
                const bool inside = ql2 > 1e-12f && hz_dot(q, make_f3(L[6], L[7], L[8])) >= 0.0f && hz_dot(q, make_f3(L[9], L[10], L[11])) >= 0.0f;
                if (inside) { float cs_q, th_q; local(q, cs_q, th_q); f = fmaxf(f, cs_q); }
            }
            f = fminf(f + kMarginF, 1.0f);
            const float rel = (th_b - th_a) - floorf((th_b - th_a) + 0.5f);
            const float lo = fminf(th_a, th_a + rel), hi = fmaxf(th_a, th_a + rel);
            const bool full = fabsf(rel) > 0.49f || f >= 0.9999f;
            const float u = f > 0.0f ? 1.0f - f * f : 1.0f;
            block(u, hz_sector_mask(lo, hi, full, K, kMarginPhi), framed && f > 0.0f);
            cs_a = cs_b; th_a = th_b;
        }
    }
    HZ_SYNC();

    // ---- free columns per row: the columns before the first blocked one ----
    uint32_t fr[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (framed) {
        uint32_t cum = 0u;
        uint32_t cnt[32];
#pragma unroll
        for (int r = 0; r < 32; r++) cnt[r] = 0u;
        for (int c = 0; c < C; c++) {
            cum |= tab_get(c);
            const uint32_t freebits = ~cum & allK;
#pragma unroll
            for (int r = 0; r < 32; r++) cnt[r] += (freebits >> r) & 1u;
        }
#pragma unroll
        for (int r = 0; r < 32; r++) fr[r >> 2] |= cnt[r] << ((r & 3) * 8);
    }
    write_out(fr);
    if (p.skipped) {
        float tot = 0.0f;
        for (int r = 0; r < 32; r++) tot += (float)((fr[r >> 2] >> ((r & 3) * 8)) & 0xffu);
        tot = hz_wave_sum(tot);
        if (lane == 0) { atomicAdd(&p.skipped[0], (unsigned long long)tot); atomicAdd(&p.skipped[1], 64ull); }
    }
    if (p.counters && lane == 0) {
        // (clock ticks per phase, summed over the tiles: A, B, C + the epilogue)
        const unsigned long long clk3 = HZ_CLOCK();
        atomicAdd(&p.counters[6], clk1 - clk0); atomicAdd(&p.counters[7], clk2 - clk1); atomicAdd(&p.counters[8], clk3 - clk2);
    }
}


} // namespace aob
#if defined(__clang__) && !defined(AOB_HZ_NO_CONTRACT)
#pragma clang fp contract(off)
#endif

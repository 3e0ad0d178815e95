// resolve.hip — integer/float resolve + nearest-neighbour edge bleed + format conversion (gfx950).
//
// Replaces PostprocessPixelShader.hlsl:31-86 drawn as a full-screen quad into the AO
// texture's own format (BakeEngine.cpp:482,688-704).  One thread per texel; the bleed
// window is (2B+1)^2 reads of 8-byte cells that neighbouring lanes share through L1.
// Streaming, HBM-bound: 8 B read + channels*bpc B written per texel.
#include "aob_internal.h"
#include "aob_math.h"

namespace aob {

static constexpr int kBlock = 256;

template <int QUANT>
__device__ __forceinline__ bool cell_covered(AobResult r)      // PostprocessPixelShader.hlsl:43
{
    return QUANT ? (int32_t)r.weight_bits != 0 : __uint_as_float(r.weight_bits) != 0.0f;
}
template <int QUANT>
__device__ __forceinline__ bool cell_positive(AobResult r)     // :66
{
    return QUANT ? (int32_t)r.weight_bits > 0 : __uint_as_float(r.weight_bits) > 0.0f;
}
// :31-35,81 — ((T*255)/Wt)/255.0f with INTEGER division in the reference's mode
template <int QUANT>
__device__ __forceinline__ float cell_value(AobResult r)
{
    if (QUANT) {
        int32_t Wt = (int32_t)r.weight_bits;
        int32_t num = (int32_t)(r.total_bits * 255u);
        int32_t q = (Wt == -1 && num == INT32_MIN) ? INT32_MIN : num / Wt;
        return (float)q / 255.0f;
    }
    return __uint_as_float(r.total_bits) / __uint_as_float(r.weight_bits);
}

__device__ __forceinline__ uint8_t to_unorm8(float v)
{
    // D3D11 float -> UNORM: NaN -> 0, clamp, scale, +0.5, truncate
    if (!(v == v)) return 0;
    float c = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    return (uint8_t)(c * 255.0f + 0.5f);
}

template <int QUANT>
__global__ __launch_bounds__(kBlock)
void k_resolve(const AobResult* __restrict__ results, int32_t W, int32_t H, int32_t B, int32_t bleed_mode,
               int32_t channels, int32_t bpc, int f16mode, uint8_t* __restrict__ out)
{
    const int64_t cells = (int64_t)W * H;
    const int64_t cell = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (cell >= cells) return;
    const int32_t py = (int32_t)(cell / W), px = (int32_t)(cell % W);
    const AobResult me = results[cell];
    float result = 0.0f;
    if (!cell_covered<QUANT>(me) && bleed_mode == AOB_BLEED_NEAREST) {
        // the nearest covered texel within B (inclusive), inside the texture only; ties: rows top to bottom, then left to right
        int32_t closest = B * B + 1;
        for (int32_t y = max(py - B, 0); y <= min(py + B, H - 1); y++) {
            for (int32_t x = max(px - B, 0); x <= min(px + B, W - 1); x++) {
                const int32_t d2 = (x - px) * (x - px) + (y - py) * (y - py);
                if (d2 >= closest) continue;
                const AobResult nb = results[(int64_t)y * W + x];
                if (cell_positive<QUANT>(nb)) { closest = d2; result = cell_value<QUANT>(nb); }
            }
        }
    } else if (!cell_covered<QUANT>(me)) {
        // :43-78 — x outer, y inner; the reference reads Results[y*W + x] with the RAW x and y
        // (its wrapped realx/realy are never used), so a negative x aliases into the previous
        // row and an index outside the buffer reads zeros (D3D11 out-of-bounds = 0)
        int32_t closest = (B + 1) * (B + 1);
        for (int32_t x = px - B; x <= px + B; x++) {
            for (int32_t y = py - B; y <= py + B; y++) {
                int64_t idx = (int64_t)y * W + x;
                if (idx < 0 || idx >= cells) continue;
                AobResult nb = results[idx];
                if (cell_positive<QUANT>(nb)) {
                    int32_t d2 = (x - px) * (x - px) + (y - py) * (y - py);
                    if (d2 < closest && d2 < B * B) { closest = d2; result = cell_value<QUANT>(nb); }
                }
            }
        }
    } else {
        result = cell_value<QUANT>(me);
    }
    // :85 float4(result, 0, 0, 1) into the texture's format; write-back layout BakeEngine.cpp:705-709
    uint8_t* dst = out + (size_t)cell * (size_t)channels * (size_t)bpc;
    for (int c = 0; c < channels; c++) {
        float v = c == 0 ? result : (c == 3 ? 1.0f : 0.0f);
        if (bpc == 1) dst[c] = to_unorm8(v);
        else if (bpc == 2) ((uint16_t*)dst)[c] = (uint16_t)f32_to_f16(v, f16mode);
        else ((float*)dst)[c] = v;
    }
}

int resolve_material(AobBake* b, int32_t material)
{
    DeviceMaterial& m = b->materials[material];
    const uint64_t cells = (uint64_t)m.width * m.height;
    if (!cells) return AOB_OK;
    const int f16mode = b->opt.normal_fp16 == AOB_FP16_RTZ ? FP16_RTZ : FP16_RNE;
    dim3 grid((uint32_t)((cells + kBlock - 1) / kBlock));
    if (b->opt.accumulate == AOB_ACCUMULATE_QUANTISED)
        hipLaunchKernelGGL(k_resolve<1>, grid, dim3(kBlock), 0, b->stream, m.results, m.width, m.height,
                           b->opt.bleed_distance, b->opt.bleed_mode, m.channels, m.bytes_per_channel, f16mode, m.image);
    else
        hipLaunchKernelGGL(k_resolve<0>, grid, dim3(kBlock), 0, b->stream, m.results, m.width, m.height,
                           b->opt.bleed_distance, b->opt.bleed_mode, m.channels, m.bytes_per_channel, f16mode, m.image);
    AOB_HIP(hipGetLastError());
    return AOB_OK;
}

} // namespace aob

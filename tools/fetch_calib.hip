// fetch_calib.hip — known-byte kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in k_trace's own
// access pattern (MI355X_MICROARCH.md §HBM: FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read; other
// widths are uncalibrated — "calibrate on a known byte count in your own access pattern").
//
//   k_stream     every lane reads 16 B, consecutive lanes consecutive addresses, 2 GiB once          (the guide's case)
//   k_gather64   every lane reads ONE random 64-byte record as 4 x 16 B (what a lane of k_trace does with a 4-ary node),
//                each record of an 8 GiB table at most once, 1 GiB in all — far beyond the 256 MiB Infinity Cache,
//                so every byte comes from HBM
//   k_gather48   the same with 48-byte records (a leaf triangle)
// Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`; tools/pmc_record.py divides the known bytes by FETCH_SIZE x 1 KiB.
// Build: hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o tools/_fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void k_stream(const uint4* __restrict__ src, uint64_t n16, uint32_t* sink)
{
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <int WORDS16>
__global__ void k_gather(const uint4* __restrict__ table, uint64_t records, uint64_t reads, uint64_t stride_records, uint32_t* sink)
{
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < reads; i += (uint64_t)gridDim.x * blockDim.x) {
        // a bijection of [0, reads) into the table: odd multiplier modulo a power of two, then spread by the stride
        const uint64_t r = ((i * 0x9E3779B97F4A7C15ull) & (reads - 1)) * stride_records % records;
        const uint4* p = table + r * WORDS16;
#pragma unroll
        for (int k = 0; k < WORDS16; k++) { const uint4 v = p[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main()
{
    const uint64_t table_bytes = 8ull << 30;
    uint4* table = nullptr; uint32_t* sink = nullptr;
    if (hipMalloc(&table, table_bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(table, 1, table_bytes);
    hipDeviceSynchronize();
    const dim3 grid(256 * 8), block(256);
    const uint64_t stream_bytes = 2ull << 30;
    hipLaunchKernelGGL(k_stream, grid, block, 0, 0, table + (4ull << 30) / 16, stream_bytes / 16, sink);
    hipDeviceSynchronize();
    const uint64_t reads = 1ull << 24;                    // 16 Mi records
    hipLaunchKernelGGL(k_gather<4>, grid, block, 0, 0, table, table_bytes / 64, reads, (table_bytes / 64) / reads, sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_gather<3>, grid, block, 0, 0, table, table_bytes / 48, reads, (table_bytes / 48) / reads, sink);
    hipDeviceSynchronize();
    printf("{\"k_stream_bytes\": %llu, \"k_gather64_bytes\": %llu, \"k_gather48_bytes\": %llu}\n",
           (unsigned long long)stream_bytes, (unsigned long long)(reads * 64), (unsigned long long)(reads * 48));
    return 0;
}

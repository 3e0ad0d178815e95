// l1_probe.hip — what a wave-instruction of k_trace's node fetch costs on the vector memory path of gfx950, by access shape.
//
// k_trace fetches a 64-byte node per lane as four 16-byte loads; extra loads in its inner loop cost far more than extra
// arithmetic (profiles/r02/vmem_path/README.md).  This probe times the load shapes in isolation: 32 wavefronts per CU (as
// k_trace), each in a dependent chain — the loaded words pick the next node, like a walk — over a table of 64-byte
// records, either 16 KiB (stays in the CU's L1) or 17 MB (k_trace's node array on BASELINE configs[2]: L2 hits mostly).
//
//   mode 0  lane-per-node:  every lane its own random node, 4 loads (the four 16-byte pieces)            [k_trace's shape]
//   mode 1  the same with 36 of the 64 lanes active (k_trace's lane occupancy)
//   mode 2  quad-per-node:  the four lanes of a quad load the four pieces of ONE node in one instruction; 4 instructions
//           fetch 64 nodes — the same bytes as mode 0 in a quarter of the lines per instruction
//   mode 3  lane-per-node, ONE 16-byte load per iteration (the per-instruction cost at 64 lines per instruction)
//   mode 4  every lane the same node, 4 loads (one line per instruction)
//   mode 5  16 random nodes per instruction, each read by 4 lanes (the same piece): coherent rays
//   mode 6  mode 0 with 18 lanes active
// Output: one JSON line per (table, mode): ns per wave-instruction per CU and the CU cycles that is at the measured clock.
// Build: hipcc -O3 --offload-arch=gfx950 tools/l1_probe.hip -o tools/_l1_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(64, 8)
void k_probe(const uint4* __restrict__ table, uint32_t mask, uint32_t iters, uint32_t* sink)
{
    const uint32_t lane = threadIdx.x;
    uint32_t state = mix(blockIdx.x * 64u + lane + 1u);
    uint32_t acc = 0;
    const uint32_t active = MODE == 1 ? 36u : (MODE == 6 ? 18u : 64u);
    if (lane < active) {
        for (uint32_t it = 0; it < iters; it++) {
            uint32_t node;
            if (MODE == 0 || MODE == 1 || MODE == 3 || MODE == 6) node = state & mask;
            else if (MODE == 4) node = __builtin_amdgcn_readfirstlane(state) & mask;
            else node = 0;
            if (MODE == 0 || MODE == 1 || MODE == 4 || MODE == 6) {
                const uint4* p = table + 4ull * node;
                const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
                state = mix(state ^ a.x ^ b.y ^ c.z ^ d.w);
            } else if (MODE == 3) {
                const uint4 a = table[4ull * node];
                state = mix(state ^ a.x);
            } else if (MODE == 2) {
                // quad q fetches node_i of "ray" 4q + i in instruction i: lane 4q + c loads piece c
                const uint32_t q = lane >> 2, c = lane & 3u;
                uint32_t x = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t n = mix(__shfl(state, 4 * q + i)) & mask;        // the owner's node, known to the quad
                    const uint4 a = table[4ull * n + c];
                    x ^= a.x ^ a.y;
                }
                state = mix(state ^ x);
            } else if (MODE == 5) {
                const uint32_t q = lane >> 2;
                uint32_t x = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t n = mix(__shfl(state, 4 * q) + i) & mask;         // the quad's four lanes agree on the node
                    const uint4 a = table[4ull * n + i];
                    x ^= a.x ^ a.y;
                }
                state = mix(state ^ x);
            }
        }
        acc = state;
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <int MODE>
static void run(const char* table_name, const uint4* table, uint32_t records, uint32_t iters, uint32_t* sink, double clock_ghz)
{
    const int cus = 256, waves = cus * 32;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(waves), dim3(64), 0, 0, table, records - 1u, iters / 8u, sink);      // warm
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(waves), dim3(64), 0, 0, table, records - 1u, iters, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const int per_iter = MODE == 3 ? 1 : 4;
    const double instr_per_cu = 32.0 * iters * per_iter;
    const double ns = ms * 1e6 / instr_per_cu;
    printf("{\"table\": \"%s\", \"mode\": %d, \"ms\": %.3f, \"loads_per_iteration\": %d, \"ns_per_wave_instruction_per_cu\": %.3f, "
           "\"cu_cycles_per_wave_instruction_at_%.1f_GHz\": %.2f}\n", table_name, MODE, ms, per_iter, ns, clock_ghz, ns * clock_ghz);
    fflush(stdout);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main()
{
    const uint32_t big_records = 1u << 18;                 // 16 MiB of 64-byte records
    const uint32_t small_records = 1u << 8;                // 16 KiB
    uint4* table = nullptr; uint32_t* sink = nullptr;
    if (hipMalloc(&table, (size_t)big_records * 64) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    std::vector<uint32_t> host((size_t)big_records * 16);
    uint32_t s = 12345u;
    for (auto& w : host) { s = s * 1664525u + 1013904223u; w = s; }
    hipMemcpy(table, host.data(), host.size() * 4, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    const double ghz = 2.4;
    const uint32_t iters = 20000;
    for (int t = 0; t < 2; t++) {
        const char* name = t == 0 ? "16 KiB (L1)" : "16 MiB (L2)";
        const uint32_t rec = t == 0 ? small_records : big_records;
        run<0>(name, table, rec, iters, sink, ghz);
        run<1>(name, table, rec, iters, sink, ghz);
        run<6>(name, table, rec, iters, sink, ghz);
        run<2>(name, table, rec, iters, sink, ghz);
        run<3>(name, table, rec, iters, sink, ghz);
        run<4>(name, table, rec, iters, sink, ghz);
        run<5>(name, table, rec, iters, sink, ghz);
    }
    return 0;
}

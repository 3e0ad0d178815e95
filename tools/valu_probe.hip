// valu_probe.hip — how many cycles does one wave64 VALU instruction cost a SIMD of gfx950?
// Settles the issue-rate model behind DESIGN.md §3.1 (k_trace is VALU-issue bound): independent v_fma_f32 /
// v_pk_fma_f32 / v_perm_b32 / v_max3_f32 streams at 1, 2, 4 and 8 wavefronts per SIMD on every CU.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o gpurun_out/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(64) void k_probe(float* out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = { a0, a1 }, p1 = { a2, a3 }, p2 = { a4, a5 }, p3 = { a6, a7 }, pb = { b, b }, pc = { c, c };
    unsigned u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7, sel = 0x07010c0c;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        } else if (KIND == 1) {
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
        } else if (KIND == 2) {
            REP16(asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5"
                               : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0), "v"(sel));)
        } else {
            REP16(asm volatile("v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        }
    }
    float r = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(u0 ^ u1 ^ u2 ^ u3);
    if (r == 12345.678f) out[0] = r;
}

template <int KIND>
static void run(const char* name, float* out, int cus)
{
    const int iters = 4000;     // x 64 instructions
    for (int waves_per_simd : { 1, 2, 4, 8 }) {
        dim3 grid(cus * 4 * waves_per_simd);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_probe<KIND>, grid, dim3(64), 0, 0, out, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_probe<KIND>, grid, dim3(64), 0, 0, out, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * 64 * waves_per_simd;
        printf("{\"probe\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"ns_per_wave_instr_per_simd\": %.4f, \"cycles_at_2.4GHz\": %.3f}\n",
               name, waves_per_simd, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
}

int main()
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* out; hipMalloc(&out, 64);
    printf("{\"cus\": %d}\n", cus);
    run<0>("v_fma_f32", out, cus);
    run<1>("v_pk_fma_f32", out, cus);
    run<2>("v_perm_b32", out, cus);
    run<3>("v_max3_f32", out, cus);
    return 0;
}

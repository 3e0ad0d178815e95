// ref_launch.cpp — loads a code object built from the REFERENCE's OpenCL kernels (oracle/ref_kernels.mk) and launches
// its kernels through the HIP module API.
//
// TEST INFRASTRUCTURE ONLY (like everything under oracle/): called by tests/refkernels.py to pin the oracle and the
// library's kernels against the reference's own device code.  Nothing under aobaker_amd/ links or loads this.
//
// The kernels are OpenCL C compiled for amdgcn-amd-amdhsa; their hidden arguments (block counts, group sizes, global
// offsets — what get_global_id() reads) are filled in by the HIP runtime from the code object's metadata, so a plain
// hipModuleLaunchKernel with the explicit arguments is a complete launch.  Buffers are plain hipMalloc allocations.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

static thread_local char g_err[512] = "";

static int fail(const char* what, hipError_t e)
{
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return -1;
}

#define REF_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail(#call, e__); } while (0)

extern "C" {

const char* ref_last_error(void) { return g_err; }

int ref_load(const char* path, void** module_out)
{
    hipModule_t m = nullptr;
    REF_HIP(hipModuleLoad(&m, path));
    *module_out = (void*)m;
    return 0;
}

int ref_unload(void* module)
{
    REF_HIP(hipModuleUnload((hipModule_t)module));
    return 0;
}

int ref_alloc(uint64_t bytes, void** out)
{
    REF_HIP(hipMalloc(out, bytes ? bytes : 1));
    return 0;
}

int ref_free(void* p)
{
    REF_HIP(hipFree(p));
    return 0;
}

int ref_h2d(void* dst, const void* src, uint64_t bytes)
{
    if (bytes) REF_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}

int ref_d2h(void* dst, const void* src, uint64_t bytes)
{
    if (bytes) REF_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int ref_memset(void* dst, int value, uint64_t bytes)
{
    if (bytes) REF_HIP(hipMemset(dst, value, bytes));
    return 0;
}

// One launch of `kernel`: `blocks` workgroups of `block_size` work-items (the reference's kernels are one-dimensional),
// args[i] points at the i-th explicit argument (a device pointer or a by-value scalar).  Synchronous; an execution
// error of the kernel comes back as the return code.
int ref_launch(void* module, const char* kernel, uint32_t blocks, uint32_t block_size, void** args, uint32_t arg_count,
               float* ms_out)
{
    (void)arg_count;
    hipFunction_t f = nullptr;
    REF_HIP(hipModuleGetFunction(&f, (hipModule_t)module, kernel));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    REF_HIP(hipEventCreate(&e0));
    REF_HIP(hipEventCreate(&e1));
    REF_HIP(hipEventRecord(e0, nullptr));
    hipError_t e = hipModuleLaunchKernel(f, blocks, 1, 1, block_size, 1, 1, 0, nullptr, args, nullptr);
    if (e != hipSuccess) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return fail("hipModuleLaunchKernel", e); }
    REF_HIP(hipEventRecord(e1, nullptr));
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return fail("hipDeviceSynchronize", e); }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms_out) *ms_out = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
}

} // extern "C"

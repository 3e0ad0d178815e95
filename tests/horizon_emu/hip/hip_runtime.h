// A stand-in for <hip/hip_runtime.h> that lets ONE kernel file (aobaker_amd/csrc/horizon_kernel.h) run on the CPU: a wavefront is 64 host
// threads in lockstep at every cross-lane operation (a barrier and an exchange buffer), LDS is static storage.  Test infrastructure only
// (tests/horizon_emu/horizon_emu.cpp, tests/test_horizon_emu.py): it checks the kernel's LOGIC against the oracle where there is no GPU; nothing of the
// product includes it.
#pragma once
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__

struct float4 { float x, y, z, w; };
struct uint4 { uint32_t x, y, z, w; };
struct uint3 { uint32_t x, y, z; };
struct uint2 { uint32_t x, y; };
struct dim3 { unsigned x = 1, y = 1, z = 1; };
static inline float4 make_float4(float x, float y, float z, float w) { return { x, y, z, w }; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return { x, y, z, w }; }
static inline uint3 make_uint3(uint32_t x, uint32_t y, uint32_t z) { return { x, y, z }; }

namespace emu {
constexpr int kLanes = 64;
struct Idx { unsigned x = 0; };
extern thread_local Idx threadIdx_, blockIdx_;
extern std::barrier<>* bar;
extern uint64_t slots[kLanes];
inline void sync() { bar->arrive_and_wait(); }
template <class T> inline uint64_t bits(T v) { uint64_t b = 0; std::memcpy(&b, &v, sizeof v); return b; }
template <class T> inline T unbits(uint64_t b) { T v; std::memcpy(&v, &b, sizeof v); return v; }
template <class T> inline T from_lane(T v, unsigned src)
{
    slots[threadIdx_.x] = bits(v);
    sync();
    const T r = unbits<T>(slots[src & (kLanes - 1)]);
    sync();
    return r;
}
}
#define threadIdx emu::threadIdx_
#define blockIdx emu::blockIdx_

template <class T> static inline T __shfl_xor(T v, int off) { return emu::from_lane(v, emu::threadIdx_.x ^ (unsigned)off); }
static inline uint64_t __ballot(bool p)
{
    emu::slots[emu::threadIdx_.x] = p ? 1u : 0u;
    emu::sync();
    uint64_t m = 0;
    for (int i = 0; i < emu::kLanes; i++) m |= (uint64_t)(emu::slots[i] & 1u) << i;
    emu::sync();
    return m;
}
static inline int __builtin_amdgcn_readfirstlane(int v) { return emu::from_lane(v, 0u); }
static inline int __builtin_amdgcn_readlane(int v, int l) { return emu::from_lane(v, (unsigned)l); }
static inline float __builtin_amdgcn_rcpf(float v) { return 1.0f / v; }
static inline float __builtin_amdgcn_rsqf(float v) { return 1.0f / std::sqrt(v); }
static inline float __builtin_amdgcn_sqrtf(float v) { return std::sqrt(v); }
#define __builtin_amdgcn_fence(...) ((void)0)
static inline void __builtin_amdgcn_wave_barrier() { std::atomic_thread_fence(std::memory_order_seq_cst); emu::sync(); }
static inline int __popcll(uint64_t v) { return __builtin_popcountll(v); }
static inline int __popc(uint32_t v) { return __builtin_popcount(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint32_t atomicMax(uint32_t* p, uint32_t v) { uint32_t o = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) { } return o; }
static inline uint32_t atomicOr(uint32_t* p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
using std::isfinite;
using std::max;
using std::min;

// tests/horizon_emu/horizon_emu.cpp — aobaker_amd/csrc/horizon_kernel.h run on the CPU, one host thread per lane (tests/horizon_emu/hip/hip_runtime.h), behind a C
// entry point for ctypes.  Test infrastructure: the kernel's logic against the oracle's rays without a GPU (tests/test_horizon_emu.py).
//   g++ -O2 -std=c++20 -fPIC -shared -pthread -DAOB_HORIZON_EMU -Itests/horizon_emu -Iaobaker_amd/csrc tests/horizon_emu/horizon_emu.cpp -o tests/horizon_emu/libhorizon_emu.so
#include <hip/hip_runtime.h>

#include <thread>
#include <vector>

namespace emu {
thread_local Idx threadIdx_, blockIdx_;
std::barrier<>* bar = nullptr;
uint64_t slots[kLanes];
}

#include "horizon_kernel.h"

extern "C" int emu_horizon(const float* frames, uint64_t begin, uint64_t end, const uint32_t* nodes4q, const float* leaf_tris, const float* quant_frame,
                           uint32_t node_count, int32_t K, int32_t C, float bias, float reach, float near_factor, float mid_factor, float small2, float big2,
                           uint8_t* out, uint32_t stride, unsigned long long* counters, uint32_t tile_begin, uint32_t tile_end, float* dbg, uint32_t* dbg_tables)
{
    aob::HorizonParams p;
    std::memset(&p, 0, sizeof p);
    p.frames = (const float4*)frames; p.begin = begin; p.end = end; p.nodes4q = (const uint4*)nodes4q; p.leaf_tris = (const float4*)leaf_tris;
    p.quant_frame = quant_frame; p.node_count = node_count; p.K = K; p.C = C; p.bias = bias; p.reach = reach;
    p.near_factor = near_factor; p.mid_factor = mid_factor; p.small2 = small2; p.big2 = big2; p.out = out; p.stride = stride; p.tile_stride = 1u; p.counters = counters; p.dbg = dbg; p.dbg_tables = dbg_tables;
    std::barrier<> b(emu::kLanes);
    emu::bar = &b;
    std::vector<std::thread> th;
    for (int l = 0; l < emu::kLanes; l++)
        th.emplace_back([&, l]() {
            emu::threadIdx_.x = (unsigned)l;
            for (uint32_t t = tile_begin; t < tile_end; t++) {
                emu::blockIdx_.x = t;
                if (C <= 16 && K <= 16) aob::k_horizon<16, true>(p); else aob::k_horizon<32, false>(p);
                emu::sync();
            }
        });
    for (auto& t : th) t.join();
    emu::bar = nullptr;
    return 0;
}

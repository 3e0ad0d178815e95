// aobaker_multi.h — the multi-GPU host of the bake: ONE process, one host thread and one RCCL rank per GPU of the node
// (the reference itself is single-device, `IntersectionApi::Create(0)` BakeEngine.cpp:368; sharding
// the texels over the 8 GPUs of a node is what BASELINE.json's north star adds).
//
// Plain C++ above the C ABI of aobaker.h: every device gets its own AobBake handle (scene + BVH replicated: a 1 M-triangle
// tree is built in ~1 ms, cheaper than broadcasting it), rasterises and traces its own strip of texel rows
// (aob_bake_rasterize_rows), and the strips' result cells travel to device 0 with one grouped ncclSend / ncclRecv — each
// peer over its own xGMI link, nobody else needs the image — where the atlas is resolved, bled and downloaded.
// BakeEngine::Bake uses it when BakeEngine::SetDeviceCount(n > 1) was called (include/BakeEngine.h).
#pragma once

#include "aobaker.h"

#include <cstdint>
#include <vector>

struct AobMultiStepStats
{
	double   wall_ms = 0.0;                 // the whole step, host clock: post to all devices -> device 0 resolved
	uint64_t surfels = 0, rays = 0;         // of all devices
	std::vector<float>    trace_ms;         // per device: its trace launches (HIP events on the device's stream)
	std::vector<double>   rank_ms;          // per device: build + rasterise + trace, host clock of its thread (gather excluded)
	std::vector<uint64_t> rank_surfels;
	std::vector<float>    rank_bvh_build_ms; // per device: the tree's build (every device builds the whole tree — redundantly; HIP events)
	std::vector<double>   rank_gather_ms;   // per device: host time from posting its side of the exchange until its stream had drained (device 0: the last strip has arrived)
	std::vector<int32_t>  row_begin, row_end;   // material 0's strip of every device in this step
};

class AobMultiGpuBake
{
public:
	// devices: HIP device ordinals, one rank each (device[0] gathers, resolves and downloads).  Uploads the scene to every
	// device.  balance: after each Step the strips of texel rows are re-cut in proportion to the devices' measured times
	// (border strips of a terrain are cheaper than the middle ones), converging within two or three steps; the result cells
	// do not depend on where the cuts are (COSINE sampler) — see Step().
	static AobMultiGpuBake* Create(const AobSceneDesc* scene, const AobOptions* options, const std::vector<int>& devices, bool balance = true);
	~AobMultiGpuBake();

	// One device-side bake of every material: BVH build on every device, then per material rasterise + trace of the device's
	// strip, gather to device 0, resolve + bleed there.  Strips: tile-row (8 texel) aligned windows of texel rows for the
	// COSINE sampler; for REFERENCE_COMPAT (whose hemisphere-table index is a surfel's number in its section's list,
	// BakeEngine.cpp:598-628) every device rasterises the whole atlas and traces an equal range of the surfel list
	// (aob_bake_shard_bounds), so that N devices write exactly the cells one device would.  Where sections of one material
	// overlap in the atlas, ranges of different devices may share cells; those are summed on device 0 (ncclReduce, or
	// aob_bake_add_results between ranks that share a device): exact in QUANTISED mode — the atlas is the single-device
	// bake's byte for byte — and, in FLOAT mode, the sum of each device's once-rounded float, i.e. within 1 ulp per extra
	// device of the single-device cell (well inside the 1e-4 AO tolerance), for the shared cells only.
	int Step(AobMultiStepStats* stats = nullptr);
	// Device 0's resolved images -> the AobMaterialDesc::data pointers given at creation (BakeEngine.cpp:705-709).
	int Download();
	// Result cells of `material` on device 0 after a Step (parity tests).
	int CopyResults(int32_t material, AobResult* dst, uint64_t capacity);
	int GetStats(AobStats* stats);          // device 0's AobStats, surfel / ray counts summed over the devices

	int DeviceCount() const;
	static const char* LastError();

private:
	AobMultiGpuBake();
	struct Impl;
	Impl* impl;
};

// A pure function of its arguments (also what aobaker_amd/dist.py rebalance_rows computes): new tile-row aligned cuts of
// `height` texel rows into bounds.size()-1 strips such that, with the per-row cost of each old strip taken as
// old_ms[r] / rows(r), every new strip costs the same.  bounds[0] = 0 and bounds.back() = height stay.
std::vector<int32_t> aob_rebalance_rows(const std::vector<int32_t>& bounds, const std::vector<double>& old_ms, int32_t height, int32_t align = 8);
// Equal split of the tile rows (what the first step uses).
std::vector<int32_t> aob_equal_rows(int32_t height, int parts, int32_t align = 8);

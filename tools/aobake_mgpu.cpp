// aobake_mgpu — the multi-GPU bake from the command line: ONE process, one host thread and one RCCL rank per GPU
// (aobaker_amd/host/MultiGpu.cpp), timed the way bench.py times its steps, one JSON line on stdout.
//
//   aobake_mgpu --scene file.aobs --gpus N [--devices 0,1,...] [--steps K] [--warmup W] [--spp S] [--sampler 0|1]
//               [--accumulate 0|1] [--seed X] [--any-hit 0|1] [--normal-fp16 M] [--falloff F] [--builder 0|1|2] [--no-balance] [--check-single]
//               [--out prefix]
//
// --devices may name a device more than once: ranks that share a GPU rehearse the N-rank orchestration on one card (their
// strips move with hipMemcpyAsync, RCCL needs one rank per device).  --check-single also bakes the scene with the
// single-device aob_bake and reports whether the two images are byte-identical.  A step = BVH build on every device,
// rasterise + trace of every device's strip, gather to device 0, resolve there; the scene is resident in HBM throughout.
#include "../include/BakeEngine.h"
#include "../include/aobaker.h"
#include "../include/aobaker_multi.h"
#include "scene_file.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static uint32_t crc32_of(const unsigned char* p, size_t n)
{
	uint32_t c = 0xffffffffu;
	for (size_t i = 0; i < n; i++)
	{
		c ^= p[i];
		for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
	}
	return ~c;
}

int main(int argc, char** argv)
{
	const char* scene_path = nullptr; std::string out;
	int gpus = 1, steps = 3, warmup = 1; bool balance = true, check_single = false;
	std::vector<int> devices;
	AobOptions opt;
	aob_default_options(&opt);
	for (int i = 1; i < argc; i++)
	{
		std::string a = argv[i];
		auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
		if (a == "--scene") scene_path = next();
		else if (a == "--gpus") gpus = std::atoi(next());
		else if (a == "--devices") { std::string v = next(); size_t at = 0; while (at < v.size()) { devices.push_back(std::atoi(v.c_str() + at)); at = v.find(',', at); if (at == std::string::npos) break; at++; } }
		else if (a == "--steps") steps = std::atoi(next());
		else if (a == "--warmup") warmup = std::atoi(next());
		else if (a == "--spp") opt.sample_count = std::atoi(next());
		else if (a == "--sampler") opt.sampler = std::atoi(next());
		else if (a == "--accumulate") opt.accumulate = std::atoi(next());
		else if (a == "--seed") opt.seed = (uint32_t)std::atoi(next());
		else if (a == "--any-hit") opt.use_any_hit = std::atoi(next());
		else if (a == "--normal-fp16") opt.normal_fp16 = std::atoi(next());
		else if (a == "--falloff") opt.falloff_distance = (float)std::atof(next());
		else if (a == "--builder") opt.bvh_builder = std::atoi(next());
		else if (a == "--no-balance") balance = false;
		else if (a == "--check-single") check_single = true;
		else if (a == "--out") out = next();
		else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
	}
	if (!scene_path || steps < 1) { std::fprintf(stderr, "usage: aobake_mgpu --scene file.aobs --gpus N [--steps K] [--warmup W] [--spp S] ...\n"); return 2; }
	if (devices.empty()) for (int i = 0; i < std::max(1, gpus); i++) devices.push_back(i);
	Scene* scene = load_scene(scene_path);
	if (!scene) return 1;
	std::vector<AobMeshDesc> meshes; std::vector<AobMaterialDesc> materials;
	BakeEngine::Flatten(scene, meshes, materials);
	AobSceneDesc desc = { meshes.data(), (uint32_t)meshes.size(), materials.data(), (uint32_t)materials.size() };

	AobMultiGpuBake* multi = AobMultiGpuBake::Create(&desc, &opt, devices, balance);
	if (!multi) { std::fprintf(stderr, "aobake_mgpu: %s\n", AobMultiGpuBake::LastError()); delete scene; return 1; }
	AobMultiStepStats st;
	int rc = AOB_OK;
	for (int i = 0; i < warmup && rc == AOB_OK; i++) rc = multi->Step(&st);
	double total_ms = 0.0, best_ms = 1e30;
	std::vector<double> rank_sum(devices.size(), 0.0);
	for (int i = 0; i < steps && rc == AOB_OK; i++)
	{
		rc = multi->Step(&st);
		total_ms += st.wall_ms; best_ms = std::min(best_ms, st.wall_ms);
		for (size_t r = 0; r < devices.size() && r < st.rank_ms.size(); r++) rank_sum[r] += st.rank_ms[r];
	}
	if (rc == AOB_OK) rc = multi->Download();
	AobStats stats; std::memset(&stats, 0, sizeof stats);
	if (rc == AOB_OK) rc = multi->GetStats(&stats);
	if (rc != AOB_OK) { std::fprintf(stderr, "aobake_mgpu: %s\n", AobMultiGpuBake::LastError()); delete multi; delete scene; return 1; }
	delete multi;

	std::vector<uint32_t> crcs;
	for (int i = 0; i < scene->GetMaterialCount(); i++)
	{
		BakeTexture* tex = scene->GetMaterial(i)->GetAOTexture();
		crcs.push_back(crc32_of(tex->GetData(), (size_t)tex->GetDataLength()));
		if (!out.empty())
		{
			FILE* f = std::fopen((out + "_" + std::to_string(i) + ".raw").c_str(), "wb");
			if (f) { std::fwrite(tex->GetData(), 1, (size_t)tex->GetDataLength(), f); std::fclose(f); }
		}
	}
	int matches_single = -1;
	if (check_single)
	{
		// the same scene through the single-device entry point, into a second set of textures
		std::vector<std::vector<unsigned char>> single;
		std::vector<AobMaterialDesc> mats(materials);
		for (auto& m : mats) { single.emplace_back((size_t)m.width * m.height * m.channels * m.bytes_per_channel, (unsigned char)0xff); m.data = single.back().data(); }
		AobSceneDesc d2 = desc; d2.materials = mats.data();
		AobDeviceConfig cfg; std::memset(&cfg, 0, sizeof cfg); cfg.device = devices[0];
		rc = aob_init(&cfg);
		if (rc == AOB_OK) rc = aob_bake(&d2, &opt, nullptr);
		if (rc != AOB_OK) { std::fprintf(stderr, "aobake_mgpu: single-device bake: %s\n", aob_last_error()); delete scene; return 1; }
		matches_single = 1;
		for (size_t i = 0; i < mats.size(); i++)
			if (std::memcmp(single[i].data(), materials[i].data, single[i].size()) != 0) matches_single = 0;
	}
	const double ms = total_ms / steps;
	std::printf("{\"tool\": \"aobake_mgpu\", \"host\": \"C++: one process, one thread + one RCCL rank per device\", \"n_gpus\": %zu, \"devices\": [", devices.size());
	for (size_t r = 0; r < devices.size(); r++) std::printf("%s%d", r ? ", " : "", devices[r]);
	std::printf("], \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.3f, \"best_ms\": %.3f, \"surfels\": %llu, \"rays\": %llu, \"value\": %.1f, \"unit\": \"rays/s\", "
	            "\"triangles\": %llu, \"balance\": %s, \"rank_ms\": [", steps, warmup, ms, best_ms, (unsigned long long)st.surfels, (unsigned long long)st.rays,
	            (double)st.rays / (ms * 1e-3), (unsigned long long)stats.triangle_count, balance ? "true" : "false");
	for (size_t r = 0; r < rank_sum.size(); r++) std::printf("%s%.3f", r ? ", " : "", rank_sum[r] / steps);
	std::printf("], \"rank_bvh_build_ms\": [");
	for (size_t r = 0; r < st.rank_bvh_build_ms.size(); r++) std::printf("%s%.3f", r ? ", " : "", st.rank_bvh_build_ms[r]);
	std::printf("], \"rank_trace_kernel_ms\": [");
	for (size_t r = 0; r < st.trace_ms.size(); r++) std::printf("%s%.3f", r ? ", " : "", st.trace_ms[r]);
	std::printf("], \"gather_ms\": %.3f, \"rank_gather_ms\": [", st.rank_gather_ms.empty() ? 0.0 : st.rank_gather_ms[0]);
	for (size_t r = 0; r < st.rank_gather_ms.size(); r++) std::printf("%s%.3f", r ? ", " : "", st.rank_gather_ms[r]);
	std::printf("], \"rows\": [");
	for (size_t r = 0; r < st.row_begin.size(); r++) std::printf("%s[%d, %d]", r ? ", " : "", st.row_begin[r], st.row_end[r]);
	std::printf("], \"image_crc32\": [");
	for (size_t i = 0; i < crcs.size(); i++) std::printf("%s%u", i ? ", " : "", crcs[i]);
	std::printf("], \"matches_single_device\": %s}\n", matches_single < 0 ? "null" : (matches_single ? "true" : "false"));
	delete scene;
	return matches_single == 0 ? 3 : 0;
}

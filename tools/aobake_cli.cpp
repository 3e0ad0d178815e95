// aobake_cli — minimal headless caller of the drop-in: builds a scene through the reference's
// own accessors (include/aobaker_scene.h), calls BakeEngine::Init / Bake / Dispose exactly as
// Main.cpp:16,21 and AOBaker.cpp:243 do, and writes each AO texture as raw bytes (and a PGM for
// 1-channel 8-bit textures).
//
//   aobake_cli --scene <file.aobs> --out <prefix> [--spp N] [--sampler 0|1] [--accumulate 0|1]
//   aobake_cli --obj <file.obj> [--atlas W [--atlas-height H]] [--channels C] [--bpc B] --out <prefix> [--png] [--exr] [--gpus N] ...
//
// --obj imports a Wavefront OBJ through BakeObject::LoadFromFile (the reference's import conventions, Assets.cpp);
// --png also saves every AO texture through BakeTexture::SavePNG, as the reference's "Save AO..." button does;
// --exr through BakeTexture::SaveEXR (full precision: bake with --bpc 4)
// (Scene.cpp:215-240).
//
// Scene file (little endian), written by aobaker_amd/scenefile.py:
//   "AOBS" u32 version=1, u32 material_count, u32 object_count
//   per material: i32 width, height, channels, bytes_per_channel
//   per object:   f32 position[3], rotation[3], scale[3], u32 section_count
//     per section: i32 material, u32 vertex_count, u32 index_count, RenderVertex[vertex_count], u32[index_count]
#include "../include/BakeEngine.h"
#include "../include/aobaker.h"
#include "../include/aobaker_scene.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "scene_file.h"

int main(int argc, char** argv)
{
	const char* scene_path = nullptr; const char* obj_path = nullptr; std::string out = "ao";
	int atlas_w = 512, atlas_h = 0, channels = 1, bpc = 1; bool png = false, exr = false;
	AobOptions& opt = BakeEngine::Options();
	for (int i = 1; i < argc; i++)
	{
		std::string a = argv[i];
		auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
		if (a == "--scene") scene_path = next();
		else if (a == "--obj") obj_path = next();
		else if (a == "--atlas") atlas_w = std::atoi(next());
		else if (a == "--atlas-height") atlas_h = std::atoi(next());
		else if (a == "--channels") channels = std::atoi(next());
		else if (a == "--bpc") bpc = std::atoi(next());
		else if (a == "--gpus") BakeEngine::SetDeviceCount(std::atoi(next()));   // 0 = every visible device (include/aobaker_multi.h)
		else if (a == "--png") png = true;
		else if (a == "--exr") exr = true;
		else if (a == "--out") out = next();
		else if (a == "--spp") opt.sample_count = std::atoi(next());
		else if (a == "--sampler") opt.sampler = std::atoi(next());
		else if (a == "--accumulate") opt.accumulate = std::atoi(next());
		else if (a == "--seed") opt.seed = (uint32_t)std::atoi(next());
		else if (a == "--any-hit") opt.use_any_hit = std::atoi(next());
		else if (a == "--normal-fp16") opt.normal_fp16 = std::atoi(next());
		else if (a == "--bleed") opt.bleed_distance = std::atoi(next());
		else if (a == "--bleed-mode") opt.bleed_mode = std::atoi(next());        // 0 reference, 1 nearest (AOB_BLEED_*)
		else if (a == "--raster-mode") opt.raster_mode = std::atoi(next());      // 0 reference, 1 conservative (AOB_RASTER_*)
		else if (a == "--raster-layers") opt.raster_layers = std::atoi(next());  // 0 last fragment, 1 all (AOB_LAYERS_*)
		else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
	}
	if (!scene_path && !obj_path) { std::fprintf(stderr, "usage: aobake_cli (--scene file.aobs | --obj file.obj [--atlas W]) --out prefix [--png] [--exr] [--spp N] ...\n"); return 2; }
	Scene* scene;
	if (obj_path)
	{
		scene = new Scene();
		BakeObject* object = new BakeObject(L"Object");
		scene->AddObject(object);
		if (!object->LoadFromFile(obj_path, scene, atlas_w, atlas_h > 0 ? atlas_h : atlas_w, channels, bpc))
		{
			std::fprintf(stderr, "%s: cannot import\n", obj_path);
			delete scene;
			return 1;
		}
	}
	else scene = load_scene(scene_path);
	if (!scene) return 1;

	BakeEngine::Init();                       // Main.cpp:16
	if (BakeEngine::LastStatus() != 0) return 1;
	BakeEngine::Bake(scene);                  // AOBaker.cpp:243
	const int status = BakeEngine::LastStatus();
	if (status == 0)
	{
		const AobStats& st = BakeEngine::LastStats();
		std::printf("triangles %llu nodes %llu surfels %llu rays %llu | ms: upload %.3f bvh %.3f raster %.3f trace %.3f resolve %.3f download %.3f\n",
		            (unsigned long long)st.triangle_count, (unsigned long long)st.node_count, (unsigned long long)st.surfel_count,
		            (unsigned long long)st.ray_count, st.ms_upload, st.ms_bvh_build, st.ms_rasterize, st.ms_trace, st.ms_resolve, st.ms_download);
		for (int i = 0; i < scene->GetMaterialCount(); i++)
		{
			BakeTexture* tex = scene->GetMaterial(i)->GetAOTexture();
			std::string path = out + "_" + std::to_string(i) + ".raw";
			FILE* f = std::fopen(path.c_str(), "wb");
			if (f) { std::fwrite(tex->GetData(), 1, tex->GetDataLength(), f); std::fclose(f); }
			if (png && !tex->SavePNG((out + "_" + std::to_string(i) + ".png").c_str())) std::fprintf(stderr, "cannot write %s_%d.png\n", out.c_str(), i);
			if (exr && !tex->SaveEXR((out + "_" + std::to_string(i) + ".exr").c_str())) std::fprintf(stderr, "cannot write %s_%d.exr\n", out.c_str(), i);
			if (tex->GetChannelCount() == 1 && tex->GetBytesPerChannel() == 1)
			{
				path = out + "_" + std::to_string(i) + ".pgm";
				f = std::fopen(path.c_str(), "wb");
				if (f) { std::fprintf(f, "P5\n%d %d\n255\n", tex->GetWidth(), tex->GetHeight()); std::fwrite(tex->GetData(), 1, tex->GetDataLength(), f); std::fclose(f); }
			}
		}
	}
	BakeEngine::Dispose();                    // Main.cpp:21
	delete scene;
	return status == 0 ? 0 : 1;
}

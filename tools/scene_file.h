// scene_file.h — reader of the .aobs scene files aobaker_amd/scenefile.py writes (shared by aobake_cli and aobake_mgpu).
//
// Little endian:
//   "AOBS" u32 version=1, u32 material_count, u32 object_count
//   per material: i32 width, height, channels, bytes_per_channel
//   per object:   f32 position[3], rotation[3], scale[3], u32 section_count
//     per section: i32 material, u32 vertex_count, u32 index_count, RenderVertex[vertex_count], u32[index_count]
#pragma once

#include "../include/aobaker_scene.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

static inline bool read_exact(FILE* f, void* dst, size_t n) { return std::fread(dst, 1, n, f) == n; }

static inline Scene* load_scene(const char* path)
{
	FILE* f = std::fopen(path, "rb");
	if (!f) { std::perror(path); return nullptr; }
	char magic[4]; uint32_t version, nmat, nobj;
	if (!read_exact(f, magic, 4) || std::memcmp(magic, "AOBS", 4) || !read_exact(f, &version, 4) || version != 1 ||
	    !read_exact(f, &nmat, 4) || !read_exact(f, &nobj, 4)) { std::fprintf(stderr, "%s: bad header\n", path); std::fclose(f); return nullptr; }
	Scene* scene = new Scene();
	for (uint32_t i = 0; i < nmat; i++)
	{
		int32_t d[4];
		if (!read_exact(f, d, sizeof d)) { std::fclose(f); delete scene; return nullptr; }
		scene->AddMaterial(new MaterialObject(L"Material", d[0], d[1], d[2], d[3]));
	}
	for (uint32_t i = 0; i < nobj; i++)
	{
		float trs[9]; uint32_t nsec;
		if (!read_exact(f, trs, sizeof trs) || !read_exact(f, &nsec, 4)) { std::fclose(f); delete scene; return nullptr; }
		BakeObject* object = new BakeObject(L"Object");
		object->SetPosition({ trs[0], trs[1], trs[2] });
		object->SetRotation({ trs[3], trs[4], trs[5] });
		object->SetScale({ trs[6], trs[7], trs[8] });
		for (uint32_t s = 0; s < nsec; s++)
		{
			int32_t material; uint32_t nv, ni;
			if (!read_exact(f, &material, 4) || !read_exact(f, &nv, 4) || !read_exact(f, &ni, 4)) { std::fclose(f); delete scene; return nullptr; }
			std::vector<RenderVertex> v(nv); std::vector<unsigned int> idx(ni);
			if (!read_exact(f, v.data(), sizeof(RenderVertex) * nv) || !read_exact(f, idx.data(), 4ull * ni)) { std::fclose(f); delete scene; return nullptr; }
			MaterialObject* mat = material >= 0 && material < scene->GetMaterialCount() ? scene->GetMaterial(material) : nullptr;
			object->AddSection(new BakeObject::Section(L"Geometry", new RenderMesh(v.data(), (int)nv, idx.data(), (int)ni), mat));
		}
		scene->AddObject(object);
	}
	std::fclose(f);
	return scene;
}


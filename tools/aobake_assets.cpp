// aobake_assets — the asset formats of the bake without a GPU: OBJ import and PNG export through the same
// BakeObject::LoadFromFile / BakeTexture::SavePNG the CLI uses (aobaker_amd/host/Assets.cpp).
//
//   aobake_assets obj2aobs <in.obj> <out.aobs> <width> <height> <channels> <bytes_per_channel>
//       import the OBJ into a scene and write it in aobake_cli's scene format
//   aobake_assets png <in.raw> <out.png> <width> <height> <channels> <bytes_per_channel>
//       wrap raw texture bytes (BakeTexture::Data layout) in a BakeTexture and save it
//   aobake_assets exr <in.raw> <out.exr> <width> <height> <channels> <bytes_per_channel>
//       the same through BakeTexture::SaveEXR
#include "../include/aobaker_scene.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

static bool write_scene(const Scene& scene, const char* path)
{
	FILE* f = std::fopen(path, "wb");
	if (!f) return false;
	const uint32_t head[3] = { 1u, (uint32_t)scene.GetMaterialCount(), (uint32_t)scene.GetObjectCount() };
	std::fwrite("AOBS", 1, 4, f); std::fwrite(head, 4, 3, f);
	for (int i = 0; i < scene.GetMaterialCount(); i++)
	{
		const BakeTexture* t = scene.GetMaterial(i)->GetAOTexture();
		const int32_t d[4] = { t->GetWidth(), t->GetHeight(), t->GetChannelCount(), t->GetBytesPerChannel() };
		std::fwrite(d, 4, 4, f);
	}
	for (int i = 0; i < scene.GetObjectCount(); i++)
	{
		const BakeObject* o = scene.GetObject(i);
		const float trs[9] = { o->GetPosition().x, o->GetPosition().y, o->GetPosition().z, o->GetRotation().x, o->GetRotation().y, o->GetRotation().z,
		                       o->GetScale().x, o->GetScale().y, o->GetScale().z };
		const uint32_t nsec = (uint32_t)o->GetSectionCount();
		std::fwrite(trs, 4, 9, f); std::fwrite(&nsec, 4, 1, f);
		for (int s = 0; s < o->GetSectionCount(); s++)
		{
			const BakeObject::Section* sec = o->GetSection(s);
			const RenderMesh* mesh = sec->GetPreviewMesh();
			const int32_t material = scene.FindMaterial(sec->GetMaterial());
			const uint32_t counts[2] = { (uint32_t)mesh->GetVertexCount(), (uint32_t)mesh->GetIndexCount() };
			std::fwrite(&material, 4, 1, f); std::fwrite(counts, 4, 2, f);
			std::fwrite(mesh->GetVertices(), sizeof(RenderVertex), counts[0], f);
			std::fwrite(mesh->GetIndices(), 4, counts[1], f);
		}
	}
	return std::fclose(f) == 0;
}

int main(int argc, char** argv)
{
	if (argc == 8 && std::strcmp(argv[1], "obj2aobs") == 0)
	{
		Scene scene;
		BakeObject* object = new BakeObject(L"Object");
		scene.AddObject(object);
		if (!object->LoadFromFile(argv[2], &scene, std::atoi(argv[4]), std::atoi(argv[5]), std::atoi(argv[6]), std::atoi(argv[7])))
		{
			std::fprintf(stderr, "%s: cannot import\n", argv[2]);
			return 1;
		}
		return write_scene(scene, argv[3]) ? 0 : 1;
	}
	if (argc == 8 && (std::strcmp(argv[1], "png") == 0 || std::strcmp(argv[1], "exr") == 0))
	{
		BakeTexture texture(std::atoi(argv[4]), std::atoi(argv[5]), std::atoi(argv[6]), std::atoi(argv[7]));
		FILE* f = std::fopen(argv[2], "rb");
		if (!f || std::fread(texture.GetData(), 1, texture.GetDataLength(), f) != texture.GetDataLength()) { std::fprintf(stderr, "%s: short read\n", argv[2]); return 1; }
		std::fclose(f);
		return (argv[1][0] == 'p' ? texture.SavePNG(argv[3]) : texture.SaveEXR(argv[3])) ? 0 : 1;
	}
	std::fprintf(stderr, "usage: aobake_assets obj2aobs in.obj out.aobs W H channels bpc | png|exr in.raw out.png|out.exr W H channels bpc\n");
	return 2;
}

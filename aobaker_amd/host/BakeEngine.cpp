// BakeEngine.cpp — C++ host side of the drop-in: BakeEngine::{Init,Dispose,Bake} over aobaker.h.
//
// Mirrors the structure of the reference's Bake (BakeEngine.cpp:335-751) at the level a caller
// can observe: every (object, section) becomes one shape/mesh in scene order (:370-398), every
// material's AO texture is written in place, tightly packed (:705-709), then RefreshPreview()
// is called on it (:734).  Everything in between runs in the HIP library.
#include "../../include/BakeEngine.h"
#include "../../include/aobaker.h"
#include "../../include/aobaker_multi.h"
#include "../../include/aobaker_scene.h"

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstring>
#include <vector>

namespace {
AobOptions g_options;
bool g_have_options = false;
AobStats g_stats;
int g_status = AOB_OK;
int g_device_count = 1;

void flatten(Scene* scene, std::vector<AobMeshDesc>& meshes, std::vector<AobMaterialDesc>& materials)
{
	for (int i = 0; i < scene->GetObjectCount(); i++)
	{
		BakeObject* object = scene->GetObject(i);
		const DirectX::SimpleMath::Matrix transform = object->BuildTransform();
		for (int s = 0; s < object->GetSectionCount(); s++)
		{
			BakeObject::Section* section = object->GetSection(s);
			RenderMesh* mesh = section->GetPreviewMesh();
			AobMeshDesc d;
			std::memset(&d, 0, sizeof d);
			// a RenderVertex* is passed as is: stride 48, position @0, normal @12, uv @24
			const unsigned char* base = reinterpret_cast<const unsigned char*>(mesh->GetVertices());
			d.positions = base + offsetof(RenderVertex, Position);
			d.normals = base + offsetof(RenderVertex, Normal);
			d.uvs = base + offsetof(RenderVertex, UV);
			d.position_stride = d.normal_stride = d.uv_stride = sizeof(RenderVertex);
			d.vertex_count = (uint32_t)mesh->GetVertexCount();
			d.indices = mesh->GetIndices();
			d.index_count = (uint32_t)mesh->GetIndexCount();
			// material <-> section association is by pointer equality (BakeEngine.cpp:523)
			d.material = scene->FindMaterial(section->GetMaterial());
			std::memcpy(d.model, transform.m, sizeof d.model);
			// BakeObject.h:28-31: declared by the reference, never read by its Bake; here they mask rays (aobaker.h)
			d.flags = (section->IsVisibleToSelf() ? 0u : (uint32_t)AOB_MESH_HIDDEN_FROM_SELF) |
			          (section->IsVisibleToOthers() ? 0u : (uint32_t)AOB_MESH_HIDDEN_FROM_OTHERS);
			meshes.push_back(d);
		}
	}
	for (int i = 0; i < scene->GetMaterialCount(); i++)
	{
		BakeTexture* tex = scene->GetMaterial(i)->GetAOTexture();
		AobMaterialDesc d;
		std::memset(&d, 0, sizeof d);
		d.ao_max_range = scene->GetMaterial(i)->GetAOMaxRange();     // used only with AobOptions::material_ranges
		d.width = tex->GetWidth();
		d.height = tex->GetHeight();
		d.channels = tex->GetChannelCount();
		d.bytes_per_channel = tex->GetBytesPerChannel();
		d.data = tex->GetData();
		materials.push_back(d);
	}
}
}

void BakeEngine::Flatten(Scene* scene, std::vector<AobMeshDesc>& meshes, std::vector<AobMaterialDesc>& materials)
{
	flatten(scene, meshes, materials);
}

AobOptions& BakeEngine::Options()
{
	if (!g_have_options) { aob_default_options(&g_options); g_have_options = true; }
	return g_options;
}

int BakeEngine::LastStatus() { return g_status; }
const char* BakeEngine::LastError() { return aob_last_error(); }
const AobStats& BakeEngine::LastStats() { return g_stats; }

void BakeEngine::Init()
{
	g_status = aob_init(nullptr);
	if (g_status != AOB_OK) std::fprintf(stderr, "BakeEngine::Init: %s\n", aob_last_error());
}

void BakeEngine::Dispose()
{
	aob_shutdown();
}

void BakeEngine::SetDeviceCount(int count)
{
	if (count <= 0)
	{
		count = 1;
		if (hipGetDeviceCount(&count) != hipSuccess || count < 1) count = 1;
	}
	g_device_count = count;
}

int BakeEngine::DeviceCount() { return g_device_count; }

// Bake(Scene*) over g_device_count GPUs: upload to every device, one step, device 0 downloads (aobaker_multi.h)
static int bake_multi(const AobSceneDesc& desc)
{
	std::vector<int> devices;
	for (int i = 0; i < g_device_count; i++) devices.push_back(i);
	AobMultiGpuBake* multi = AobMultiGpuBake::Create(&desc, &BakeEngine::Options(), devices, false);
	if (!multi) { std::fprintf(stderr, "BakeEngine::Bake: %s\n", AobMultiGpuBake::LastError()); return AOB_ERR_DEVICE; }
	int status = multi->Step(nullptr);
	if (status == AOB_OK) status = multi->Download();
	if (status == AOB_OK) status = multi->GetStats(&g_stats);
	if (status != AOB_OK) std::fprintf(stderr, "BakeEngine::Bake: %s\n", AobMultiGpuBake::LastError());
	delete multi;
	return status;
}

void BakeEngine::Bake(Scene* scene)
{
	std::vector<AobMeshDesc> meshes;
	std::vector<AobMaterialDesc> materials;
	flatten(scene, meshes, materials);
	AobSceneDesc desc = { meshes.data(), (uint32_t)meshes.size(), materials.data(), (uint32_t)materials.size() };
	if (g_device_count > 1)
	{
		g_status = bake_multi(desc);
		if (g_status != AOB_OK) return;
		for (int i = 0; i < scene->GetMaterialCount(); i++)
			scene->GetMaterial(i)->GetAOTexture()->RefreshPreview();
		return;
	}
	g_status = aob_bake(&desc, &Options(), &g_stats);
	if (g_status != AOB_OK) { std::fprintf(stderr, "BakeEngine::Bake: %s\n", aob_last_error()); return; }
	for (int i = 0; i < scene->GetMaterialCount(); i++)
		scene->GetMaterial(i)->GetAOTexture()->RefreshPreview();
}

void BakeEngine::Bake(Scene* scene, BakeTexture* texture)
{
	std::vector<AobMeshDesc> meshes;
	std::vector<AobMaterialDesc> materials;
	flatten(scene, meshes, materials);
	AobSceneDesc desc = { meshes.data(), (uint32_t)meshes.size(), materials.data(), (uint32_t)materials.size() };
	AobBake* bake = nullptr;
	g_status = aob_bake_create(&desc, &Options(), &bake);
	if (g_status == AOB_OK) g_status = aob_bake_build_bvh(bake);
	bool found = false;
	for (int i = 0; g_status == AOB_OK && i < scene->GetMaterialCount(); i++)
	{
		if (scene->GetMaterial(i)->GetAOTexture() != texture) continue;
		found = true;
		if (g_status == AOB_OK) g_status = aob_bake_rasterize(bake, i);
		if (g_status == AOB_OK) g_status = aob_bake_trace(bake, i, 0, UINT64_MAX);
		if (g_status == AOB_OK) g_status = aob_bake_resolve(bake, i);
		if (g_status == AOB_OK) g_status = aob_bake_download(bake, i, nullptr, 0);
	}
	if (g_status == AOB_OK) g_status = aob_bake_get_stats(bake, &g_stats);
	if (bake) aob_bake_destroy(bake);
	if (g_status != AOB_OK) { std::fprintf(stderr, "BakeEngine::Bake: %s\n", aob_last_error()); return; }
	if (found) texture->RefreshPreview();
}

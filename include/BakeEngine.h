// BakeEngine.h — the reference's entry points (BakeEngine.h:5-10), kept verbatim,
// plus the two-argument overload BASELINE.json's north star names.
//
//   namespace BakeEngine { void Init(); void Dispose(); void Bake(Scene* scene); }
//
// Behind them sits the MI355X HIP library through the C ABI of aobaker.h; the
// implementation (aobaker_amd/host/BakeEngine.cpp) walks the same accessors the
// reference's Bake walks and nothing else.
#pragma once

#include <vector>

class Scene;
class BakeTexture;
struct AobOptions;
struct AobStats;
struct AobMeshDesc;
struct AobMaterialDesc;

namespace BakeEngine
{
	void Init();
	void Dispose();
	void Bake(Scene* scene);
	// Bakes only the material whose GetAOTexture() == texture.
	void Bake(Scene* scene, BakeTexture* texture);

	// Multi-GPU (not in the reference, which bakes on device 0 only, BakeEngine.cpp:368): with a device count above 1
	// Bake() cuts every material's atlas into strips of texel rows, one per GPU of this node — one host thread and one RCCL
	// rank per device inside this process — gathers the strips' result cells on device 0 over xGMI and resolves there
	// (include/aobaker_multi.h).  1 (default): the single-device path; 0: every visible device.
	void SetDeviceCount(int count);
	int DeviceCount();

	// The walk of the scene Bake() starts with (BakeEngine.cpp:370-398 and :404-410): one AobMeshDesc per (object, section) in
	// scene order pointing INTO the scene's RenderVertex / index arrays, one AobMaterialDesc per material pointing at its
	// BakeTexture's data.  For callers of the staged C ABI (aob_bake_*, AobMultiGpuBake) that start from a Scene.
	void Flatten(Scene* scene, std::vector<AobMeshDesc>& meshes, std::vector<AobMaterialDesc>& materials);

	// The reference hard-codes its settings as file-scope constants (BakeEngine.cpp:13-18);
	// they are the defaults here and can be replaced before a bake.
	AobOptions& Options();
	// Status of the last Bake: the reference returns void and ignores every HRESULT; the
	// signatures stay void for source compatibility and failures are reported here
	// (0 = ok, otherwise an AobStatus; LastError() is aob_last_error()).
	int LastStatus();
	const char* LastError();
	const AobStats& LastStats();
}

// BakeEngine.h — the reference's entry points (BakeEngine.h:5-10), kept verbatim,
// plus the two-argument overload BASELINE.json's north star names.
//
//   namespace BakeEngine { void Init(); void Dispose(); void Bake(Scene* scene); }
//
// Behind them sits the MI355X HIP library through the C ABI of aobaker.h; the
// implementation (aobaker_amd/host/BakeEngine.cpp) walks the same accessors the
// reference's Bake walks and nothing else.
#pragma once

class Scene;
class BakeTexture;
struct AobOptions;
struct AobStats;

namespace BakeEngine
{
	void Init();
	void Dispose();
	void Bake(Scene* scene);
	// Bakes only the material whose GetAOTexture() == texture.
	void Bake(Scene* scene, BakeTexture* texture);

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

// api.hip — the C ABI of include/aobaker.h: device/stream setup, scene upload,
// stage orchestration and read-back.  Host logic only; the kernels live in
// bvh_build.hip, raster.hip, trace.hip and resolve.hip.
#include "aob_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <rocprim/rocprim.hpp>

namespace aob {

static thread_local char g_error[512] = "";
// per host thread, like HIP's current device: a multi-GPU host runs one thread per device (aobaker_amd/host/MultiGpu.cpp)
static thread_local bool g_initialised = false;
static thread_local int g_device = 0;
static std::atomic<int> g_process_inits{0};     // threads between their aob_init and their aob_shutdown
// the device of the most recent aob_init on ANY thread (-1: none, or shut down): what a thread that has not called aob_init
// itself creates handles on — a host that initialises once on its main thread and bakes from a worker keeps working
static std::atomic<int> g_process_device{-1};

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
}


// BakeEngine.cpp:312-333 with the MSVC CRT rand() the reference draws from (never seeded, so
// the state starts at 1): x = sin(theta)cos(phi), z = sin(theta)sin(phi), y = 1 - sin(theta).
// Entries [first, last) of the table.  The generator is the LCG s -> 214013 s + 2531011 (mod 2^32), two draws per entry: its
// state before entry `first` is 2 * first steps from 1, reached in O(log) by squaring the affine map — so every host thread
// fills its own run of entries and the bytes are those of the serial loop (same libm calls, same order per entry).
static void hemisphere_entries(float* out, uint64_t first, uint64_t last)
{
    uint32_t mul = 1u, add = 0u;                    // the map applied 2 * first times: s -> mul * s + add
    uint32_t m = 214013u, a = 2531011u;
    for (uint64_t k = 2 * first; k; k >>= 1) {
        if (k & 1) { mul = mul * m; add = add * m + a; }
        a = a * m + a; m = m * m;
    }
    uint32_t state = mul * 1u + add;
    auto msvc_rand = [&]() { state = state * 214013u + 2531011u; return (state >> 16) & 0x7fffu; };
    const float two_pi = 2.0f * 3.14159265358979323846f;
    for (uint64_t i = first; i < last; i++) {
        float u = (float)msvc_rand() / 32767.0f;
        float v = (float)msvc_rand() / 32767.0f;
        float sintheta = std::sqrt(-u * (u - 2.0f));
        float phi = two_pi * v;
        float x = sintheta * std::cos(phi);
        float z = sintheta * std::sin(phi);
        float len = std::sqrt(x * x + z * z);
        out[3 * i + 0] = x; out[3 * i + 1] = 1.0f - len; out[3 * i + 2] = z;
    }
}

// The table of the first `count` entries, filled on all host cores and kept for the process: a prefix of the sequence is the
// same for every bake, so a later bake that needs no more entries reuses it (the reference refills its 16.7 M entries on every
// Bake, serially: BakeEngine.cpp:348-356 — 98 ms of a 160 ms one-shot bake of BASELINE configs[2]'s terrain here, before this).
static std::mutex g_table_mutex;
static std::shared_ptr<const std::vector<float>> g_table;       // 3 floats per entry; grown by replacement, never in place:
                                                                // a bake that is still uploading from an older one keeps it alive
static std::shared_ptr<const std::vector<float>> hemisphere_table(uint64_t count)
{
    std::lock_guard<std::mutex> lock(g_table_mutex);
    const uint64_t have = g_table ? g_table->size() / 3 : 0;
    if (count > have) {
        auto grown = std::make_shared<std::vector<float>>(3 * count);
        if (have) memcpy(grown->data(), g_table->data(), sizeof(float) * 3 * have);
        const uint64_t todo = count - have;
        unsigned threads = std::thread::hardware_concurrency();
        if (threads < 1) threads = 1;
        if (threads > 32) threads = 32;
        if (todo < 65536) threads = 1;
        std::vector<std::thread> pool;
        float* out = grown->data();
        for (unsigned t = 1; t < threads; t++)
            pool.emplace_back(hemisphere_entries, out, have + todo * t / threads, have + todo * (t + 1) / threads);
        hemisphere_entries(out, have, have + todo / threads);
        for (auto& th : pool) th.join();
        g_table = grown;
    }
    return g_table;
}

static int validate_options(const AobOptions* o)
{
    if (!o) { set_error("options is NULL"); return AOB_ERR_INVALID_ARGUMENT; }
    if (o->sample_count <= 0 || o->sample_count > (1 << 20)) { set_error("sample_count %d out of range", o->sample_count); return AOB_ERR_INVALID_ARGUMENT; }
    if (o->sampler != AOB_SAMPLER_REFERENCE_COMPAT && o->sampler != AOB_SAMPLER_COSINE) { set_error("unknown sampler %d", o->sampler); return AOB_ERR_INVALID_ARGUMENT; }
    if (o->accumulate != AOB_ACCUMULATE_QUANTISED && o->accumulate != AOB_ACCUMULATE_FLOAT) { set_error("unknown accumulate mode %d", o->accumulate); return AOB_ERR_INVALID_ARGUMENT; }
    if (o->normal_fp16 < AOB_FP16_OFF || o->normal_fp16 > AOB_FP16_RNE) { set_error("unknown normal_fp16 mode %d", o->normal_fp16); return AOB_ERR_INVALID_ARGUMENT; }
    if (o->bleed_distance < 0 || o->bleed_distance > 64) { set_error("bleed_distance %d out of range", o->bleed_distance); return AOB_ERR_INVALID_ARGUMENT; }
    if (o->bleed_mode != AOB_BLEED_REFERENCE && o->bleed_mode != AOB_BLEED_NEAREST) { set_error("unknown bleed_mode %d", o->bleed_mode); return AOB_ERR_INVALID_ARGUMENT; }
    if (o->raster_mode != AOB_RASTER_REFERENCE && o->raster_mode != AOB_RASTER_CONSERVATIVE) { set_error("unknown raster_mode %d", o->raster_mode); return AOB_ERR_INVALID_ARGUMENT; }
    if (o->raster_layers != AOB_LAYERS_LAST && o->raster_layers != AOB_LAYERS_ALL) { set_error("unknown raster_layers %d", o->raster_layers); return AOB_ERR_INVALID_ARGUMENT; }
    if (o->raster_layers == AOB_LAYERS_ALL && o->raster_mode == AOB_RASTER_CONSERVATIVE) { set_error("raster_layers ALL needs raster_mode REFERENCE"); return AOB_ERR_UNSUPPORTED; }
    if (o->morton_adaptive != 0 && o->morton_adaptive != 1) { set_error("morton_adaptive %d: must be 0 or 1", o->morton_adaptive); return AOB_ERR_INVALID_ARGUMENT; }
    if (o->material_ranges != 0 && o->material_ranges != 1) { set_error("material_ranges %d: must be 0 or 1", o->material_ranges); return AOB_ERR_INVALID_ARGUMENT; }
    if (o->bvh_builder < -1 || o->bvh_builder > 2) { set_error("bvh_builder %d: must be -1 (by workload), 0 (LBVH), 1 (PLOC) or 2 (PLOC clusters under a binned-SAH top)", o->bvh_builder); return AOB_ERR_INVALID_ARGUMENT; }
    if (o->bvh_width != 0 && o->bvh_width != 2 && o->bvh_width != 4) { set_error("bvh_width %d: must be 0, 2 or 4", o->bvh_width); return AOB_ERR_INVALID_ARGUMENT; }
    // (the quad-per-ray walk of the fp32 4-ary nodes was an experiment, 2.7x slower than the default; retired in round 5: tools/experiments/trace4.hip)
    if (o->bvh_width == 4 && !o->bvh_quantised) { set_error("bvh_width 4 needs bvh_quantised 1 (the fp32 4-ary walk is retired: tools/experiments/trace4.hip); the fp32 nodes are walked with bvh_width 2"); return AOB_ERR_UNSUPPORTED; }
    if (o->trace_refill_idle < 0 || o->trace_refill_idle > 64 || o->trace_leaf_wait < 0 || o->trace_leaf_wait > 64) { set_error("trace_refill_idle %d / trace_leaf_wait %d: must be 0 (default) or 1..64", o->trace_refill_idle, o->trace_leaf_wait); return AOB_ERR_INVALID_ARGUMENT; }
    if (o->trace_parts < 0 || o->trace_parts > 64 || o->trace_part_min < 0 || o->trace_fine_div < 0) { set_error("trace_parts %d (0..64), trace_part_min %d, trace_fine_div %d: must not be negative", o->trace_parts, o->trace_part_min, o->trace_fine_div); return AOB_ERR_INVALID_ARGUMENT; }
    if (o->trace_scalar_min < 0 || o->trace_scalar_min > 64) { set_error("trace_scalar_min %d: must be 0 (default) or 1..64", o->trace_scalar_min); return AOB_ERR_INVALID_ARGUMENT; }
    if (o->trace_cut < 0 || o->trace_cut > 32 || o->trace_cut == 2 || o->trace_cut == 3) { set_error("trace_cut %d: must be 0 (default: 32 entries), 1 (off) or 4..32", o->trace_cut); return AOB_ERR_INVALID_ARGUMENT; }
    if (o->trace_horizon < 0 || o->trace_horizon > 2) { set_error("trace_horizon %d: must be 0 (default: horizon tables from 128 samples per surfel on), 1 (off) or 2 (wherever they apply)", o->trace_horizon); return AOB_ERR_INVALID_ARGUMENT; }
    if (o->trace_bands < 0 || o->trace_bands > 2) { set_error("trace_bands %d: must be 0 (default), 1 or 2", o->trace_bands); return AOB_ERR_INVALID_ARGUMENT; }
    if (!(o->falloff_distance > 0.0f)) { set_error("falloff_distance must be positive"); return AOB_ERR_INVALID_ARGUMENT; }
    return AOB_OK;
}

static bool valid_format(int channels, int bpc)
{
    // Scene/BakeTexture.cpp:9-49
    if (bpc == 1 || bpc == 2) return channels == 1 || channels == 2 || channels == 4;
    if (bpc == 4) return channels >= 1 && channels <= 4;
    return false;
}

// Device memory of a bake is ONE allocation: every buffer aob_bake_create sizes is a 256-byte aligned slice of `arena`
// (50 hipFree calls, each a device synchronisation, made destroying a bake cost 5.8 ms — 7 % of the one-shot aob_bake of
// BASELINE configs[2]; one hipFree costs 0.1).  want() records a slice, carve() allocates and hands the slices out.
struct ArenaPlan {
    struct Want { void** slot; size_t bytes; };
    std::vector<Want> wants;
    template <class T> void want(T** p, uint64_t count) { *p = nullptr; wants.push_back({ (void**)p, sizeof(T) * (count ? count : 1) }); }
    void want_bytes(void** p, size_t bytes) { *p = nullptr; wants.push_back({ p, bytes ? bytes : 1 }); }
};

// ---- the pool of released device buffers (aob_internal.h) ----
struct PoolEntry { int device; void* p; size_t bytes; };
static std::mutex g_pool_mutex;
static std::vector<PoolEntry> g_pool;
static constexpr size_t kPoolEntries = 12;                 // buffers kept per process
static constexpr size_t kPoolBytes = 64ull << 30;          // ... and their total size (an MI355X has 288 GB)

hipError_t pool_alloc(void** p, size_t bytes, int device)
{
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        // the smallest kept buffer that is large enough and not more than twice (+ 1 MB) what is asked for
        int best = -1;
        for (size_t i = 0; i < g_pool.size(); i++)
            if (g_pool[i].device == device && g_pool[i].bytes >= bytes && g_pool[i].bytes <= 2 * bytes + (1u << 20) && (best < 0 || g_pool[i].bytes < g_pool[(size_t)best].bytes)) best = (int)i;
        if (best >= 0) { *p = g_pool[(size_t)best].p; g_pool.erase(g_pool.begin() + best); return hipSuccess; }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        // out of memory with buffers kept aside: give them back and try once more
        (void)hipGetLastError();
        pool_clear();
        e = hipMalloc(p, bytes);
    }
    return e;
}

void pool_free(void* p, size_t bytes, int device)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        size_t total = bytes;
        for (const auto& e : g_pool) total += e.bytes;
        if (g_process_inits.load() > 0 && g_pool.size() < kPoolEntries && total <= kPoolBytes && bytes >= (1u << 20)) { g_pool.push_back({ device, p, bytes }); return; }
    }
    hipFree(p);
}

void pool_clear()
{
    std::vector<PoolEntry> drop;
    { std::lock_guard<std::mutex> lock(g_pool_mutex); drop.swap(g_pool); }
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (const auto& e : drop) { (void)hipSetDevice(e.device); hipFree(e.p); }
    if (prev >= 0) (void)hipSetDevice(prev);
}

static int carve(AobBake* b, const ArenaPlan& plan)
{
    size_t total = 0;
    for (const auto& w : plan.wants) total += (w.bytes + 255) & ~(size_t)255;
    AOB_HIP(pool_alloc((void**)&b->arena, total ? total : 256, b->device));
    b->arena_bytes = total ? total : 256;
    size_t at = 0;
    for (const auto& w : plan.wants) { *w.slot = b->arena + at; at += (w.bytes + 255) & ~(size_t)255; }
    return AOB_OK;
}

static void free_bake(AobBake* b)
{
    if (!b) return;
    // buffers that were allocated (or re-allocated: the surfel lists of overlapping sections can outgrow W*H) on their own
    for (auto& m : b->materials) {
        release_device(b, m.prev_layer); release_device(b, m.surfels); release_device(b, m.section_index); release_device(b, m.frames); release_device(b, m.perm);
        if (m.tile_cuts) pool_free(m.tile_cuts, sizeof(uint32_t) * m.tile_cut_capacity, b->device);
        pool_free(m.horizon, m.horizon_capacity, b->device);
        if (m.tile_row_prefix_host) hipHostFree(m.tile_row_prefix_host);
    }
    release_device(b, b->gstack);
    release_device(b, b->acc_cells);
    if (b->arena) pool_free(b->arena, b->arena_bytes, b->device);
    if (b->pinned) hipHostFree(b->pinned);
    for (auto& e : b->ev) if (e) hipEventDestroy(e);
    for (auto& e : b->ev_trace) if (e) hipEventDestroy(e);
    for (auto& e : b->ev_part) if (e) hipEventDestroy(e);
    for (auto& e : b->ev_ploc) if (e) hipEventDestroy(e);
    if (b->ev_peer) hipEventDestroy(b->ev_peer);
    if (b->stream) hipStreamDestroy(b->stream);
    delete b;
}

// A handle lives on the device it was created on, whatever the calling thread's current device is: every entry point selects
// that device for its own duration and gives the caller's device back when it returns — a host thread that shares HIP with
// other code (torch, its own kernels on another GPU) does not find its current device changed by an aob_bake_* call.
struct DeviceScope {
    int prev = -1;
    bool changed = false;
    ~DeviceScope() { if (changed) hipSetDevice(prev); }
};

static int check_bake(DeviceScope& scope, AobBake* b, int32_t material, bool need_material)
{
    if (!b) { set_error("bake handle is NULL"); return AOB_ERR_INVALID_ARGUMENT; }
    if (!scope.changed) {
        AOB_HIP(hipGetDevice(&scope.prev));
        if (scope.prev != b->device) { AOB_HIP(hipSetDevice(b->device)); scope.changed = true; }
    } else {
        AOB_HIP(hipSetDevice(b->device));
    }
    if (need_material && (material < 0 || (size_t)material >= b->materials.size())) {
        set_error("material %d out of range (%zu)", material, b->materials.size());
        return AOB_ERR_INVALID_ARGUMENT;
    }
    return AOB_OK;
}

static int ensure_depth(AobBake* b)
{
    if (b->depth_known) return AOB_OK;
    AOB_HIP(hipStreamSynchronize(b->stream));
    b->bvh_depth = b->node_count > 1 ? b->pinned[1] : 0;
    b->depth_known = true;
    // the walk's HBM overflow stack, for the deepest walk this tree allows: allocated here, once per tree, not by a trace launch
    return reserve_trace_stack(b);
}

} // namespace aob

using namespace aob;

// cells += other's cells, word by word, as an RCCL sum-reduce of the same buffers adds them
template <bool AS_FLOAT>
__global__ __launch_bounds__(256)
void k_add_results(uint32_t* __restrict__ cells, const uint32_t* __restrict__ other, uint64_t words)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= words) return;
    cells[i] = AS_FLOAT ? __float_as_uint(__uint_as_float(cells[i]) + __uint_as_float(other[i])) : cells[i] + other[i];
}

extern "C" {

int aob_abi_version(void) { return AOB_ABI_VERSION; }

const char* aob_last_error(void) { return g_error; }

// bvh_builder -1: the builder by workload.  The SAH top costs 0.35 .. 0.64 ms more build than PLOC alone and takes 5 % of the node visits off
// any-hit rays on the terrain (61.0 -> 58.3 ms of trace at 1.07 G rays — BASELINE configs[2] —, about even at 67 M —
// configs[1]); closest-hit rays do not gain (configs[4]: 1.7 .. 2.5 % slower); and a scene of at most kSahTopClusters triangles would get a
// full top-down tree over triangle centres, which long thin triangles punish (hairball: +57 % visits) — there PLOC stays.
// The ray count is not known before the rasterisation: texels x samples bounds it from above.
int aob_resolve_bvh_builder(const AobOptions* options, uint64_t texels, uint64_t triangles)
{
    if (!options) return 1;
    if (options->bvh_builder >= 0) return options->bvh_builder;
    const uint64_t rays = texels * (uint64_t)(options->sample_count > 0 ? options->sample_count : 0);
    return (options->use_any_hit && rays >= (1ull << 28) && triangles > 131072ull) ? 2 : 1;
}

void aob_default_options(AobOptions* o)
{
    if (!o) return;
    memset(o, 0, sizeof *o);
    // BakeEngine.cpp:13-18 and GenerateRaysComputeShader.hlsl:47,49
    o->sample_count = 100;
    o->sampler = AOB_SAMPLER_REFERENCE_COMPAT;
    o->seed = 0;
    o->use_any_hit = 1;
    o->falloff_distance = 1.25f;
    o->ray_tmax = 1000000.0f;
    o->ray_bias = 0.001f;
    o->bleed_distance = 5;
    o->bleed_mode = AOB_BLEED_REFERENCE;
    o->accumulate = AOB_ACCUMULATE_QUANTISED;
    o->normal_fp16 = AOB_FP16_RNE;
    o->ray_chunk_size = 0xFFFFFF;
    o->clamp_tmax_to_falloff = 0;
    // measured fastest on MI355X (profiles/): lane-per-ray walk of the 64-byte quantised 4-ary nodes
    o->bvh_width = 4;
    o->bvh_quantised = 1;
    o->bvh_builder = AOB_BVH_BUILDER_AUTO;   // by workload: PLOC, or PLOC under the binned-SAH top where the bake is long enough to pay for it (aob_resolve_bvh_builder)
    o->morton_adaptive = 1;
    o->raster_mode = AOB_RASTER_REFERENCE;
    o->raster_layers = AOB_LAYERS_LAST;
    // schedule of k_trace: the measured optima on BASELINE configs[2]
    o->trace_refill_idle = 0;     // by scoring mode (trace.hip): COSINE sampler with any-hit scoring 16 (42 without tile cuts), table sampler with any-hit scoring 24, closest-hit scoring 8 (16 without tile cuts)
    o->trace_leaf_wait = 0;       // by scoring mode: 24 with any-hit scoring, 16 with closest-hit scoring (trace.hip)
    o->trace_parts = 4;
    o->trace_part_min = 16;
    o->trace_fine_div = 8;
    o->trace_bands = 1;
    o->trace_scalar_min = 0;       // by sampler (trace.hip)
    o->trace_cut = 0;              // tile cuts of 32 entries where they apply (trace.hip)
    o->trace_horizon = 0;          // horizon tables where they apply (horizon.hip)
}

int aob_init(const AobDeviceConfig* config)
{
    int count = 0;
    AOB_HIP(hipGetDeviceCount(&count));
    if (count <= 0) { set_error("no HIP device visible"); return AOB_ERR_DEVICE; }
    int dev = config ? config->device : -1;
    if (dev < 0) AOB_HIP(hipGetDevice(&dev));
    if (dev >= count) { set_error("device %d out of range (%d visible)", dev, count); return AOB_ERR_INVALID_ARGUMENT; }
    AOB_HIP(hipSetDevice(dev));
    hipDeviceProp_t prop;
    AOB_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
        return AOB_ERR_UNSUPPORTED;
    }
    g_device = dev;
    if (!g_initialised) g_process_inits.fetch_add(1);
    g_initialised = true;
    g_process_device.store(dev);
    // the first time on this device: one throw-away bake of a triangle (include/aobaker.h, AOB_INIT_NO_WARMUP)
    static std::atomic<uint64_t> warmed[4];            // one bit per device ordinal, set when the warm-up bake has FINISHED
    static std::mutex warm_mutex;                      // (a second thread's aob_init on the same device waits for it instead of starting a cold bake beside it)
    const uint64_t bit = 1ull << (dev & 63);
    std::unique_lock<std::mutex> warm_lock(warm_mutex, std::defer_lock);
    if (!(config && (config->flags & AOB_INIT_NO_WARMUP)) && dev < 256 && !(warmed[dev >> 6].load() & bit)) warm_lock.lock();
    if (warm_lock.owns_lock() && !(warmed[dev >> 6].load() & bit)) {
        const float pos[9] = { 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f }, nrm[9] = { 0.f, 1.f, 0.f, 0.f, 1.f, 0.f, 0.f, 1.f, 0.f };
        const float uv[6] = { 0.1f, 0.1f, 0.9f, 0.1f, 0.1f, 0.9f };
        const uint32_t idx[3] = { 0u, 1u, 2u };
        AobMeshDesc mesh; memset(&mesh, 0, sizeof mesh);
        mesh.positions = pos; mesh.normals = nrm; mesh.uvs = uv; mesh.position_stride = 12; mesh.normal_stride = 12; mesh.uv_stride = 8;
        mesh.indices = idx; mesh.vertex_count = 3; mesh.index_count = 3; mesh.material = 0;
        mesh.model[0] = mesh.model[5] = mesh.model[10] = mesh.model[15] = 1.0f;
        uint8_t texels[16 * 16] = { 0 };
        AobMaterialDesc mat; memset(&mat, 0, sizeof mat);
        mat.width = 16; mat.height = 16; mat.channels = 1; mat.bytes_per_channel = 1; mat.data = texels;
        AobSceneDesc scene; memset(&scene, 0, sizeof scene);
        scene.meshes = &mesh; scene.mesh_count = 1; scene.materials = &mat; scene.material_count = 1;
        AobOptions o; aob_default_options(&o);
        o.sample_count = 16; o.sampler = AOB_SAMPLER_COSINE;      // (no hemisphere table to fill)
        const int rc = aob_bake(&scene, &o, nullptr);
        if (rc != AOB_OK) { aob_shutdown(); return rc; }
        warmed[dev >> 6].fetch_or(bit);
    }
    return AOB_OK;
}

// aob_init acquires nothing (every stream, event and buffer belongs to an AobBake handle), so there is nothing to release
// (the process-wide fallback device — what threads without an aob_init of their own create handles on — goes with the LAST thread
// that shuts down, not with the first: the workers of a multi-GPU bake shutting down leave a main-thread aob_init alive)
void aob_shutdown(void)
{
    if (!g_initialised) return;
    g_initialised = false;
    if (g_process_inits.fetch_sub(1) == 1) {
        g_process_device.store(-1);
        // the last thread out: the process-wide hemisphere table (REFERENCE_COMPAT bakes: up to 200 MB of host memory) goes too; a bake that
        // is still uploading from it keeps its own reference
        {
            std::lock_guard<std::mutex> lock(g_table_mutex);
            g_table.reset();
        }
        pool_clear();          // ... and the device buffers kept for the next bake
    }
}

int aob_bake_create(const AobSceneDesc* scene, const AobOptions* options, AobBake** out)
{
    if (!out) { set_error("out is NULL"); return AOB_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    const int device = g_initialised ? g_device : g_process_device.load();
    if (device < 0) { set_error("aob_init has not been called"); return AOB_ERR_NOT_INITIALISED; }
    if (!scene) { set_error("scene is NULL"); return AOB_ERR_INVALID_ARGUMENT; }
    AOB_TRY(validate_options(options));
    if ((scene->mesh_count && !scene->meshes) || (scene->material_count && !scene->materials)) {
        set_error("scene arrays are NULL"); return AOB_ERR_INVALID_ARGUMENT;
    }
    for (uint32_t i = 0; i < scene->material_count; i++) {
        const AobMaterialDesc& m = scene->materials[i];
        if (m.width <= 0 || m.height <= 0 || m.width > 32768 || m.height > 32768) { set_error("material %u: bad size %dx%d", i, m.width, m.height); return AOB_ERR_INVALID_ARGUMENT; }
        if (!valid_format(m.channels, m.bytes_per_channel)) { set_error("material %u: invalid combination of channelCount %d and bytesPerChannel %d", i, m.channels, m.bytes_per_channel); return AOB_ERR_INVALID_ARGUMENT; }
    }
    for (uint32_t i = 0; i < scene->mesh_count; i++) {
        const AobMeshDesc& m = scene->meshes[i];
        if (m.vertex_count && (!m.positions || !m.normals || !m.uvs)) { set_error("mesh %u: NULL vertex attribute", i); return AOB_ERR_INVALID_ARGUMENT; }
        if (m.index_count && !m.indices) { set_error("mesh %u: NULL indices", i); return AOB_ERR_INVALID_ARGUMENT; }
        if (m.material >= (int32_t)scene->material_count) { set_error("mesh %u: material %d out of range", i, m.material); return AOB_ERR_INVALID_ARGUMENT; }
        if (m.vertex_count && (m.position_stride < 12 || m.normal_stride < 12 || m.uv_stride < 8)) { set_error("mesh %u: attribute stride too small", i); return AOB_ERR_INVALID_ARGUMENT; }
        for (uint32_t k = 0; k < 3 * (m.index_count / 3); k++) {
            if (m.indices[k] >= m.vertex_count) { set_error("mesh %u: index %u out of range (%u vertices)", i, m.indices[k], m.vertex_count); return AOB_ERR_INVALID_ARGUMENT; }
        }
    }

    DeviceScope scope;                                  // (the caller's current device comes back when the call returns)
    AOB_HIP(hipGetDevice(&scope.prev));
    if (scope.prev != device) { AOB_HIP(hipSetDevice(device)); scope.changed = true; }
    AobBake* b = new (std::nothrow) AobBake();
    if (!b) { set_error("out of host memory"); return AOB_ERR_OUT_OF_MEMORY; }
    b->opt = *options;
    b->device = device;
    int rc = AOB_OK;
    auto fail = [&](int code) { free_bake(b); return code; };
#define AOB_C(expr) do { rc = (expr); if (rc != AOB_OK) return fail(rc); } while (0)
#define AOB_CH(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { set_error("%s failed: %s", #call, hipGetErrorString(e__)); return fail(e__ == hipErrorOutOfMemory ? AOB_ERR_OUT_OF_MEMORY : AOB_ERR_DEVICE); } } while (0)

    AOB_CH(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    for (auto& e : b->ev) AOB_CH(hipEventCreate(&e));
    for (auto& e : b->ev_trace) AOB_CH(hipEventCreate(&e));
    for (auto& e : b->ev_part) AOB_CH(hipEventCreate(&e));
    for (auto& e : b->ev_ploc) AOB_CH(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    AOB_CH(hipEventCreateWithFlags(&b->ev_peer, hipEventDisableTiming));
    AOB_CH(hipHostMalloc((void**)&b->pinned, 64));
    AOB_CH(hipDeviceGetAttribute(&b->cu_count, hipDeviceAttributeMultiprocessorCount, device));
    AOB_CH(hipEventRecord(b->ev[0], b->stream));

    // ---- plan: what every stage needs, sized once ----
    ArenaPlan plan;
    plan.want(&b->trace_unit_next, 8);
    plan.want(&b->horizon_counters, 16 + scene->material_count);
    b->meshes.resize(scene->mesh_count);
    uint64_t tri_total = 0;
    for (uint32_t i = 0; i < scene->mesh_count; i++) {
        const AobMeshDesc& src = scene->meshes[i];
        DeviceMesh& d = b->meshes[i];
        d.vertex_count = src.vertex_count;
        d.triangle_count = src.index_count / 3;
        d.triangle_offset = tri_total;
        d.material = src.material;
        d.flags = src.flags;
        if (d.flags & ~(uint32_t)(AOB_MESH_HIDDEN_FROM_SELF | AOB_MESH_HIDDEN_FROM_OTHERS)) { set_error("mesh %u: unknown flags 0x%x", i, d.flags); return fail(AOB_ERR_INVALID_ARGUMENT); }
        if (d.flags) b->any_mesh_flags = true;
        memcpy(d.model, src.model, sizeof d.model);
        tri_total += d.triangle_count;
        plan.want(&d.positions, 3ull * src.vertex_count);
        plan.want(&d.normals, 3ull * src.vertex_count);
        plan.want(&d.uvs, 2ull * src.vertex_count);
        plan.want(&d.indices, 3ull * d.triangle_count);
    }
    b->triangle_count = tri_total;
    const uint64_t n = tri_total;
    {
        // bvh_builder -1: the builder by workload, settled here for the life of the handle
        uint64_t texels = 0;
        for (uint32_t i = 0; i < scene->material_count; i++) texels += (uint64_t)scene->materials[i].width * (uint64_t)scene->materials[i].height;
        b->opt.bvh_builder = aob_resolve_bvh_builder(options, texels, tri_total);
    }
    // shape ids start at 1 (RadeonRays/radeon_rays.h:238): bit s of the set is mesh s - 1
    std::vector<uint32_t> bits((scene->mesh_count + 1 + 31) / 32 + 1, 0u);
    for (uint32_t i = 0; i < scene->mesh_count; i++)
        if (b->meshes[i].flags & AOB_MESH_HIDDEN_FROM_OTHERS) bits[(i + 1) >> 5] |= 1u << ((i + 1) & 31);
    plan.want(&b->hidden_from_others, bits.size());

    // BVH workspace
    plan.want(&b->tri_verts, 3 * n);
    plan.want(&b->prim_bounds, n);
    plan.want(&b->scene_bounds_enc, 8);
    plan.want(&b->codes[0], n); plan.want(&b->codes[1], n);
    plan.want(&b->prim_idx[0], n); plan.want(&b->prim_idx[1], n);
    plan.want(&b->child_left, n); plan.want(&b->child_right, n);
    plan.want(&b->parent, 2 * n);
    plan.want(&b->node_bounds, 2 * n);
    plan.want(&b->refit_flags, n);
    plan.want(&b->depth_dev, 1);
    plan.want(&b->node_height, n);
    plan.want(&b->ploc_sums, n / 1024 + 4);
    plan.want(&b->ploc_totals, 8);
    plan.want(&b->sah_counts, 1024 + 16 + 64 + 8);
    plan.want(&b->sah_big, 1u << 20);
    b->sah_list_capacity = n < 131072u ? n : 131072u;           // kSahTopClusters (bvh_build.hip)
    plan.want(&b->sah_lists, 2 * 28 * b->sah_list_capacity);
    plan.want(&b->nodes, 2 * n);
    plan.want(&b->nodes4, 8 * n);
    plan.want(&b->leaf_tris, 3 * n);
    plan.want(&b->nodes2q, 2 * n);
    plan.want(&b->nodes4q, 4 * n);
    plan.want(&b->packed_index, n);
    plan.want(&b->node4q_count_dev, 1);
    plan.want(&b->quant_frame, 8);
    if (n) {
        // temporary storage of the sort and of the scan (a size query: no pointer is dereferenced)
        size_t temp = 0;
        AOB_CH(rocprim::radix_sort_pairs(nullptr, temp, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                         (size_t)n, 0u, 63u, b->stream));
        size_t temp_scan = 0;
        AOB_CH(rocprim::exclusive_scan(nullptr, temp_scan, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)n, rocprim::plus<uint32_t>(), b->stream));
        if (temp_scan > temp) temp = temp_scan;
        b->sort_temp_bytes = temp;
        plan.want_bytes(&b->sort_temp, temp);
    }

    b->materials.resize(scene->material_count);
    for (uint32_t i = 0; i < scene->material_count; i++) {
        const AobMaterialDesc& src = scene->materials[i];
        DeviceMaterial& d = b->materials[i];
        d.width = src.width; d.height = src.height; d.channels = src.channels; d.bytes_per_channel = src.bytes_per_channel;
        d.host_data = src.data;
        d.ao_max_range = src.ao_max_range;
        const uint64_t cells = (uint64_t)d.width * d.height;
        d.scan_blocks = (uint32_t)((cells + 2047) / 2048);
        plan.want(&d.cover, cells);
        plan.want(&d.block_sums, (uint64_t)d.scan_blocks + 2);
        plan.want(&d.seen, cells);
        plan.want(&d.results, cells);
        plan.want(&d.image, cells * d.channels * d.bytes_per_channel);
        plan.want(&d.surfels, cells);
        plan.want(&d.section_index, cells);
        plan.want(&d.frames, 4 * cells);
        plan.want(&d.perm, cells);
        // (The tile cuts of a trace launch — 512 B per 64-surfel tile and sample part, up to 96 B per covered texel with the COSINE
        // sampler's default schedule — are NOT part of the arena: they are an accelerator, sized by what a rasterisation actually covered
        // (a rank of a multi-GPU bake covers 1/N of the rows), allocated when the rasterisation ends (trace.hip ensure_tile_cuts), and a
        // bake that cannot have them still bakes: it walks from the root and says so, AobStats::trace_cuts = 2.)
        const uint64_t tiles_x = (uint64_t)(d.width + 7) / 8, tiles_y = (uint64_t)(d.height + 7) / 8;
        plan.want(&d.tile_sums, (tiles_x * tiles_y * 64 + 2047) / 2048 + 1);
        plan.want(&d.tile_row_prefix, tiles_y + 1);
        AOB_CH(hipHostMalloc((void**)&d.tile_row_prefix_host, sizeof(uint32_t) * (tiles_y + 1)));
        d.surfel_capacity = cells;
        d.section_offsets.push_back(0);
    }
    // float cells that hold fractions (weighted samples, falloff scoring) with more than one sample part per tile: the parts' exact
    // sums meet in two 64-bit accumulators per cell (trace.hip k_acc_to_cells) — sized for the largest material here, with
    // everything else, so that an out-of-memory error surfaces at creation and no trace launch carries an allocation
    {
        const int parts_default = std::max(1, std::min(options->trace_parts > 0 ? options->trace_parts : 4,
                                                       options->sample_count / (options->trace_part_min > 0 ? options->trace_part_min : 16)));
        const bool direct = options->accumulate == AOB_ACCUMULATE_QUANTISED || (options->sampler == AOB_SAMPLER_COSINE && options->use_any_hit);
        if (!direct && parts_default > 1 && !(options->bvh_width == 4 && !options->bvh_quantised)) {
            uint64_t most = 0;
            for (auto& d : b->materials) most = std::max<uint64_t>(most, (uint64_t)d.width * d.height);
            b->acc_cells_count = most;
            plan.want(&b->acc_cells, 2 * most);
        }
    }
    if (options->sampler == AOB_SAMPLER_REFERENCE_COMPAT) {
        // S0 (BakeEngine.cpp:348-356): only the entries a chunk can index are generated
        b->hemisphere_entries = chunk_texel_max(*options) * (uint64_t)options->sample_count;
        plan.want(&b->hemisphere, 3 * b->hemisphere_entries);
    }

    // ---- one allocation ----
    AOB_C(carve(b, plan));

    // ---- S1 upload (BakeEngine.cpp:370-398): de-interleave the strided vertex attributes ----
    std::vector<float> staging;
    for (uint32_t i = 0; i < scene->mesh_count; i++) {
        const AobMeshDesc& src = scene->meshes[i];
        DeviceMesh& d = b->meshes[i];
        if (src.vertex_count) {
            staging.resize(8ull * src.vertex_count);
            float* sp = staging.data(); float* sn = sp + 3ull * src.vertex_count; float* su = sn + 3ull * src.vertex_count;
            for (uint32_t v = 0; v < src.vertex_count; v++) {
                memcpy(sp + 3ull * v, (const uint8_t*)src.positions + (size_t)src.position_stride * v, 12);
                memcpy(sn + 3ull * v, (const uint8_t*)src.normals + (size_t)src.normal_stride * v, 12);
                memcpy(su + 2ull * v, (const uint8_t*)src.uvs + (size_t)src.uv_stride * v, 8);
            }
            AOB_CH(hipMemcpyAsync(d.positions, sp, 12ull * src.vertex_count, hipMemcpyHostToDevice, b->stream));
            AOB_CH(hipMemcpyAsync(d.normals, sn, 12ull * src.vertex_count, hipMemcpyHostToDevice, b->stream));
            AOB_CH(hipMemcpyAsync(d.uvs, su, 8ull * src.vertex_count, hipMemcpyHostToDevice, b->stream));
            AOB_CH(hipStreamSynchronize(b->stream));     // staging is reused by the next mesh
        }
        if (d.triangle_count)
            AOB_CH(hipMemcpyAsync(d.indices, src.indices, 12ull * d.triangle_count, hipMemcpyHostToDevice, b->stream));
    }
    AOB_CH(hipMemcpyAsync(b->hidden_from_others, bits.data(), sizeof(uint32_t) * bits.size(), hipMemcpyHostToDevice, b->stream));
    for (auto& d : b->materials)
        AOB_CH(hipMemsetAsync(d.results, 0, sizeof(AobResult) * (uint64_t)d.width * d.height, b->stream));
    if (b->acc_cells) AOB_CH(hipMemsetAsync(b->acc_cells, 0, 16ull * b->acc_cells_count, b->stream));      // (kept zero between launches)
    AOB_CH(hipMemsetAsync(b->horizon_counters, 0, sizeof(unsigned long long) * (16 + b->materials.size()), b->stream));
    std::shared_ptr<const std::vector<float>> table;                // (held until the stream has drained, below)
    if (b->hemisphere_entries) {
        table = hemisphere_table(b->hemisphere_entries);
        AOB_CH(hipMemcpyAsync(b->hemisphere, table->data(), sizeof(float) * 3 * b->hemisphere_entries, hipMemcpyHostToDevice, b->stream));
    }
    AOB_CH(hipEventRecord(b->ev[1], b->stream));
    AOB_CH(hipStreamSynchronize(b->stream));
    float ms = 0.f;
    hipEventElapsedTime(&ms, b->ev[0], b->ev[1]);
    b->stats.ms_upload = ms;
    b->stats.triangle_count = n;
#undef AOB_C
#undef AOB_CH
    *out = b;
    return AOB_OK;
}

void aob_bake_destroy(AobBake* bake)
{
    if (!bake) return;
    DeviceScope scope;
    if (hipGetDevice(&scope.prev) == hipSuccess && scope.prev != bake->device) { hipSetDevice(bake->device); scope.changed = true; }
    if (bake->stream) hipStreamSynchronize(bake->stream);
    free_bake(bake);
}

int aob_bake_build_bvh(AobBake* b)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, 0, false));
    AOB_HIP(hipEventRecord(b->ev[2], b->stream));
    AOB_TRY(build_bvh(b));
    b->depth_known = false;
    if (b->node_count > 1)
        AOB_HIP(hipMemcpyAsync(&b->pinned[1], b->depth_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
    AOB_HIP(hipEventRecord(b->ev[3], b->stream));
    b->bvh_timed = true;
    return AOB_OK;
}

int aob_bake_rasterize(AobBake* b, int32_t material)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, material, true));
    AOB_HIP(hipEventRecord(b->ev[4], b->stream));
    const int rc = rasterize_material(b, material, 0, b->materials[material].height);
    AOB_HIP(hipEventRecord(b->ev[5], b->stream));
    b->raster_timed = rc == AOB_OK;
    return rc;
}

int aob_bake_rasterize_rows(AobBake* b, int32_t material, int32_t row_begin, int32_t row_end)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, material, true));
    if (row_begin < 0 || row_end < row_begin || row_end > b->materials[material].height) {
        set_error("rows [%d, %d) outside the %d-row atlas", row_begin, row_end, b->materials[material].height);
        return AOB_ERR_INVALID_ARGUMENT;
    }
    AOB_HIP(hipEventRecord(b->ev[4], b->stream));
    const int rc = rasterize_material(b, material, row_begin, row_end);
    AOB_HIP(hipEventRecord(b->ev[5], b->stream));
    b->raster_timed = rc == AOB_OK;
    return rc;
}

int aob_bake_trace(AobBake* b, int32_t material, uint64_t begin, uint64_t end)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, material, true));
    if (!b->bvh_built) { set_error("aob_bake_build_bvh has not run"); return AOB_ERR_INVALID_ARGUMENT; }
    if (!b->materials[material].rasterized) { set_error("material %d has not been rasterised", material); return AOB_ERR_INVALID_ARGUMENT; }
    AOB_TRY(ensure_depth(b));
    return trace_range(b, material, begin, end);
}

int aob_bake_resolve(AobBake* b, int32_t material)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, material, true));
    return resolve_material(b, material);
}

int aob_bake_clear_results(AobBake* b, int32_t material)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, material, true));
    DeviceMaterial& m = b->materials[material];
    AOB_HIP(hipMemsetAsync(m.results, 0, sizeof(AobResult) * (uint64_t)m.width * m.height, b->stream));
    return AOB_OK;
}

int aob_bake_add_results(AobBake* b, int32_t material, const void* other_cells, uint64_t cell_count)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, material, true));
    DeviceMaterial& m = b->materials[material];
    const uint64_t cells = (uint64_t)m.width * m.height;
    if (!other_cells || cell_count != cells) { set_error("aob_bake_add_results: %llu cells given, the material has %llu", (unsigned long long)cell_count, (unsigned long long)cells); return AOB_ERR_INVALID_ARGUMENT; }
    const uint64_t words = 2 * cells;
    const dim3 grid((uint32_t)((words + 255) / 256));
    if (b->opt.accumulate == AOB_ACCUMULATE_QUANTISED) hipLaunchKernelGGL(k_add_results<false>, grid, dim3(256), 0, b->stream, (uint32_t*)m.results, (const uint32_t*)other_cells, words);
    else hipLaunchKernelGGL(k_add_results<true>, grid, dim3(256), 0, b->stream, (uint32_t*)m.results, (const uint32_t*)other_cells, words);
    AOB_HIP(hipGetLastError());
    return AOB_OK;
}

int aob_bake_add_results_from(AobBake* b, int32_t material, AobBake* other, int32_t other_material)
{
    DeviceScope scope;
    AOB_TRY(check_bake(scope, other, other_material, true));
    AOB_TRY(check_bake(scope, b, material, true));
    if (other->device != b->device) { set_error("aob_bake_add_results_from: the handles live on devices %d and %d (between devices: an RCCL sum-reduce of the result buffers)", b->device, other->device); return AOB_ERR_INVALID_ARGUMENT; }
    const DeviceMaterial& om = other->materials[other_material];
    if (other != b) {
        // everything `other` has enqueued so far — its trace — comes first
        AOB_HIP(hipEventRecord(other->ev_peer, other->stream));
        AOB_HIP(hipStreamWaitEvent(b->stream, other->ev_peer, 0));
    }
    return aob_bake_add_results(b, material, om.results, (uint64_t)om.width * om.height);
}

int aob_bake_download(AobBake* b, int32_t material, void* dst, size_t dst_bytes)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, material, true));
    DeviceMaterial& m = b->materials[material];
    const size_t bytes = (size_t)m.width * m.height * m.channels * m.bytes_per_channel;
    if (!dst) { dst = m.host_data; dst_bytes = bytes; }
    if (!dst) { set_error("material %d has no host buffer", material); return AOB_ERR_INVALID_ARGUMENT; }
    if (dst_bytes < bytes) { set_error("destination too small (%zu < %zu)", dst_bytes, bytes); return AOB_ERR_INVALID_ARGUMENT; }
    // BakeEngine.cpp:705-709: tightly packed W*H*channels*bytes, no row pitch
    AOB_HIP(hipMemcpyAsync(dst, m.image, bytes, hipMemcpyDeviceToHost, b->stream));
    AOB_HIP(hipStreamSynchronize(b->stream));
    return AOB_OK;
}

int aob_bake_run_device(AobBake* b)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, 0, false));
    hipStream_t s = b->stream;
    AOB_TRY(aob_bake_build_bvh(b));
    float ms_r = 0.f, ms_t = 0.f, ms_p = 0.f;
    for (size_t i = 0; i < b->materials.size(); i++) {
        AOB_HIP(hipEventRecord(b->ev[4], s));
        AOB_TRY(rasterize_material(b, (int32_t)i, 0, b->materials[i].height));
        AOB_HIP(hipEventRecord(b->ev[5], s));
        AOB_TRY(ensure_depth(b));
        AOB_TRY(trace_range(b, (int32_t)i, 0, UINT64_MAX));
        AOB_HIP(hipEventRecord(b->ev[6], s));
        AOB_TRY(resolve_material(b, (int32_t)i));
        AOB_HIP(hipEventRecord(b->ev[7], s));
        AOB_HIP(hipStreamSynchronize(s));
        float a = 0.f, c = 0.f, d = 0.f;
        hipEventElapsedTime(&a, b->ev[4], b->ev[5]);
        hipEventElapsedTime(&c, b->ev_trace[0], b->ev_trace[1]);
        hipEventElapsedTime(&d, b->ev[6], b->ev[7]);
        ms_r += a; ms_t += c; ms_p += d;
    }
    AOB_HIP(hipStreamSynchronize(s));
    b->stats.ms_rasterize = ms_r; b->stats.ms_trace = ms_t; b->stats.ms_resolve = ms_p;
    return AOB_OK;
}

int aob_bake_sync(AobBake* b)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, 0, false));
    AOB_HIP(hipStreamSynchronize(b->stream));
    return AOB_OK;
}

int aob_bake_get_stats(AobBake* b, AobStats* stats)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, 0, false));
    if (!stats) { set_error("stats is NULL"); return AOB_ERR_INVALID_ARGUMENT; }
    AOB_TRY(ensure_depth(b));
    if (b->bvh_timed) { float ms = 0.f; hipEventElapsedTime(&ms, b->ev[2], b->ev[3]); b->stats.ms_bvh_build = ms; }
    b->stats.triangle_count = b->triangle_count;
    b->stats.node_count = b->node_count;
    b->stats.bvh_depth = b->bvh_depth;
    uint64_t surfels = 0;
    for (auto& m : b->materials) surfels += m.surfel_count;
    b->stats.surfel_count = surfels;
    b->stats.ray_count = surfels * (uint64_t)b->opt.sample_count;
    b->stats.ms_total_device = b->stats.ms_bvh_build + b->stats.ms_rasterize + b->stats.ms_trace + b->stats.ms_resolve;
    // tile cuts / horizon tables: what the materials' launches reported, the worst of them (2: wanted and could not be had)
    b->stats.trace_cuts = 0u; b->stats.trace_horizon = 0u;
    for (auto& m : b->materials) { b->stats.trace_cuts = std::max(b->stats.trace_cuts, m.cuts_state); b->stats.trace_horizon = std::max(b->stats.trace_horizon, m.horizon_state); }
    // the samples the trace launches cast: all of them, less what the horizon tables resolved (launches since each material's last rasterisation)
    b->stats.rays_cast = b->stats.ray_count;
    {
        uint64_t asked = 0;
        for (auto& m : b->materials) asked += m.horizon_asked;
        if (asked) {
            std::vector<unsigned long long> c(16 + b->materials.size());
            AOB_HIP(hipMemcpyAsync(c.data(), b->horizon_counters, sizeof(unsigned long long) * c.size(), hipMemcpyDeviceToHost, b->stream));
            AOB_HIP(hipStreamSynchronize(b->stream));
            uint64_t cast = 0;
            for (size_t i = 0; i < b->materials.size(); i++) cast += c[16 + i];
            b->stats.rays_cast = b->stats.ray_count > asked ? b->stats.ray_count - asked + cast : cast;
        }
    }
    *stats = b->stats;
    return AOB_OK;
}

int aob_bake_node_count(AobBake* b, uint64_t* count)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, 0, false));
    if (!count) { set_error("count is NULL"); return AOB_ERR_INVALID_ARGUMENT; }
    *count = b->node_count;
    return AOB_OK;
}

int aob_bake_copy_nodes(AobBake* b, AobBvhNode* dst, uint64_t capacity)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, 0, false));
    if (capacity < b->node_count || (!dst && b->node_count)) { set_error("node buffer too small"); return AOB_ERR_INVALID_ARGUMENT; }
    AOB_TRY(ensure_nodes(b));
    if (b->node_count) AOB_HIP(hipMemcpyAsync(dst, b->nodes, sizeof(AobBvhNode) * b->node_count, hipMemcpyDeviceToHost, b->stream));
    AOB_HIP(hipStreamSynchronize(b->stream));
    return AOB_OK;
}

int aob_bake_node4_count(AobBake* b, uint64_t* count)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, 0, false));
    if (!count) { set_error("count is NULL"); return AOB_ERR_INVALID_ARGUMENT; }
    *count = b->node4_count;
    return AOB_OK;
}

int aob_bake_copy_nodes4(AobBake* b, AobBvh4Node* dst, uint64_t capacity)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, 0, false));
    if (capacity < b->node4_count || (!dst && b->node4_count)) { set_error("node4 buffer too small"); return AOB_ERR_INVALID_ARGUMENT; }
    if (b->node4_count) AOB_HIP(hipMemcpyAsync(dst, b->nodes4, sizeof(AobBvh4Node) * b->node4_count, hipMemcpyDeviceToHost, b->stream));
    AOB_HIP(hipStreamSynchronize(b->stream));
    return AOB_OK;
}

int aob_bake_copy_leaf_tris(AobBake* b, AobLeafTri* dst, uint64_t capacity)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, 0, false));
    const uint64_t n = b->bvh_built ? b->triangle_count : 0;
    if (capacity < n || (!dst && n)) { set_error("leaf triangle buffer too small"); return AOB_ERR_INVALID_ARGUMENT; }
    if (n) AOB_HIP(hipMemcpyAsync(dst, b->leaf_tris, sizeof(AobLeafTri) * n, hipMemcpyDeviceToHost, b->stream));
    AOB_HIP(hipStreamSynchronize(b->stream));
    return AOB_OK;
}

int aob_bake_copy_nodes2q(AobBake* b, AobBvh2QNode* dst, uint64_t capacity)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, 0, false));
    const uint64_t n = b->bvh_built && b->triangle_count ? b->triangle_count - 1 : 0;
    if (capacity < n || (!dst && n)) { set_error("quantised node buffer too small"); return AOB_ERR_INVALID_ARGUMENT; }
    AOB_TRY(ensure_nodes2q(b));
    if (n) AOB_HIP(hipMemcpyAsync(dst, b->nodes2q, sizeof(AobBvh2QNode) * n, hipMemcpyDeviceToHost, b->stream));
    AOB_HIP(hipStreamSynchronize(b->stream));
    return AOB_OK;
}

static int node4q_count(AobBake* b, uint64_t* count)
{
    if (!b->bvh_built || b->triangle_count == 0) { *count = 0; return AOB_OK; }
    if (b->node4q_count == UINT64_MAX) {
        uint32_t c = 0;
        AOB_HIP(hipMemcpyAsync(&c, b->node4q_count_dev, sizeof c, hipMemcpyDeviceToHost, b->stream));
        AOB_HIP(hipStreamSynchronize(b->stream));
        b->node4q_count = c;
    }
    *count = b->node4q_count;
    return AOB_OK;
}

int aob_bake_node4q_count(AobBake* b, uint64_t* count)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, 0, false));
    if (!count) { set_error("count is null"); return AOB_ERR_INVALID_ARGUMENT; }
    return node4q_count(b, count);
}

int aob_bake_copy_nodes4q(AobBake* b, AobBvh4QNode* dst, uint64_t capacity)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, 0, false));
    uint64_t n = 0;
    AOB_TRY(node4q_count(b, &n));
    if (capacity < n || (!dst && n)) { set_error("node4q buffer too small"); return AOB_ERR_INVALID_ARGUMENT; }
    if (n) AOB_HIP(hipMemcpyAsync(dst, b->nodes4q, sizeof(AobBvh4QNode) * n, hipMemcpyDeviceToHost, b->stream));
    AOB_HIP(hipStreamSynchronize(b->stream));
    return AOB_OK;
}

int aob_bake_horizon_counters(AobBake* b, unsigned long long* out16, int clear)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, 0, false));
    if (!out16) { set_error("out is NULL"); return AOB_ERR_INVALID_ARGUMENT; }
    AOB_HIP(hipMemcpyAsync(out16, b->horizon_counters, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost, b->stream));
    if (clear) AOB_HIP(hipMemsetAsync(b->horizon_counters, 0, sizeof(unsigned long long) * 16, b->stream));
    if (clear == 2) b->horizon_diag = true;               // from here on k_horizon counts (it does not by default: the atomics cost more than the kernel)
    AOB_HIP(hipStreamSynchronize(b->stream));
    return AOB_OK;
}

int aob_bake_copy_horizon(AobBake* b, int32_t material, uint8_t* dst, uint64_t capacity_bytes, uint32_t* stride, uint64_t* surfels)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, material, true));
    DeviceMaterial& m = b->materials[material];
    if (!stride || !surfels) { set_error("stride / surfels is NULL"); return AOB_ERR_INVALID_ARGUMENT; }
    *stride = 0; *surfels = 0;
    if (!m.horizon || m.horizon_state != 1u || m.surfel_count == 0) return AOB_OK;
    const uint32_t st = horizon_stride(b);
    *stride = st; *surfels = m.surfel_count;
    const uint64_t bytes = (uint64_t)st * m.surfel_count;
    if (!dst || capacity_bytes < bytes) { set_error("horizon buffer too small (%llu bytes needed)", (unsigned long long)bytes); return AOB_ERR_INVALID_ARGUMENT; }
    // the tables are kept in trace order; hand them out in the order of aob_bake_copy_surfels
    std::vector<uint8_t> hz(bytes);
    std::vector<uint32_t> perm(m.surfel_count);
    AOB_HIP(hipMemcpyAsync(hz.data(), m.horizon, bytes, hipMemcpyDeviceToHost, b->stream));
    AOB_HIP(hipMemcpyAsync(perm.data(), m.perm, sizeof(uint32_t) * m.surfel_count, hipMemcpyDeviceToHost, b->stream));
    AOB_HIP(hipStreamSynchronize(b->stream));
    for (size_t si = 0; si + 1 < m.section_offsets.size(); si++)
        for (uint64_t j = m.section_offsets[si]; j < m.section_offsets[si + 1]; j++) {
            if (perm[j] >= m.section_offsets[si + 1] - m.section_offsets[si]) { set_error("the trace order of material %d is not a permutation", material); return AOB_ERR_DEVICE; }
            memcpy(dst + (m.section_offsets[si] + perm[j]) * st, hz.data() + j * st, st);
        }
    return AOB_OK;
}

int aob_bake_copy_tile_cuts(AobBake* b, int32_t material, uint32_t* dst, uint64_t capacity_words, uint32_t* tiles, uint32_t* records_per_tile)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, material, true));
    DeviceMaterial& m = b->materials[material];
    if (!tiles || !records_per_tile) { set_error("tiles / records_per_tile is NULL"); return AOB_ERR_INVALID_ARGUMENT; }
    *tiles = m.tile_cut_tiles; *records_per_tile = m.tile_cut_records;
    const uint64_t words = (uint64_t)m.tile_cut_tiles * m.tile_cut_records * 128u;
    if (words == 0) return AOB_OK;
    if (!dst || capacity_words < words) { set_error("tile cut buffer too small (%llu words needed)", (unsigned long long)words); return AOB_ERR_INVALID_ARGUMENT; }
    AOB_HIP(hipMemcpyAsync(dst, m.tile_cuts, sizeof(uint32_t) * words, hipMemcpyDeviceToHost, b->stream));
    AOB_HIP(hipStreamSynchronize(b->stream));
    return AOB_OK;
}

int aob_bake_quant_frame(AobBake* b, AobQuantFrame* frame)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, 0, false));
    if (!frame) { set_error("frame is NULL"); return AOB_ERR_INVALID_ARGUMENT; }
    AOB_HIP(hipMemcpyAsync(frame, b->quant_frame, sizeof(AobQuantFrame), hipMemcpyDeviceToHost, b->stream));
    AOB_HIP(hipStreamSynchronize(b->stream));
    return AOB_OK;
}

int aob_bake_surfel_count(AobBake* b, int32_t material, uint64_t* count)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, material, true));
    if (!count) { set_error("count is NULL"); return AOB_ERR_INVALID_ARGUMENT; }
    *count = b->materials[material].surfel_count;
    return AOB_OK;
}

int aob_bake_copy_surfels(AobBake* b, int32_t material, AobSurfel* dst, uint64_t capacity)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, material, true));
    DeviceMaterial& m = b->materials[material];
    if (capacity < m.surfel_count || (!dst && m.surfel_count)) { set_error("surfel buffer too small"); return AOB_ERR_INVALID_ARGUMENT; }
    if (m.surfel_count) AOB_HIP(hipMemcpyAsync(dst, m.surfels, sizeof(AobSurfel) * m.surfel_count, hipMemcpyDeviceToHost, b->stream));
    AOB_HIP(hipStreamSynchronize(b->stream));
    return AOB_OK;
}

int aob_bake_copy_results(AobBake* b, int32_t material, AobResult* dst, uint64_t capacity)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, material, true));
    DeviceMaterial& m = b->materials[material];
    const uint64_t cells = (uint64_t)m.width * m.height;
    if (capacity < cells || !dst) { set_error("result buffer too small"); return AOB_ERR_INVALID_ARGUMENT; }
    AOB_HIP(hipMemcpyAsync(dst, m.results, sizeof(AobResult) * cells, hipMemcpyDeviceToHost, b->stream));
    AOB_HIP(hipStreamSynchronize(b->stream));
    return AOB_OK;
}

int aob_bake_shard_bounds(AobBake* b, int32_t material, uint32_t parts, uint64_t* bounds,
                          uint64_t* first_cell, uint64_t* last_cell, int32_t* overlapping)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, material, true));
    DeviceMaterial& m = b->materials[material];
    if (!parts || !bounds) { set_error("parts is 0 or bounds is NULL"); return AOB_ERR_INVALID_ARGUMENT; }
    if (!m.rasterized) { set_error("material %d has not been rasterised", material); return AOB_ERR_INVALID_ARGUMENT; }
    AOB_HIP(hipStreamSynchronize(b->stream));
    const uint64_t n = m.surfel_count, cells = (uint64_t)m.width * m.height;
    const bool single = m.section_offsets.size() == 2 && m.tile_prefix_valid;
    if (overlapping) *overlapping = (single || n == 0) ? 0 : 1;
    if (n == 0) {
        // an empty strip (or a material nothing maps to): every part is empty, no cell range
        for (uint32_t r = 0; r <= parts; r++) bounds[r] = 0;
        for (uint32_t r = 0; r < parts; r++) { if (first_cell) first_cell[r] = 1; if (last_cell) last_cell[r] = 0; }
        return AOB_OK;
    }
    // tile rows of the strip the last rasterisation covered (the whole atlas by default)
    const uint32_t tiles_y = (uint32_t)(m.window_rows + 7) / 8;
    const uint64_t row0 = (uint64_t)m.window_row0, row_end = (uint64_t)m.window_row0 + (uint64_t)m.window_rows;
    std::vector<uint32_t> row_of(parts + 1, 0);
    bounds[0] = 0; bounds[parts] = n; row_of[parts] = tiles_y;
    for (uint32_t r = 1; r < parts; r++) {
        const uint64_t want = n * r / parts;
        if (!single) { bounds[r] = want; continue; }
        // nearest tile-row boundary (prefix is non-decreasing), never before the previous bound
        uint32_t lo = row_of[r - 1], best = lo;
        for (uint32_t ty = lo; ty <= tiles_y; ty++) {
            const uint64_t pv = m.tile_row_prefix_host[ty];
            const uint64_t pb = m.tile_row_prefix_host[best];
            if ((pv > want ? pv - want : want - pv) <= (pb > want ? pb - want : want - pb)) best = ty;
            if (pv > want) break;
        }
        row_of[r] = best; bounds[r] = m.tile_row_prefix_host[best];
    }
    for (uint32_t r = 0; r < parts; r++) {
        uint64_t c0 = 1, c1 = 0;                      // empty: first > last
        if (bounds[r + 1] > bounds[r]) {
            if (single) { c0 = (row0 + (uint64_t)row_of[r] * 8u) * m.width; c1 = std::min<uint64_t>(row0 + (uint64_t)row_of[r + 1] * 8u, row_end) * m.width - 1; }
            else { c0 = 0; c1 = cells - 1; }
        }
        if (first_cell) first_cell[r] = c0;
        if (last_cell) last_cell[r] = c1;
    }
    return AOB_OK;
}

int aob_bake_results_device_ptr(AobBake* b, int32_t material, void** ptr, uint64_t* bytes)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, material, true));
    DeviceMaterial& m = b->materials[material];
    if (ptr) *ptr = m.results;
    if (bytes) *bytes = sizeof(AobResult) * (uint64_t)m.width * m.height;
    return AOB_OK;
}

int aob_bake_image_device_ptr(AobBake* b, int32_t material, void** ptr, uint64_t* bytes)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, material, true));
    DeviceMaterial& m = b->materials[material];
    if (ptr) *ptr = m.image;
    if (bytes) *bytes = (uint64_t)m.width * m.height * m.channels * m.bytes_per_channel;
    return AOB_OK;
}

int aob_bake_stream(AobBake* b, void** stream)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, 0, false));
    if (stream) *stream = (void*)b->stream;
    return AOB_OK;
}

int aob_bake_last_trace_ms(AobBake* b, float* ms)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, 0, false));
    if (!ms) { set_error("ms is NULL"); return AOB_ERR_INVALID_ARGUMENT; }
    if (!b->trace_timed) { set_error("no trace has been launched"); return AOB_ERR_INVALID_ARGUMENT; }
    AOB_HIP(hipEventSynchronize(b->ev_trace[1]));
    AOB_HIP(hipEventElapsedTime(ms, b->ev_trace[0], b->ev_trace[1]));
    return AOB_OK;
}

int aob_bake_last_stage_ms(AobBake* b, float* out6)
{
    DeviceScope scope; AOB_TRY(check_bake(scope, b, 0, false));
    if (!out6) { set_error("out is NULL"); return AOB_ERR_INVALID_ARGUMENT; }
    AOB_HIP(hipStreamSynchronize(b->stream));
    for (int i = 0; i < 6; i++) out6[i] = -1.0f;
    if (b->bvh_timed) hipEventElapsedTime(&out6[0], b->ev[2], b->ev[3]);
    if (b->raster_timed) hipEventElapsedTime(&out6[1], b->ev[4], b->ev[5]);
    if (b->trace_timed) hipEventElapsedTime(&out6[2], b->ev_trace[0], b->ev_trace[1]);
    if (b->parts_timed) {
        hipEventElapsedTime(&out6[3], b->ev_part[0], b->ev_part[1]);
        hipEventElapsedTime(&out6[4], b->ev_part[1], b->ev_part[2]);
        hipEventElapsedTime(&out6[5], b->ev_part[2], b->ev_part[3]);
    }
    (void)hipGetLastError();
    return AOB_OK;
}

// BakeEngine::Bake (BakeEngine.cpp:335-751) end to end
int aob_bake(const AobSceneDesc* scene, const AobOptions* options, AobStats* stats)
{
    AobBake* b = nullptr;
    AOB_TRY(aob_bake_create(scene, options, &b));
    int rc = aob_bake_run_device(b);
    if (rc == AOB_OK) {
        hipEventRecord(b->ev[0], b->stream);
        for (size_t i = 0; i < b->materials.size() && rc == AOB_OK; i++)
            if (b->materials[i].host_data) rc = aob_bake_download(b, (int32_t)i, nullptr, 0);
        hipEventRecord(b->ev[1], b->stream);
        hipStreamSynchronize(b->stream);
        float ms = 0.f; hipEventElapsedTime(&ms, b->ev[0], b->ev[1]);
        b->stats.ms_download = ms;
    }
    if (rc == AOB_OK && stats) rc = aob_bake_get_stats(b, stats);
    aob_bake_destroy(b);
    return rc;
}

} // extern "C"

// aob_internal.h — internal state of the HIP bake library (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/aobaker.h"

namespace aob {

void set_error(const char* fmt, ...);

#define AOB_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            aob::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__,  \
                           __LINE__);                                                         \
            return e__ == hipErrorOutOfMemory ? AOB_ERR_OUT_OF_MEMORY : AOB_ERR_DEVICE;       \
        }                                                                                     \
    } while (0)

#define AOB_TRY(expr)                       \
    do {                                    \
        int rc__ = (expr);                  \
        if (rc__ != AOB_OK) return rc__;    \
    } while (0)

struct Box { float pmin[3], pmax[3]; };

struct DeviceMesh {
    float*    positions = nullptr;   // packed float3
    float*    normals = nullptr;     // packed float3
    float*    uvs = nullptr;         // packed float2
    uint32_t* indices = nullptr;
    uint32_t  vertex_count = 0, triangle_count = 0;
    uint64_t  triangle_offset = 0;   // first global triangle
    int32_t   material = -1;
    float     model[16];
    uint32_t  flags = 0;             // AOB_MESH_*
};

struct DeviceMaterial {
    int32_t width = 0, height = 0, channels = 0, bytes_per_channel = 0;
    float ao_max_range = 0.0f;            // AobMaterialDesc::ao_max_range
    int32_t window_rows = 0;              // texel rows the last rasterisation covered (height, or a strip of it)
    int32_t window_row0 = 0;              // ... and the first of them
    uint8_t* host_data = nullptr;
    uint32_t* cover = nullptr;            // W*H: winning triangle index + 1, per section pass
    uint32_t* prev_layer = nullptr;       // W*H, AOB_LAYERS_ALL only: triangle index + 1 of the texel's fragment in the layer before
    uint32_t* block_sums = nullptr;       // scan partials (+ the total, + the overlap flag)
    uint8_t*  seen = nullptr;             // W*H: 1 where a section rasterised so far put a surfel (overlap detection)
    bool      sections_overlap = false;   // some texel has surfels of two sections (or two layers): their launches stay apart
    uint32_t  scan_blocks = 0;
    AobSurfel* surfels = nullptr;         // concatenated, row-major per section
    uint32_t* section_index = nullptr;    // surfel -> index inside its section's list
    float4*   frames = nullptr;           // 64 B per surfel IN TRACE (8x8 tile) ORDER: P, N, tangent X/Z, cell, table base, valid
    uint32_t* perm = nullptr;             // trace order -> row-major rank (per section)
    uint32_t* tile_cuts = nullptr;        // 128 words per 64-surfel tile and sample part of a trace launch: the tile's cut of the tree (trace.hip k_tile_cuts)
    uint64_t  tile_cut_capacity = 0;      // ... in words
    uint64_t  tile_cut_failed = 0;        // ... the smallest request that could not be allocated since the last rasterisation (0: none): not tried again
    uint32_t  cuts_state = 0, horizon_state = 0;   // this material's launches since its last rasterisation: AobStats::trace_cuts / trace_horizon (the stats carry the maximum over the materials)
    uint32_t  tile_cut_tiles = 0, tile_cut_records = 0;   // ... of the most recent launch (0 tiles: it walked from the root)
    uint8_t*  horizon = nullptr;          // horizon tables of the surfels (horizon.hip): per surfel, in trace order, the free columns of every azimuth row of the COSINE sampler's grid
    uint64_t  horizon_capacity = 0;       // ... in bytes
    uint32_t  horizon_verdict = 0;        // trace_horizon 0 on a large launch: 0 not sampled yet, 1 the sample said worth it, 2 not worth it (kept for the handle)
    float     horizon_sampled = 0.0f;     // ... samples per surfel the sample's tables resolved
    bool      horizon_declined = false;   // the last horizon_build returned false because of that verdict (not for lack of memory)
    uint64_t  horizon_asked = 0;          // samples of the launches that had tables, since the last rasterisation (AobStats::rays_cast)
    uint32_t* tile_sums = nullptr;        // scan partials of the tile-order pass
    uint32_t* tile_row_prefix = nullptr;  // covered texels before each 8-texel tile row (first section)
    uint32_t* tile_row_prefix_host = nullptr;   // pinned copy
    bool      tile_prefix_valid = false;        // ... filled by the last rasterisation (it is not when the first section is empty)
    uint64_t  surfel_capacity = 0, surfel_count = 0;
    std::vector<uint64_t> section_offsets; // prefix offsets per contributing mesh (+ total)
    std::vector<uint32_t> section_mesh;
    AobResult* results = nullptr;         // W*H
    uint8_t* image = nullptr;             // W*H*channels*bpc
    bool rasterized = false;
};

} // namespace aob

struct AobBake {
    AobOptions opt;
    int device = 0;                     // the HIP device every buffer of the handle lives on; entry points select it (api.hip check_bake)
    hipStream_t stream = nullptr;
    uint8_t*  arena = nullptr;          // ONE device allocation: every buffer sized at creation is a slice of it (api.hip)
    size_t    arena_bytes = 0;
    std::vector<aob::DeviceMesh> meshes;
    std::vector<aob::DeviceMaterial> materials;
    uint64_t triangle_count = 0;

    // BVH build workspace and result
    float4*   tri_verts = nullptr;      // 3 x float4 per triangle: (v0,shape) (v1,prim) (v2,0)
    aob::Box* prim_bounds = nullptr;
    uint32_t* scene_bounds_enc = nullptr; // 6 ordered-uint encoded floats
    uint64_t* codes[2] = { nullptr, nullptr };
    uint32_t* prim_idx[2] = { nullptr, nullptr };
    void*     sort_temp = nullptr; size_t sort_temp_bytes = 0;
    uint32_t* child_left = nullptr; uint32_t* child_right = nullptr; uint32_t* parent = nullptr;
    aob::Box* node_bounds = nullptr;
    uint32_t* refit_flags = nullptr;
    uint32_t* depth_dev = nullptr;
    uint32_t* node_height = nullptr;      // PLOC: height of every inner node (the root's is the tree depth)
    uint2*    ploc_sums = nullptr;      // PLOC builder: scan partials (alive, merged)
    uint32_t* ploc_totals = nullptr;
    uint32_t* sah_counts = nullptr;       // bvh_builder 2: ranges per level, short ranges, the root's cut (bvh_build.hip build_sah_top)
    uint8_t*  sah_lists = nullptr;        // ... and its two cluster lists, ids + boxes (2 x 28 B per cluster)
    uint64_t  sah_list_capacity = 0;
    uint8_t*  sah_big = nullptr;          // ... and its long ranges: tables, the slices' bins, the cuts
    AobBvhNode* nodes = nullptr;
    float4*   nodes4 = nullptr;         // 8 x float4 per 4-ary node (AobBvh4Node), indexed by BVH2 inner index
    float4*   leaf_tris = nullptr;      // 3 x float4 per triangle in Morton order (AobLeafTri)
    uint64_t  node4_count = 0;
    uint4*    nodes4q = nullptr;        // 4 x uint4 per quantised 4-ary node (AobBvh4QNode), packed
    uint32_t* packed_index = nullptr;   // BVH2 inner-node index -> index in nodes4q
    uint32_t* node4q_count_dev = nullptr;
    uint64_t  node4q_count = 0;         // UINT64_MAX: not read back yet
    uint4*    nodes2q = nullptr;        // 2 x uint4 per quantised BVH2 node (AobBvh2QNode)
    float*    quant_frame = nullptr;    // origin[3], scale[3] (device)
    uint64_t node_count = 0;
    uint32_t bvh_depth = 0;
    bool bvh_built = false, depth_known = false, bvh_timed = false;
    bool nodes_emitted = false, nodes2q_emitted = false;   // `nodes` / `nodes2q` are written when something first wants them (bvh_build.hip)
    uint32_t* pinned = nullptr;         // 64 B of pinned host memory for small read-backs

    uint32_t* gstack = nullptr;         // traversal stacks in HBM, only for trees deeper than 62 levels
    uint32_t* trace_unit_next = nullptr; // [8] per-XCD work queue heads of the trace launch
    uint32_t* hidden_from_others = nullptr;  // bit s: shape s (mesh s-1) is hidden from the rays of other meshes
    bool      any_mesh_flags = false;
    int cu_count = 256;
    uint64_t gstack_words = 0;
    bool horizon_diag = false;          // k_horizon adds its diagnostics to horizon_counters (off by default: the atomics are slow)
    unsigned long long* horizon_counters = nullptr;   // k_horizon's diagnostics [0..8] (rounds, box tests, near triangles, middle leaves, tiles given up, tiles, clock ticks of its three phases), then from [16] one word per material: the samples k_trace cast under horizon tables
    unsigned long long* acc_cells = nullptr;   // 2 x 64-bit exact sums per result cell of the largest material: where the sample parts of a tile
    uint64_t acc_cells_count = 0;              // add up when the cells are floats that hold fractions (trace.hip; sized at creation, kept zero)

    float* hemisphere = nullptr;        // compat sampler table (float3 packed)
    uint64_t hemisphere_entries = 0;

    hipEvent_t ev[8] = {};
    hipEvent_t ev_trace[2] = {};
    hipEvent_t ev_part[4] = {};         // the most recent trace launch: before its tile cuts, after them, after its horizon tables, after its walk (aob_bake_last_stage_ms)
    bool parts_timed = false, raster_timed = false;
    hipEvent_t ev_peer = nullptr;       // aob_bake_add_results_from: "everything enqueued on this handle's stream so far", for another handle to wait on
    hipEvent_t ev_ploc[2] = {};         // PLOC rounds: the list length of round r is read back while round r + 1 runs
    bool trace_timed = false;
    AobStats stats = {};
};

namespace aob {
// The large device buffers of a handle — its arena, a material's tile cuts and horizon tables — come from and go back to a small per-device pool
// (api.hip): a process that bakes one scene after another (BakeEngine::Bake from the UI, the one-shot aob_bake) pays their hipMalloc / hipFree — a
// device synchronisation each, 4 ms per bake of BASELINE configs[2] — once.  What the reference's Init does with its ~3 GB of fixed buffers
// (BakeEngine.cpp:163-212), sized by the scenes actually baked; aob_shutdown of the last thread releases it.
hipError_t pool_alloc(void** p, size_t bytes, int device);
void pool_free(void* p, size_t bytes, int device);
void pool_clear();
// hipFree for a buffer that may be a slice of the bake's arena (then it is not freed on its own)
inline void release_device(const AobBake* b, void* p)
{
    if (!p) return;
    const uint8_t* q = (const uint8_t*)p;
    if (b->arena && q >= b->arena && q < b->arena + b->arena_bytes) return;
    hipFree(p);
}
// stage launchers (each enqueues on bake->stream; only the ones that say so synchronise)
int build_bvh(AobBake* b);                                   // bvh_build.hip
int ensure_nodes(AobBake* b);                                // the fp32 BVH2 node array (`nodes`), from the tree that is built
int ensure_nodes2q(AobBake* b);                              // the quantised BVH2 node array (`nodes2q`)
int rasterize_material(AobBake* b, int32_t material, int32_t row_begin, int32_t row_end);   // raster.hip (syncs once per section)
int trace_range(AobBake* b, int32_t material, uint64_t begin, uint64_t end);   // trace.hip
void reserve_tile_cuts(AobBake* b, int32_t material);        // trace.hip: the tile cuts of the material's launches, sized by what was rasterised
int reserve_trace_stack(AobBake* b);                         // trace.hip: the walk's HBM overflow stack, sized by the tree's depth
bool horizon_applies(const AobOptions& o);                   // horizon.hip: the options ask for horizon tables (COSINE sampler on a full grid, quantised 4-ary nodes, trace_horizon)
bool horizon_applies(const AobBake* b);
uint32_t horizon_stride(const AobBake* b);                   // ... bytes per surfel
bool horizon_build(AobBake* b, DeviceMaterial& m, uint64_t begin, uint64_t end, float tmax, bool verdict_only = false);   // ... the tables of a launch's surfels (false: no memory)
int resolve_material(AobBake* b, int32_t material);          // resolve.hip
uint64_t chunk_texel_max(const AobOptions& opt);             // trace.hip (BakeEngine.cpp:359-364)
}

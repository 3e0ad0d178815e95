/*
 * aob_oracle.h — CPU restatement of the reference AO bake path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under aobaker_amd/ (the product) may
 * include, link, load or call this.  Callers: tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg.
 *
 * PARITY: the reference (epicabsol/aobaker) has no tests, golden vectors or fixtures for this path, and its host
 * program (D3D11/HLSL + the stripped RadeonRays 2.0 library) cannot be built here.  What CAN be
 * built is the device code RadeonRays runs: oracle/ref_kernels.mk compiles kernels/CL/intersect_bvh2_lds.cl,
 * build_hlbvh.cl and intersect_bvh2level_skiplinks.cl from /root/reference for gfx950, and
 * tests/test_gpu_reference_kernels.py runs them on the GPU against this file — traverse(), intersect_triangle(),
 * intersect_bbox(), the Morton codes and the LBVH hierarchy + refit are PINNED that way, bit for bit, and so are the
 * ray records (orc_generate_rays) and the scoring (orc_process_intersections) through the cells they produce from the
 * reference kernel's hits.  The rasteriser and the resolve have no runnable reference: they stay pinned only by the
 * closed forms and known-answer tests of tests/test_oracle_kat.py, and by the MSVC rand() sequence for the sampler.
 *
 * Every function cites the reference file:line it restates.  The POD types
 * of the boundary (AobOptions, AobSceneDesc, AobSurfel, AobBvhNode,
 * AobResult) come from include/aobaker.h so the same buffers can be handed
 * to the oracle and to the HIP library.
 */
#ifndef AOB_ORACLE_H
#define AOB_ORACLE_H

#include "../include/aobaker.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar primitives (exported for known-answer tests) ---------------- */

/* MSVC CRT rand(): s = s*214013 + 2531011; return (s >> 16) & 0x7fff.
 * The reference never seeds it (BakeEngine.cpp:316-317), so s starts at 1. */
uint32_t orc_msvc_rand(uint32_t* state);
/* D3D ftoi: truncate toward zero, NaN -> 0, saturate
 * (the (int) casts at ProcessIntersectionsComputeShader.hlsl:87-88). */
int32_t orc_ftoi(float v);
uint16_t orc_f32_to_f16(float v, int mode /* AOB_FP16_RTZ | AOB_FP16_RNE */);
float orc_f16_to_f32(uint16_t h);
/* common.cl:177-218 with USE_SAFE_MATH (1.f/denom): returns t, or t_max on a miss. */
float orc_intersect_triangle(const float o[3], const float d[3],
                             const float v1[3], const float v2[3], const float v3[3],
                             float t_max);
/* common.cl:220-232 */
void orc_safe_invdir(const float d[3], float invdir[3]);
/* common.cl:236-246 / intersect_bvh2_lds.cl:54-63: out[0]=t0, out[1]=t1; hit iff t0<=t1 */
void orc_intersect_bbox(const float pmin[3], const float pmax[3], const float invdir[3],
                        const float oxinvdir[3], float t_max, float out[2]);
/* The COSINE sampler: local-frame direction (x,z tangent, y normal) of sample i
 * of the texel with linear index `texel`. */
void orc_cosine_sample(uint32_t seed, uint32_t texel, int32_t i, int32_t sample_count, float out[3]);
/* The polynomial sin/cos of 2*pi*u the COSINE sampler is defined with. */
void orc_sincos2pi(float u, float* s, float* c);

/* ---- stages ---------------------------------------------------------------- */

/* BakeEngine.cpp:312-333: fills `count` local-frame directions (x, 1-sin, z). */
void orc_hemisphere_table(float* out_xyz, uint64_t count);
/* Entries of the table a bake with these options can index:
 * min(ray_chunk_size / spp, 65535) * spp (BakeEngine.cpp:359-364). */
uint64_t orc_compat_table_entries(const AobOptions* opt);

/* World-space triangle soup of the whole scene in the order the reference
 * attaches shapes (BakeEngine.cpp:370-398): 9 floats + shape id + prim id. */
typedef struct OrcTriangle { float v[3][3]; uint32_t shape_id, prim_id; } OrcTriangle;
uint64_t orc_scene_triangle_count(const AobSceneDesc* scene);
void orc_scene_triangles(const AobSceneDesc* scene, OrcTriangle* out);

/* LBVH (build_hlbvh.cl:69-314, 63-bit codes) emitted in the 64-byte
 * traversal layout of intersect_bvh2_lds.cl:40-52.  `nodes` must hold
 * max(1, 2n-1) entries.  Returns the node count; *depth = max leaf depth
 * (root = depth 0). */
uint64_t orc_bvh_build(const OrcTriangle* tris, uint64_t n, AobBvhNode* nodes, uint32_t* depth);
/* builder 0: the LBVH above; 1: PLOC (AobOptions::bvh_builder), same layout and id conventions. */
uint64_t orc_bvh_build_ex(const OrcTriangle* tris, uint64_t n, AobBvhNode* nodes, uint32_t* depth, int builder, int adaptive);

/* The 4-ary collapse of that BVH2 (every second level removed, boxes unchanged) which the
 * default GPU kernel walks with four lanes per ray; orc_trace uses it unless
 * AobOptions::bvh_width == 2.  nodes4: max(1, n-1) entries, tris: n entries. */
uint64_t orc_bvh4_build(const AobBvhNode* nodes2, uint64_t node_count2, AobBvh4Node* nodes4, AobLeafTri* tris);
/* pairs: BVH2 nodes whose children are the leaves p and p + 1 become one child slot (AOB_PAIR_FLAG) — what the library
 * builds for bvh_width 4 with bvh_quantised 1 */
uint64_t orc_bvh4_build_ex(const AobBvhNode* nodes2, uint64_t node_count2, AobBvh4Node* nodes4, AobLeafTri* tris, int pairs);

/* The same BVH2 with 16-bit quantised child boxes (AobBvh2QNode, 32 B) that the lane-per-ray GPU
 * kernel walks when AobOptions::bvh_quantised is set (bvh_width 2).  out: max(1, n-1) entries. */
uint64_t orc_bvh2q_build(const AobBvhNode* nodes2, uint64_t node_count2, AobBvh2QNode* out, AobQuantFrame* frame);

/* The 4-ary nodes with the quantised boxes (AobBvh4QNode; bvh_width 4 + bvh_quantised). */
uint64_t orc_bvh4q_build(const AobBvh4Node* nodes4, uint64_t count4, const AobQuantFrame* frame, AobBvh4QNode* out);

/* UV rasterisation + texel extraction for one material
 * (Rasterize{Vertex,Pixel}Shader.hlsl, BakeEngine.cpp:130-145,531-588).
 * Writes up to `capacity` surfels (row-major per section, sections
 * concatenated in scene order) and, per surfel, its index within its own
 * section's list (the reference's chunk-local sample indexing needs it).
 * Returns the number of surfels the material has (may exceed capacity). */
uint64_t orc_rasterize(const AobSceneDesc* scene, int32_t material, const AobOptions* opt,
                       AobSurfel* surfels, uint32_t* section_index, uint64_t capacity);

/* ... and, per surfel, the index of the mesh (section) it was rasterised from. */
uint64_t orc_rasterize_ex(const AobSceneDesc* scene, int32_t material, const AobOptions* opt,
                          AobSurfel* surfels, uint32_t* section_index, uint32_t* surfel_mesh, uint64_t capacity);

typedef struct OrcTraceStats {
    uint64_t rays;
    uint64_t inner_visits;    /* inner-node fetches (64 B BVH2 nodes, or 128 B 4-ary nodes) */
    uint64_t leaf_visits;     /* leaf fetches = triangle tests (64 B BVH2 leaf, or 48 B AobLeafTri) */
    uint64_t hits;
} OrcTraceStats;

/* Ray generation + traversal + scoring + accumulation for surfels
 * [begin,end) (GenerateRaysComputeShader.hlsl:31-55, intersect_bvh2_lds.cl,
 * ProcessIntersectionsComputeShader.hlsl:52-92).  Accumulates (+=) into
 * `results` (width*height cells).  `hemisphere` is the compat table (NULL
 * for the COSINE sampler).  threads <= 1 runs inline. */
int orc_trace(const AobBvhNode* nodes, uint64_t node_count,
              const AobSurfel* surfels, const uint32_t* section_index,
              uint64_t begin, uint64_t end,
              const AobOptions* opt, const float* hemisphere,
              int32_t width, int32_t height, AobResult* results,
              int32_t threads, OrcTraceStats* stats);

/* orc_trace with section visibility (AOB_MESH_* flags per mesh + the mesh of every surfel; NULL: none) and the
 * material's ao_max_range (used when AobOptions::material_ranges is set; 0: none). */
int orc_trace_ex(const AobBvhNode* nodes, uint64_t node_count,
                 const AobSurfel* surfels, const uint32_t* section_index,
                 const uint32_t* surfel_mesh, const uint32_t* mesh_flags, uint32_t mesh_count, float ao_max_range,
                 uint64_t begin, uint64_t end,
                 const AobOptions* opt, const float* hemisphere,
                 int32_t width, int32_t height, AobResult* results,
                 int32_t threads, OrcTraceStats* stats);

/* PostprocessPixelShader.hlsl:31-86 + the render-target format conversion
 * (BakeEngine.cpp:482) + write-back layout (:705-709). */
void orc_resolve(const AobResult* results, int32_t width, int32_t height, const AobOptions* opt,
                 int32_t channels, int32_t bytes_per_channel, uint8_t* out);

/* BakeEngine::Bake end to end (BakeEngine.cpp:335-751) into
 * scene->materials[i].data. */
int orc_bake(const AobSceneDesc* scene, const AobOptions* opt, int32_t threads, OrcTraceStats* stats);

/* ---- the reference's own records (for the tests that run the reference's OpenCL kernels, oracle/ref_kernels.mk) ---- */

/* 48 B: RadeonRays/math/ray.h:103-107, kernels/CL/common.cl:53-60 */
typedef struct OrcRay { float o[4]; float d[4]; int32_t extra[2]; int32_t backface_cull; int32_t padding; } OrcRay;
/* 32 B: RadeonRays/radeon_rays.h:140-154, kernels/CL/common.cl:63-70 */
typedef struct OrcIntersection { int32_t shape_id, prim_id, pad[2]; float uvwt[4]; } OrcIntersection;

/* GenerateRaysComputeShader.hlsl:43-54: sample_count ray records per surfel of [begin, end) — the rays orc_trace casts.
 * t_max goes to o.w, `mask` to extra.x (the reference writes 0); surfels without a finite tangent frame get inactive
 * records (extra.y = 0).  Returns the record count. */
uint64_t orc_generate_rays(const AobSurfel* surfels, const uint32_t* section_index, uint64_t begin, uint64_t end,
                           const AobOptions* opt, const float* hemisphere, int32_t width, float t_max, int32_t mask,
                           OrcRay* out);
/* intersect_main (intersect_bvh2_lds.cl:65-226) / occluded_main (:228-363) over ray records, as the kernels write them. */
void orc_intersect_rays(const AobBvhNode* nodes, uint64_t node_count, const OrcRay* rays, uint64_t n, int32_t use_mask,
                        OrcIntersection* hits);
void orc_occluded_rays(const AobBvhNode* nodes, uint64_t node_count, const OrcRay* rays, uint64_t n, int32_t* hits);
/* high-water mark of deferred entries per ray in that walk (>= 16: the kernel's LDS stack spilt, :144-153) */
void orc_reference_stack_depths(const AobBvhNode* nodes, uint64_t node_count, const OrcRay* rays, uint64_t n, int32_t any_hit,
                                uint32_t* depth_out);
/* ProcessIntersectionsComputeShader.hlsl:52-92 from Intersection records (accumulates into results like orc_trace). */
void orc_process_intersections(const AobSurfel* surfels, uint64_t begin, uint64_t end, const OrcRay* rays,
                               const OrcIntersection* hits, const AobOptions* opt, float falloff,
                               int32_t width, int32_t height, AobResult* results);
/* The Morton codes orc_bvh_build_ex sorts by, per primitive (unsorted), with the primitive boxes and the scene box in
 * the reference's bbox record form (8 floats each; NULL: not wanted). */
void orc_morton_codes(const OrcTriangle* tris, uint64_t n, int adaptive, uint64_t* codes, float* prim_boxes8, float* scene_box8);

/* ---- two-level walk (instancing; kernels/CL/intersect_bvh2level_skiplinks.cl:75-131,134-307) ---------------------- */

/* One shape of the top level: rows m0, m1, m2 of its world-to-object matrix in column-vector form (Shape::m0..m2,
 * :84-87; the reference hands RadeonRays transpose(inverse(model)), BakeEngine.cpp:390-393), the id its hits report
 * and which bottom-level tree it shares. */
typedef struct OrcInstance { float world_to_object[12]; uint32_t shape_id; uint32_t blas; } OrcInstance;
/* Closest (any_hit 0) or any (1) hit of ray records against top-level tree + instances + bottom-level trees, with the
 * stack discipline of traverse() on both levels.  hits may be NULL (counting only).  st_top: inner_visits = top-level
 * node fetches, leaf_visits = instances entered; st_bottom: the bottom-level trees' node fetches and triangle tests. */
int orc_two_level_intersect(const AobBvhNode* tlas, uint64_t tlas_count, const OrcInstance* instances,
                            const AobBvhNode* const* blas_nodes, const uint64_t* blas_counts,
                            const OrcRay* rays, uint64_t n, int32_t any_hit, OrcIntersection* hits,
                            OrcTraceStats* st_top, OrcTraceStats* st_bottom);
/* The flat tree over the same records, counted the same way. */
int orc_flat_intersect(const AobBvhNode* nodes, uint64_t node_count, const OrcRay* rays, uint64_t n, int32_t any_hit,
                       OrcIntersection* hits, OrcTraceStats* st);

/* ---- measurement helpers (tests/measure_tree_variants.py) ---------------------------------------------------------- */
int orc_walk4q_from(const AobBvh4QNode* nq, uint64_t count4, const AobQuantFrame* fr, const AobLeafTri* tris,
                    const OrcRay* rays, uint64_t n, int32_t any_hit, const uint32_t* start, int32_t n_start, OrcTraceStats* all, double* t_sum);
int orc_walk4q(const AobBvh4QNode* nq, uint64_t count4, const AobQuantFrame* fr, const AobLeafTri* tris,
               const OrcRay* rays, uint64_t n, int32_t any_hit, OrcTraceStats* all, OrcTraceStats* of_hit_rays);
int orc_kary_visits(const AobBvhNode* nodes2, uint64_t node_count2, const OrcRay* rays, uint64_t n, int32_t levels, int32_t pairs,
                    int32_t any_hit, OrcTraceStats* st, uint64_t* box_tests);
uint64_t orc_bvh_build_sah(const OrcTriangle* tris, uint64_t n, AobBvhNode* nodes, uint32_t* depth);
/* the any-hit walk with a tile's entry list (trace.hip, ENTRY mode), counted per phase: tests/measure_entry_lists.py */
typedef struct OrcEntryStats { uint64_t rays, hits, inner_visits, leaf_visits, entries_tested, mask_bits; } OrcEntryStats;
/* binned SAH with spatial splits, measurement only (tools/measure_spatial_splits.py) */
uint64_t orc_bvh_build_sbvh(const OrcTriangle* tris, uint64_t n, AobBvhNode* nodes, uint64_t node_capacity, uint64_t max_refs, float alpha, uint32_t* depth, uint64_t* leaves);
/* tile cuts (trace.hip CUT mode): a cut of the 4-ary tree for a tile's box, and the any-hit walk from it, counted */
int32_t orc_tile_cut(const AobBvh4QNode* nq, uint64_t count4, const AobQuantFrame* fr, const float lo[3], const float hi[3],
                     int32_t max_entries, int32_t rule, uint32_t* out);
int orc_cut_walk(const AobBvh4QNode* nq, uint64_t count4, const AobQuantFrame* fr, const AobLeafTri* tris,
                 const OrcRay* rays, uint64_t n, const uint32_t* cut, int32_t n_cut, OrcEntryStats* out, int32_t* hit_out, uint64_t* mask_out);
int orc_walk4q_t(const AobBvh4QNode* nq, uint64_t count4, const AobQuantFrame* fr, const AobLeafTri* tris,
                 const OrcRay* rays, uint64_t n, int32_t any_hit, OrcTraceStats* all, float* t_out);
int32_t orc_resolve_bvh_builder(const AobOptions* opt, uint64_t texels, uint64_t triangles);
/* round 5: cuts for rays of bounded reach, and the closest-hit walk from a cut (trace.hip CUT mode, closest-hit scoring) */
int32_t orc_tile_cut_reach(const AobBvh4QNode* nq, uint64_t count4, const AobQuantFrame* fr, const float lo[3], const float hi[3],
                           const float flo[3], const float fhi[3], float reach, int32_t max_entries, float grow2, uint32_t* out);
int orc_cut_walk_closest(const AobBvh4QNode* nq, uint64_t count4, const AobQuantFrame* fr, const AobLeafTri* tris,
                         const OrcRay* rays, uint64_t n, const uint32_t* cut, int32_t n_cut, int32_t order, OrcEntryStats* out, float* t_out);
int orc_entry_walk(const AobBvh4QNode* nq, uint64_t count4, const AobQuantFrame* fr, const AobLeafTri* tris,
                   const OrcRay* rays, uint64_t n, const uint32_t* entries, int32_t n_entries, OrcEntryStats* out);

#ifdef __cplusplus
}
#endif
#endif

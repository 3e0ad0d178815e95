/*
 * aobaker.h — C ABI of the MI355X-native ambient-occlusion bake path.
 *
 * This is the drop-in boundary.  The reference (epicabsol/aobaker) has no FFI
 * layer of its own: its bake is the free function
 *     namespace BakeEngine { void Init(); void Dispose(); void Bake(Scene*); }
 * (reference BakeEngine.h:5-10, implementation BakeEngine.cpp:118-751).
 * The entry points below are what a C++ shim for that function binds
 * (include/BakeEngine.h + aobaker_amd/host/BakeEngine.cpp in this repo);
 * every entry point names the piece of the reference it replaces.
 *
 * Plain C, plain pointers and sizes; no torch / HIP types in any signature.
 * All functions return AOB_OK (0) or a negative AobStatus; the text of the
 * last failure on the calling thread is available from aob_last_error().
 */
#ifndef AOBAKER_H
#define AOBAKER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AOB_ABI_VERSION 7
#define AOB_BVH_BUILDER_AUTO (-1)

typedef enum AobStatus {
    AOB_OK = 0,
    AOB_ERR_INVALID_ARGUMENT = -1,
    AOB_ERR_NOT_INITIALISED = -2,
    AOB_ERR_DEVICE = -3,          /* a HIP call failed; see aob_last_error() */
    AOB_ERR_OUT_OF_MEMORY = -4,
    AOB_ERR_UNSUPPORTED = -5,
    AOB_ERR_BVH_TOO_DEEP = -6
} AobStatus;

/* ---- options ------------------------------------------------------------
 * Defaults (aob_default_options) are the reference's compile-time constants,
 * BakeEngine.cpp:13-18: sampleCount=100, falloffDistance=1.25f,
 * useAnyHit=true, rayChunkSize=0xFFFFFF, bleedDistance=5. */

enum { AOB_SAMPLER_REFERENCE_COMPAT = 0,   /* BakeEngine.cpp:312-333, MSVC rand() table */
       AOB_SAMPLER_COSINE = 1 };           /* counter-based stratified cosine hemisphere */
enum { AOB_ACCUMULATE_QUANTISED = 0,       /* ProcessIntersectionsComputeShader.hlsl:87-90 */
       AOB_ACCUMULATE_FLOAT = 1 };
enum { AOB_FP16_OFF = 0,                   /* keep normal + V in fp32 */
       AOB_FP16_RTZ = 1,                   /* through fp16, round toward zero */
       AOB_FP16_RNE = 2 };                 /* through fp16, round to nearest even */

/* Edge bleed of uncovered texels (resolve).  REFERENCE reproduces PostprocessPixelShader.hlsl:43-78 including its
 * quirks: the window is read with the raw x and y, so at the left/right border it aliases into
 * the neighbouring rows; the radius is exclusive (d^2 < B^2) and ties go to the first hit of an x-outer scan.
 * NEAREST is the dilation the reference's readme says is unfinished: neighbours outside the texture are ignored
 * (no aliasing, no wrap), the radius is inclusive (d^2 <= B^2), and the nearest covered texel wins with ties broken
 * by scan order (rows top to bottom, then left to right). */
enum { AOB_BLEED_REFERENCE = 0, AOB_BLEED_NEAREST = 1 };

/* Which texels of a UV triangle get a surfel (rasterise).  REFERENCE is the D3D11 rule the reference draws with: the
 * texels whose CENTRE the triangle covers (top-left rule), the last triangle drawn winning a texel.  CONSERVATIVE adds
 * the texels whose SQUARE the triangle overlaps without covering the centre — a chart thinner than a texel still gets
 * its surfels instead of borrowing a neighbour's value from the bleed pass.  Such a surfel is sampled at the
 * point of the triangle nearest to the texel centre; a triangle that
 * covers the centre always wins over one that only touches the square, the last drawn winning within each class. */
enum { AOB_RASTER_REFERENCE = 0, AOB_RASTER_CONSERVATIVE = 1 };

/* Which fragments of a texel become surfels where UV triangles of ONE section overlap.  LAST is the reference: its
 * stencil peel draws one layer per pass and keeps a texel only in the pass that reaches its last fragment
 * (BakeEngine.cpp:130-140,531-588), so the earlier fragments never cast a ray.  ALL keeps every
 * layer of the peel: layer j holds, per texel, the j-th triangle (in index order) that covers its centre, each layer
 * is traced like a section of its own (its surfels numbered from 0, row-major) and all of them sum into the texel's
 * cell — overlapping charts average instead of the last one winning.  At most 255 layers, the stencil's range.
 * Centre sampling only: not combinable with AOB_RASTER_CONSERVATIVE. */
enum { AOB_LAYERS_LAST = 0, AOB_LAYERS_ALL = 1 };

typedef struct AobOptions {
    int32_t  sample_count;        /* rays per surfel (reference sampleCount)              */
    int32_t  sampler;             /* AOB_SAMPLER_*                                        */
    uint32_t seed;                /* COSINE sampler only                                  */
    int32_t  use_any_hit;         /* 1: binary occlusion; 0: t/falloff scoring            */
    float    falloff_distance;    /* reference falloffDistance                            */
    float    ray_tmax;            /* parametric max t (reference 1e6, GenerateRays:49)    */
    float    ray_bias;            /* origin offset along the ray (reference 0.001, :47)   */
    int32_t  bleed_distance;      /* reference bleedDistance                              */
    int32_t  accumulate;          /* AOB_ACCUMULATE_*                                     */
    int32_t  normal_fp16;         /* AOB_FP16_*: the RGBA16F NormalV target, :446         */
    int32_t  ray_chunk_size;      /* reference rayChunkSize; bounds the compat table      */
    int32_t  clamp_tmax_to_falloff; /* falloff mode: cast with tmax=min(tmax,falloff)     */
    int32_t  bvh_width;           /* traversal structure: 2 = the reference's BVH2 nodes; 4 = BVH2 collapsed to
                                     4-ary nodes (quantised only: bvh_quantised must be 1); 0 = 2.  One lane per
                                     ray either way.  Default 4                                                    */
    int32_t  bvh_quantised;       /* default 1.  With bvh_width 4: 1 = the walk of AobBvh4QNode; 0 is REFUSED with
                                     AOB_ERR_UNSUPPORTED since ABI 6 (the quad-per-ray walk of the fp32 AobBvh4Node was an
                                     experiment 2.7x slower, retired to tools/experiments/).  With bvh_width 2:  0: 64-byte
                                     fp32 nodes (the reference layout); 1: 32-byte nodes, child boxes as 15-bit fixed point
                                     on the scene box, rounded outward (AobBvh2QNode)                               */
    int32_t  bvh_builder;         /* -1 (AOB_BVH_BUILDER_AUTO, the default): 1 or 2 by workload, see aob_resolve_bvh_builder;
                                     0: LBVH, the reference stack's GPU builder (kernels/CL/build_hlbvh.cl);
                                     1: PLOC (locally-ordered agglomerative clustering over the same Morton
                                     order) — same node layouts, cheaper trees, ~0.5 ms more build time;
                                     2: PLOC until 131 072 clusters are left, the tree above them top-down by binned SAH
                                     (the reference stack's "sah" builder, RadeonRays/radeon_rays.h:286-294): 5 % fewer
                                     node visits per ray on the 1 M-triangle terrain for 0.6 ms more build — worth it
                                     from a few hundred million rays per bake on (any-hit bakes)          */
    int32_t  bleed_mode;          /* AOB_BLEED_*: how uncovered texels borrow a neighbour's value     */
    int32_t  material_ranges;     /* 1: honour AobMaterialDesc::ao_max_range; default 0 (the reference ignores it) */
    int32_t  morton_adaptive;     /* default 1: the Morton bits both builders sort by are dealt to the axes in the order a
                                     spatial-median builder would split them (the longest cell first, radeon_rays.h:283-287)
                                     instead of x, y, z in turn (0: build_hlbvh.cl's order).  Flat scenes gain: 8 % fewer
                                     node visits per ray on the 1 M-triangle terrain                                  */
    int32_t  raster_mode;         /* AOB_RASTER_*; default REFERENCE                                  */
    int32_t  raster_layers;       /* AOB_LAYERS_*; default LAST                                       */
    /* Schedule of the fused trace kernel (trace.hip).  These eight move work between wavefronts and never change a result
     * cell (tests/test_gpu_parity.py::test_trace_schedule_knobs_do_not_change_results).  aob_default_options writes the
     * values measured fastest on MI355X for BASELINE configs[2]; 0 in any of them also means "that default". */
    int32_t  trace_refill_idle;   /* lanes of a wavefront without a ray that trigger the refill section: 1..64; default 0 = by
                                     scoring mode, for rays that start below tile cuts (trace_cut, the default): 16 with the COSINE sampler
                                     and any-hit scoring, 24 with the table sampler and any-hit scoring, 8 with closest-hit scoring (from
                                     the root: 42 / 24 / 16)                                                                 */
    int32_t  trace_leaf_wait;     /* lanes parked at a leaf that trigger the leaf section: 1..64; default 0 = by scoring mode:
                                     24 with any-hit scoring, 16 with closest-hit scoring                                  */
    int32_t  trace_parts;         /* sample parts a tile's rays are cut into (units of one work queue), default 4          */
    int32_t  trace_part_min;      /* a part holds at least this many samples, default 16                                   */
    int32_t  trace_fine_div;      /* the last 1/trace_fine_div of every queue is cut into half-length units, default 8     */
    int32_t  trace_bands;         /* 1 (default): on a fully covered atlas, 8x8-tile blocks inside bands of 8 tile rows;
                                     2: plain chunks of 2048 tiles                                                          */
    int32_t  trace_scalar_min;    /* any-hit walk of the quantised 4-ary nodes: lanes that must stand on ONE node for it to be
                                     visited from scalar registers (after a refill): 1..64; 0 (default) = by sampler: 24
                                     with the COSINE sampler, 8 with the table sampler (its rays part ways sooner)          */
    int32_t  trace_cut;           /* walk of the quantised 4-ary nodes (both samplers; any-hit and, since ABI 6, closest-hit scoring): every
                                     tile of 64 surfels gets a CUT of the tree — the siblings of the chain of nodes its rays start inside,
                                     at most this many entries, none farther away than its rays reach — which a new ray tests in one go
                                     when it is generated, starting its walk below the entries it entered instead of at the root
                                     (trace.hip k_tile_cuts).  The COSINE sampler's units get one record per sample part,
                                     culled by the part's azimuth wedge; the table sampler one record per tile, no wedge culling.
                                     0 (default) = 32 entries, 4..32 = that many, 1 = off (the walk from the root).  Like the knobs above
                                     it moves work and never changes a result cell.  The cuts take 512 B per tile and record of device
                                     memory (up to 96 B per covered texel), allocated when a material is rasterised; a bake that cannot
                                     have them walks from the root and says so (AobStats::trace_cuts).  (the field: ABI 5; closest-hit
                                     rays below cuts and AobStats::trace_cuts: ABI 6)                                                  */
    int32_t  trace_horizon;       /* COSINE sampler on a full sx x sy grid (sample_count = sx * sy, both 4..32), quantised 4-ary nodes: every surfel
                                     gets a HORIZON TABLE — for each azimuth row of the sampler's grid the number of leading columns (the
                                     directions nearest the normal) that no triangle of the scene can reach, proven from the tree with
                                     conservative bounds (horizon.hip) — and the trace kernel does not cast the samples of those cells: they are
                                     unoccluded whatever their jitter and score as cast rays would.  The tables cost time per surfel and save
                                     time per sample: 0 (default) = from 128 samples per surfel on, 1 = off, 2 = wherever they apply.
                                     Moves work, never changes a result cell.  16 or 32 bytes per covered texel of device memory; a bake that
                                     cannot have them casts every sample and says so (AobStats::trace_horizon = 2).  (ABI 7)              */
} AobOptions;

/* ---- scene description ----------------------------------------------------
 * One AobMeshDesc per (BakeObject, Section) pair in the order the reference
 * walks them (BakeEngine.cpp:370-398 and :509-523).  Vertex attributes are
 * strided so a reference RenderVertex* (48 B: pos@0, normal@12, uv@24,
 * Render/RenderMesh.h:13-19) can be passed as is. */
typedef struct AobMeshDesc {
    const void*     positions;        /* float[3] per vertex */
    const void*     normals;          /* float[3] per vertex */
    const void*     uvs;              /* float[2] per vertex */
    uint32_t        position_stride;  /* bytes */
    uint32_t        normal_stride;
    uint32_t        uv_stride;
    uint32_t        vertex_count;
    const uint32_t* indices;          /* triangle list */
    uint32_t        index_count;      /* multiple of 3 (a trailing partial triangle is ignored) */
    int32_t         material;         /* index into AobSceneDesc::materials, or -1: occluder only */
    float           model[16];        /* row-major, row-vector: world = [p 1] * model
                                         (TransformObject.cpp:21-24, RasterizeVertexShader.hlsl:30-32) */
    uint32_t        flags;            /* AOB_MESH_*; 0 = the reference's behaviour */
    uint32_t        reserved;
} AobMeshDesc;

/* Section visibility (BakeObject.h:28-31,49-50: `VisibleToSelf`, `VisibleToOthers` — declared by the reference, shown
 * in its property grid, never read by its Bake).  Here they act as RadeonRays' ray masks would
 * (common.cl:76-80, intersect_bvh2_lds.cl:163-166 skip the triangles of one mesh id): a ray cast from a surfel of
 * mesh A ignores the triangles of mesh B when B == A and A is hidden from itself, or B != A and B is hidden from
 * others.  Hidden triangles still get their own texels baked. */
enum { AOB_MESH_HIDDEN_FROM_SELF = 1u, AOB_MESH_HIDDEN_FROM_OTHERS = 2u };

/* The output side: BakeTexture's data contract (Scene/BakeTexture.cpp:7-67,
 * write-back BakeEngine.cpp:705-709): row-major, top row first, tightly
 * packed width*height*channels*bytes_per_channel bytes; AO in channel 0. */
typedef struct AobMaterialDesc {
    int32_t  width;
    int32_t  height;
    int32_t  channels;            /* 1,2,4 (1- and 2-byte) or 1..4 (4-byte) */
    int32_t  bytes_per_channel;   /* 1: UNORM8, 2: fp16, 4: fp32 */
    uint8_t* data;                /* host memory, written by aob_bake / aob_bake_download */
    float    ao_max_range;        /* MaterialObject::AOMaxRange (MaterialObject.h:10-11,18; never read by the reference's
                                     Bake).  Used only with AobOptions::material_ranges: rays cast from this material's
                                     texels end at min(ray_tmax, ao_max_range) and, in falloff mode, fall off over
                                     ao_max_range instead of falloff_distance.  <= 0: the options' values */
    int32_t  reserved;
} AobMaterialDesc;

typedef struct AobSceneDesc {
    const AobMeshDesc*     meshes;
    uint32_t               mesh_count;
    const AobMaterialDesc* materials;
    uint32_t               material_count;
} AobSceneDesc;

/* aob_init flags.  By default the first aob_init of a process on a device bakes a one-triangle scene there and throws it away:
 * the library's code object is loaded, the device's context, allocator and streams have been used once, and the FIRST real
 * Bake() costs what every later one costs (BASELINE configs[2]: 236 ms -> the 65 ms of a warm bake; the reference's Init pays
 * its shader compiles and ~3 GB of buffers up front likewise, BakeEngine.cpp:66-212).  NO_WARMUP skips it. */
#define AOB_INIT_NO_WARMUP 1
typedef struct AobDeviceConfig {
    int32_t device;               /* HIP device ordinal, -1 = current device */
    int32_t flags;                /* AOB_INIT_* (was reserved[0]: zero keeps meaning "defaults") */
    int32_t reserved[6];
} AobDeviceConfig;

typedef struct AobStats {
    uint64_t triangle_count;
    uint64_t node_count;
    uint64_t surfel_count;        /* covered texels summed over sections and materials */
    uint64_t ray_count;           /* surfel_count * sample_count */
    uint32_t bvh_depth;
    uint32_t trace_cuts;          /* the worst over the materials of what each one's trace launches since ITS last rasterisation reported — 0: walked from the root by design (a mode without tile
                                     cuts, trace_cut 1); 1: started their rays below tile cuts (AobOptions::trace_cut); 2: at least one wanted
                                     cuts and walked from the root because their memory could not be allocated — the same cells, slower;
                                     aob_last_error() then holds a line that starts with "warning:" (ABI 6; was reserved0)           */
    float ms_upload;
    float ms_bvh_build;
    float ms_rasterize;
    float ms_trace;
    float ms_resolve;
    float ms_download;
    float ms_total_device;        /* bvh_build + rasterize + trace + resolve */
    float reserved1;
    uint64_t rays_cast;           /* of ray_count, the samples the trace kernels generated and walked; the rest were resolved by the surfels' horizon
                                     tables (AobOptions::trace_horizon) as unoccluded without being cast.  ray_count where no launch had tables (ABI 7) */
    uint32_t trace_horizon;       /* the trace launches since the last rasterisation — 0: no horizon tables (off, or a mode they do not apply to);
                                     1: had them; 2: at least one wanted them and could not allocate them (same cells, every sample cast)   */
    uint32_t reserved2;
} AobStats;

/* ---- device records (layouts shared with the parity tests) ---------------- */

/* 40 B, identical to the reference TexelRasterization (BakeEngine.cpp:98-116,
 * GenerateRaysComputeShader.hlsl:1-8). */
typedef struct AobSurfel {
    int32_t texel_x, texel_y;
    float   position[3];
    float   texcoord[2];
    float   normal[3];
} AobSurfel;

/* 64 B traversal node, the layout of the reference's bvh_node
 * (kernels/CL/intersect_bvh2_lds.cl:40-52).  Inner: child boxes + child
 * addresses; leaf (addr_left == 0xffffffff): the triangle, inlined. */
typedef struct AobBvhNode {
    float    lmin_or_v0[3];  uint32_t addr_left;
    float    lmax_or_v1[3];  uint32_t shape_id;
    float    rmin_or_v2[3];  uint32_t addr_right;
    float    rmax[3];        uint32_t prim_id;
} AobBvhNode;
#define AOB_INVALID_ADDR 0xffffffffu

/* 128 B 4-ary node: every second level of the BVH2 above collapsed away (the reference stack's
 * own 4-wide kernel is kernels/CL/intersect_bvh2_lds_fp16.cl; boxes stay fp32 here).
 * lo[j] = {min.x, min.y, min.z, ref_j}, hi[j] = {max.x, max.y, max.z, 0}.  ref < 0x80000000: index
 * of a 4-ary node (= the BVH2 index of the inner node it was made from); ref = 0x80000000 | p:
 * leaf triangle p of AobLeafTri[]; empty slot: ref = AOB_INVALID_ADDR and an inverted box.
 * The two 64-byte halves are what the four lanes of a ray's quad fetch with one access each. */
typedef struct AobBvh4Node { float lo[4][4]; float hi[4][4]; } AobBvh4Node;
#define AOB_LEAF_FLAG 0x80000000u
/* With bvh_width 4 and bvh_quantised 1 a leaf ref may also carry AOB_PAIR_FLAG: ref = AOB_LEAF_FLAG | AOB_PAIR_FLAG | p
 * names the two triangles p and p + 1 of AobLeafTri[] (a BVH2 node whose children are those two leaves takes one
 * child slot instead of being a node of its own).  AobBvh4Node[] and AobBvh4QNode[] of such a bake both carry them. */
#define AOB_PAIR_FLAG 0x40000000u
/* 32 B BVH2 node with quantised child boxes: q = 15-bit fixed point on the scene box (0..32767,
 * scale = extent / 32767), box.min rounded down, box.max rounded up; decode: frame.origin + q * frame.scale.
 * One word per axis and box, min in the low half: w0 = lmin.x | lmax.x<<16, w1 = lmin.y | lmax.y<<16,
 * w2 = lmin.z | lmax.z<<16, w3..w5 the same for the right child, w6 = addr_left, w7 = addr_right.  Inner nodes only ([0, n-1)); a child address >= n-1 is leaf
 * triangle (address - (n-1)) of AobLeafTri[].  Two 16-byte fetches per visit instead of four. */
typedef struct AobBvh2QNode { uint32_t w[8]; } AobBvh2QNode;
typedef struct AobQuantFrame { float origin[3]; float scale[3]; } AobQuantFrame;
/* 64 B 4-ary node with quantised child boxes (same grid and rounding as AobBvh2QNode): child j is
 * w[4j..4j+2] = x, y, z words and w[4j+3] = ref_j (as in AobBvh4Node); an empty slot has
 * the words 0x8000ffff — an inverted box (min plane 0x7fff, max plane 0) that no ray enters, so the walk does not test the
 * ref — and ref = AOB_LEAF_FLAG | 0: should rounding ever let such a box pass (a scene without extent), the walk tests
 * triangle 0 once more, which changes no answer.  A word is (0x8000 | min) | (0x8000 | max) << 16: bit 15 of each
 * half is set so that the bytes {0x3f, half>>8, half&0xff, 0x00} are the IEEE float 1 + q/32768 — one
 * v_perm_b32 per plane in the traversal; decode q = half & 0x7fff.  The array is packed (no records for the
 * collapsed-away levels): inner refs index this array, not AobBvh4Node[].  Walked one lane per ray
 * (bvh_width 4, bvh_quantised 1). */
typedef struct AobBvh4QNode { uint32_t w[16]; } AobBvh4QNode;
/* 48 B leaf triangle in Morton order: v0.w = shape id bits, v1.w = prim id bits. */
typedef struct AobLeafTri { float v0[4], v1[4], v2[4]; } AobLeafTri;

/* 8 B per texel (ResultValue, BakeEngine.cpp:298-309); both fields are int32
 * in QUANTISED mode and float32 in FLOAT mode. */
typedef struct AobResult { uint32_t total_bits, weight_bits; } AobResult;

/* ---- one-shot API --------------------------------------------------------- */

/* Replaces BakeEngine::Init (BakeEngine.cpp:118-213): checks that the device is a gfx950 and makes it the device every
 * later aob_bake_create of the calling THREAD allocates on (one host thread per GPU: each thread calls aob_init with
 * its own device); a thread that never called aob_init creates its handles on the device of the most recent aob_init of
 * any thread.  A handle stays on the device it was created on: every aob_bake_* entry point selects that device for the
 * duration of the call and restores the calling thread's current device before it returns, so a handle may be driven from
 * any thread — one call at a time — and a thread that shares HIP with other code keeps its own current device.  It KEEPS nothing:
 * streams, events and buffers belong to the AobBake handles (the reference's Init allocates ~3 GB of fixed buffers here,
 * BakeEngine.cpp:163-212; this library sizes them per scene).  Unless AOB_INIT_NO_WARMUP is set, the first aob_init of the process on a
 * device does RUN one throw-away bake of a triangle there (a 16x16 atlas, 16 samples: every stage's kernels, k_trace included, are
 * launched once and show up in a profile of the process); a second thread's aob_init on the same device waits for it to finish. */
int aob_init(const AobDeviceConfig* config);
/* Replaces BakeEngine::Dispose (BakeEngine.cpp:215-279).  Since aob_init keeps nothing there is nothing to release beyond the device
 * selection (the calling thread's, and the process-wide one threads without their own fall back to) and — when the LAST thread that called
 * aob_init shuts down — the process-wide hemisphere table of REFERENCE_COMPAT bakes (host memory: 12 bytes per entry a chunk can index,
 * 200 MB with the reference's constants; kept between bakes because a prefix of the sequence is the same for every bake).  Handles still
 * alive stay valid until aob_bake_destroy. */
void aob_shutdown(void);
/* Fills `out` with the reference's constants (BakeEngine.cpp:13-18). */
void aob_default_options(AobOptions* out);
/* Replaces BakeEngine::Bake (BakeEngine.cpp:335-751): uploads the scene,
 * builds the BVH, and for every material rasterises, traces, resolves,
 * bleeds and writes materials[i].data. `stats` may be NULL. */
int aob_bake(const AobSceneDesc* scene, const AobOptions* options, AobStats* stats);
const char* aob_last_error(void);
int aob_abi_version(void);
/* AobOptions::bvh_builder -1 (the default since ABI 6) lets the library choose by workload: 2 — PLOC under the binned-SAH top — for an
 * any-hit bake of at least 2^28 rays (texels x sample_count, the bound known before the rasterisation) over more than 131 072 triangles,
 * else 1 (PLOC).  aob_bake_create settles it for the handle; this returns what it would settle on (an explicit 0 / 1 / 2 comes back
 * unchanged).  `texels`: width x height summed over the scene's materials. */
int aob_resolve_bvh_builder(const AobOptions* options, uint64_t texels, uint64_t triangles);

/* ---- staged API ------------------------------------------------------------
 * The same path cut at the reference's own stage boundaries (S1..S6 in
 * BakeEngine.cpp), so that a caller can keep the scene resident in HBM,
 * time stages, shard the surfel list over GPUs, and read back intermediate
 * buffers for parity tests. */
typedef struct AobBake AobBake;

/* S1 upload: copies vertex/index data to HBM (BakeEngine.cpp:370-398). */
int aob_bake_create(const AobSceneDesc* scene, const AobOptions* options, AobBake** out);
void aob_bake_destroy(AobBake* bake);
/* S1 Commit(): world-space triangles + LBVH build (BakeEngine.cpp:400;
 * kernels/CL/build_hlbvh.cl). */
int aob_bake_build_bvh(AobBake* bake);
/* S2 + S2': UV rasterisation, last-fragment-wins, row-major compaction into
 * the surfel list of `material` (BakeEngine.cpp:531-588). */
int aob_bake_rasterize(AobBake* bake, int32_t material);
/* S3..S5 for surfels [begin,end) of the material's surfel list IN TRACE ORDER (sections
 * concatenated in scene order; inside a section 8x8 texel tiles, see aob_bake_shard_bounds):
 * ray generation, traversal, scoring, accumulation into the material's
 * W*H result buffer (BakeEngine.cpp:598-675).  end = UINT64_MAX means all. */
/* aob_bake_rasterize restricted to the texel rows [row_begin, row_end) of the atlas: the surfel list then holds that
 * strip only (the rest of the atlas stays uncovered).  A multi-GPU bake gives every rank its own strip this way instead
 * of rasterising the whole atlas on every rank (aob_bake_shard_bounds balances by surfel count but needs the whole list).
 * Surfels are numbered within the strip: the COSINE sampler (keyed by texel) gives the cells of the whole bake, the
 * REFERENCE_COMPAT sampler (whose table index is the surfel's number in its section's list, BakeEngine.cpp:598-628)
 * an equally valid but different assignment of table entries — use aob_bake_shard_bounds where that must match. */
int aob_bake_rasterize_rows(AobBake* bake, int32_t material, int32_t row_begin, int32_t row_end);
int aob_bake_trace(AobBake* bake, int32_t material, uint64_t begin, uint64_t end);
/* S6: resolve + bleed into the device image (BakeEngine.cpp:688-704). */
int aob_bake_resolve(AobBake* bake, int32_t material);
/* S6 write-back: device image -> host (BakeEngine.cpp:705-709).
 * dst == NULL writes to the AobMaterialDesc::data given at creation. */
int aob_bake_download(AobBake* bake, int32_t material, void* dst, size_t dst_bytes);
/* All device stages for all materials, no host transfers:
 * build_bvh, then per material rasterize + trace(all) + resolve. */
int aob_bake_run_device(AobBake* bake);
int aob_bake_sync(AobBake* bake);
int aob_bake_get_stats(AobBake* bake, AobStats* stats);
/* Zeroes the material's result buffer (done by rasterize as well). */
int aob_bake_clear_results(AobBake* bake, int32_t material);

/* Introspection used by the parity tests and the multi-GPU gather. */
int aob_bake_node_count(AobBake* bake, uint64_t* count);
int aob_bake_copy_nodes(AobBake* bake, AobBvhNode* dst, uint64_t capacity);
/* The 4-ary structure (built by aob_bake_build_bvh when bvh_width is 4): node4_count entries of
 * AobBvh4Node (indexed by BVH2 inner-node index; entries of collapsed-away levels are zero) and
 * triangle_count entries of AobLeafTri. */
int aob_bake_node4_count(AobBake* bake, uint64_t* count);
int aob_bake_copy_nodes4(AobBake* bake, AobBvh4Node* dst, uint64_t capacity);
int aob_bake_copy_leaf_tris(AobBake* bake, AobLeafTri* dst, uint64_t capacity);
/* The quantised BVH2 (max(0, n-1) AobBvh2QNode entries) and its frame. */
int aob_bake_copy_nodes2q(AobBake* bake, AobBvh2QNode* dst, uint64_t capacity);
int aob_bake_quant_frame(AobBake* bake, AobQuantFrame* frame);
/* The quantised 4-ary nodes, packed: the live entries of the AobBvh4Node array (those whose slot 0 holds a ref) in
 * index order, refs renumbered accordingly; node4q_count entries of AobBvh4QNode. */
int aob_bake_node4q_count(AobBake* bake, uint64_t* count);
int aob_bake_copy_nodes4q(AobBake* bake, AobBvh4QNode* dst, uint64_t capacity);
/* The tile cuts the material's most recent trace launch walked from (trace_cut; any-hit walk of the quantised 4-ary nodes): per
 * 64-surfel tile of the launch (trace order: 8x8-texel tiles) `records_per_tile` records of 128 words — ref[32], then the 32 entries'
 * {x, y, z} plane words as in AobBvh4QNode, the entries in use first (in the cut's breadth-first list order), empty slots (planes
 * 0x8000ffff) after.  One record per tile, or one per sample part (the coarse parts, then the fine ones) holding the entries that
 * part's rays can reach.  *tiles = 0 if the launch used no cuts.  For the parity tests (the oracle's orc_tile_cut). */
int aob_bake_copy_tile_cuts(AobBake* bake, int32_t material, uint32_t* dst, uint64_t capacity_words, uint32_t* tiles, uint32_t* records_per_tile);
/* The horizon tables the material's most recent trace launches used (AobOptions::trace_horizon; horizon.hip): `*stride` bytes per surfel, in the
 * order of aob_bake_copy_surfels; byte cy = the number of leading cells of row cy of the COSINE sampler's grid whose samples were not cast
 * (no triangle of the scene can be reached through them).  *surfels = 0 if the launches had no tables.  For the tests: every ray of such a
 * cell must be unoccluded for the oracle. */
int aob_bake_copy_horizon(AobBake* bake, int32_t material, uint8_t* dst, uint64_t capacity_bytes, uint32_t* stride, uint64_t* surfels);
/* Diagnostics of the horizon-table kernel since the bake was created (or last cleared), summed over its tiles: out16[0..8] = walk rounds, child
 * boxes measured, near triangles, middle-distance leaves, tiles that got no tables (lists full), tiles, clock ticks of the kernel's three phases.
 * The kernel counts only after a call with clear = 2 (nine contended atomics per tile cost more than the kernel itself); clear = 1 zeroes the counters. */
int aob_bake_horizon_counters(AobBake* bake, unsigned long long* out16, int clear);
/* HIP-event times of the handle's most recent stages, in ms (-1: the stage has not run through its staged entry point): out6[0] the tree's build,
 * [1] the rasterisation (aob_bake_rasterize / _rows: host read-backs included), [2] the trace (everything aob_bake_trace enqueued), and of the
 * trace's most recent LAUNCH [3] its tile cuts, [4] its horizon tables, [5] its walk (k_trace).  Waits for the stream. */
int aob_bake_last_stage_ms(AobBake* bake, float* out6);
/* Tests only: from a call with on != 0 every allocation of tile cuts fails (process-wide), as on a device without memory for them — the bakes walk
 * from the root, write the same cells and report AobStats::trace_cuts = 2 with a "warning:" in aob_last_error(). */
void aob_debug_fail_tile_cuts(int on);
int aob_bake_surfel_count(AobBake* bake, int32_t material, uint64_t* count);
int aob_bake_copy_surfels(AobBake* bake, int32_t material, AobSurfel* dst, uint64_t capacity);
int aob_bake_copy_results(AobBake* bake, int32_t material, AobResult* dst, uint64_t capacity);
/* Balanced split of the material's trace-order surfel list into `parts` contiguous ranges
 * [bounds[r], bounds[r+1]) for a multi-GPU bake (bounds has parts+1 entries).  The trace order is
 * 8x8 texel tiles, tile rows top to bottom, so with a single section the boundaries are placed on
 * tile rows and part r owns the disjoint strip of result cells [first_cell[r], last_cell[r]] — the
 * "texel tiles" rank 0 gathers (first > last: empty part).  With several sections the strips of
 * different parts may overlap: *overlapping is set and the caller must sum-reduce instead.
 * After aob_bake_rasterize_rows the list, the tile rows and the cells are those of that strip. */
int aob_bake_shard_bounds(AobBake* bake, int32_t material, uint32_t parts, uint64_t* bounds,
                          uint64_t* first_cell, uint64_t* last_cell, int32_t* overlapping);
/* Device address of the material's W*H AobResult buffer (for an RCCL
 * collective issued by the caller) and of its resolved image. */
int aob_bake_results_device_ptr(AobBake* bake, int32_t material, void** ptr, uint64_t* bytes);
int aob_bake_image_device_ptr(AobBake* bake, int32_t material, void** ptr, uint64_t* bytes);
/* cells += other_cells (cell_count = W*H AobResult records in the SAME device's memory, e.g. another handle's buffer from
 * aob_bake_results_device_ptr), enqueued on the bake's stream: int32 sums in QUANTISED mode, float sums in FLOAT mode —
 * what ncclReduce(ncclSum) does between devices, for hosts whose ranks share one device (aobaker_multi.h rehearsals).
 * QUANTISED sums are exact; FLOAT sums of cells that two parts both touched are the sum of two rounded floats.  The kernel
 * runs on THIS bake's stream: whatever writes other_cells (another handle's trace on its own stream) must have finished
 * (aob_bake_sync on that handle) before the call. */
int aob_bake_add_results(AobBake* bake, int32_t material, const void* other_cells, uint64_t cell_count);
/* The same sum with the other HANDLE named instead of its buffer: cells(bake, material) += cells(other, other_material).
 * The call orders itself behind whatever `other` has enqueued so far (an event recorded on other's stream, waited for on this
 * bake's stream) — no aob_bake_sync, nothing for the caller to forget.  Both handles on one device, materials of equal size. */
int aob_bake_add_results_from(AobBake* bake, int32_t material, AobBake* other, int32_t other_material);
/* The HIP stream the stages are enqueued on (a hipStream_t as void*). */
int aob_bake_stream(AobBake* bake, void** stream);
/* Timing of the most recent aob_bake_trace launch, measured with HIP events
 * on the library's own stream. */
int aob_bake_last_trace_ms(AobBake* bake, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* AOBAKER_H */

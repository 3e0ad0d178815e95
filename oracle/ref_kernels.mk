# oracle/ref_kernels.mk — builds the REFERENCE's own OpenCL kernels for gfx950 (test infrastructure: the pin of the oracle).
#
# Sources are compiled where they lie under /root/reference (never copied): the in-image ROCm clang turns
#   kernels/CL/intersect_bvh2_lds.cl          intersect_main :65-226, occluded_main :228-363 (+ common.cl:177-246)
#   kernels/CL/build_hlbvh.cl                 calculate_morton_code_main :101-125, emit_hierarchy_main :216-261,
#                                             refit_bounds_main :264-314
#   kernels/CL/intersect_bvh2level_skiplinks.cl  intersect_main :134-307, occluded_main :309-459 (the instancing path)
# into code objects under oracle/_ref/ which tests/refkernels.py loads with hipModuleLoad and launches with
# hipModuleLaunchKernel (oracle/ref_launch.cpp).  No stand-in headers, libraries or generated code are involved; the
# OpenCL builtins (dot, cross, mad, native_recip ...) come from the ROCm device libraries clang links by itself.
#
#   *_exact.co   -DUSE_SAFE_MATH -cl-fp32-correctly-rounded-divide-sqrt: `1.f / denom` (common.cl:204-205) and
#                safe_invdir's divisions (:226-230) are IEEE divisions — the variant the oracle restates bit for bit.
#   *_native.co  the kernels as RadeonRays builds them by default (native_recip, OpenCL's 2.5-ulp division): compared
#                within the 1e-4 tolerance of BASELINE.json.
#
# /root/reference does not exist on the GPU box: the code objects are built in the build container
# (__graft_entry__.build()) and travel with the snapshot (oracle/_ref/ is git-ignored, not gpurun-ignored).
REF     ?= /root/reference
CLANG   ?= /opt/rocm/llvm/bin/clang
CLFLAGS ?= -x cl -cl-std=CL1.2 -target amdgcn-amd-amdhsa -mcpu=gfx950 -I$(REF) -O2 -Wno-everything
EXACT   = -DUSE_SAFE_MATH -cl-fp32-correctly-rounded-divide-sqrt
OUT     = _ref

COS = $(OUT)/intersect_bvh2_lds_exact.co $(OUT)/intersect_bvh2_lds_native.co $(OUT)/intersect_bvh2_lds_mask_exact.co \
      $(OUT)/build_hlbvh.co $(OUT)/intersect_bvh2level_skiplinks_exact.co

all: $(COS)

$(OUT)/intersect_bvh2_lds_exact.co: $(REF)/kernels/CL/intersect_bvh2_lds.cl $(REF)/kernels/CL/common.cl
	@mkdir -p $(OUT)
	$(CLANG) $(CLFLAGS) $(EXACT) $< -o $@
$(OUT)/intersect_bvh2_lds_native.co: $(REF)/kernels/CL/intersect_bvh2_lds.cl $(REF)/kernels/CL/common.cl
	@mkdir -p $(OUT)
	$(CLANG) $(CLFLAGS) $< -o $@
# RR_RAY_MASK: a ray skips the triangles of the mesh whose id equals its mask (intersect_bvh2_lds.cl:163-166)
$(OUT)/intersect_bvh2_lds_mask_exact.co: $(REF)/kernels/CL/intersect_bvh2_lds.cl $(REF)/kernels/CL/common.cl
	@mkdir -p $(OUT)
	$(CLANG) $(CLFLAGS) $(EXACT) -DRR_RAY_MASK $< -o $@
$(OUT)/build_hlbvh.co: $(REF)/kernels/CL/build_hlbvh.cl $(REF)/kernels/CL/common.cl
	@mkdir -p $(OUT)
	$(CLANG) $(CLFLAGS) $(EXACT) $< -o $@
$(OUT)/intersect_bvh2level_skiplinks_exact.co: $(REF)/kernels/CL/intersect_bvh2level_skiplinks.cl $(REF)/kernels/CL/common.cl
	@mkdir -p $(OUT)
	$(CLANG) $(CLFLAGS) $(EXACT) $< -o $@

clean:
	rm -f $(COS)
.PHONY: all clean

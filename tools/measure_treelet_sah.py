"""tools/measure_treelet_sah.py — what re-building subtrees by binned SAH would buy (CPU, oracle only; measurement, no test).

Takes the LBVH (build_hlbvh.cl's tree) or the PLOC tree of BASELINE configs[2]'s terrain, replaces every maximal subtree of at
most K triangles by the oracle's top-down binned-SAH tree over the same triangles (the shape a one-workgroup-per-treelet
GPU pass could build in LDS), and walks the same rays through the quantised 4-ary collapse of each.  Result (round 2,
profiles/r02/treelet_sah_prototype.json): under an LBVH top K = 128 / 512 / 2048 -> +1.5 / -0.9 / -3.0 % of PLOC's cost
model; under the PLOC top K = 16 / 64 / 512 / 4096 -> -0.3 / -1.2 / -3.2 / -5.2 %.  The gain of a SAH tree sits in the top
levels; treelets do not reach it.

  PYTHONPATH=. python tools/measure_treelet_sah.py
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from aobaker_amd import capi, scenes
from aobaker_amd.scene import FlatScene
from oracle import oracle as orc
import bench

def treelet_sah(lbvh, K):
    nodes = lbvh.copy()
    n = (len(nodes) + 1) // 2
    leaf_base = n - 1
    L = nodes['addr_left'].astype(np.int64); R = nodes['addr_right'].astype(np.int64)
    # leaf counts, bottom-up through an explicit DFS order
    order = []; stack = [0]
    while stack:
        x = stack.pop(); order.append(x)
        if x < leaf_base: stack.append(int(L[x])); stack.append(int(R[x]))
    cnt = np.zeros(len(nodes), dtype=np.int64); cnt[leaf_base:] = 1
    for x in reversed(order):
        if x < leaf_base: cnt[x] = cnt[L[x]] + cnt[R[x]]
    # treelet roots: inner nodes with count <= K whose parent has more
    roots = []
    stack = [0]
    while stack:
        x = stack.pop()
        if x >= leaf_base: continue
        if cnt[x] <= K: roots.append(x); continue
        stack.append(int(L[x])); stack.append(int(R[x]))
    for r in roots:
        inner = []; leaves = []; st = [r]
        while st:
            x = st.pop()
            if x >= leaf_base: leaves.append(x)
            else: inner.append(x); st.append(int(R[x])); st.append(int(L[x]))
        leaves.sort()
        k = len(leaves)
        assert len(inner) == k - 1
        sub = np.zeros(k, dtype=orc.TRIANGLE)
        lf = lbvh[leaves]
        sub['v'][:, 0] = lf['lmin_or_v0']; sub['v'][:, 1] = lf['lmax_or_v1']; sub['v'][:, 2] = lf['rmin_or_v2']
        sub['shape_id'] = lf['shape_id']; sub['prim_id'] = lf['prim_id']
        sn, _ = orc.bvh_build_sah(sub)
        assert len(sn) == 2 * k - 1
        gid = np.empty(2 * k - 1, dtype=np.int64)
        inner_ids = [r] + [x for x in inner if x != r]
        gid[:k - 1] = inner_ids
        gid[k - 1:] = np.array(leaves)
        for i in range(k - 1):
            g = gid[i]
            nodes[g] = sn[i]
            nodes[g]['addr_left'] = gid[sn[i]['addr_left']]
            nodes[g]['addr_right'] = gid[sn[i]['addr_right']]
        nodes[gid[k - 1:]] = sn[k - 1:]
    return nodes, len(roots)

def main():
    atlas, spp, one_in = 2048, 64, 2048
    flat = FlatScene(scenes.terrain_scene(grid=708, atlas=atlas, seed=3))
    opt = orc.default_options()
    opt.sample_count, opt.sampler, opt.accumulate, opt.normal_fp16, opt.seed = spp, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, capi.FP16_OFF, 3
    tris = orc.scene_triangles(flat.desc)
    surfels, _ = orc.rasterize(flat.desc, 0, opt)
    sub = bench._subsample(surfels, one_in)
    rays = orc.generate_rays(sub, None, opt, atlas)
    def walk(nodes2, label):
        n4, lt = orc.bvh4_build(nodes2, pairs=True)
        _, frame = orc.bvh2q_build(nodes2)
        n4q = orc.bvh4q_build(n4, frame)
        a, h = orc.walk4q(n4q, frame, lt, rays, True)
        r = {"label": label, "inner": round(a.inner_visits / a.rays, 3), "tri": round(a.leaf_visits / a.rays, 3), "hits": a.hits, "model": round(70 * a.inner_visits / a.rays + 64 * a.leaf_visits / a.rays, 1), "nodes4": len(n4q)}
        print(json.dumps(r), flush=True); return r
    ploc, _ = orc.bvh_build(tris, 1, 1)
    base = walk(ploc, 'ploc')
    lbvh, _ = orc.bvh_build(tris, 0, 1)
    walk(lbvh, 'lbvh')
    for K in (128, 512, 2048):
        t = time.time()
        nodes, nr = treelet_sah(lbvh, K)
        r = walk(nodes, f'lbvh top + SAH treelets <= {K} ({nr} treelets, {time.time()-t:.0f} s)')
        print('   vs ploc', round(r['model'] / base['model'], 4), flush=True)
    for K in (16, 64, 512, 4096):
        nodes, nr = treelet_sah(ploc, K)
        r = walk(nodes, f'PLOC top + SAH treelets <= {K} ({nr} treelets)')
        print('   vs ploc', round(r['model'] / base['model'], 4), flush=True)
if __name__ == "__main__":
    main()

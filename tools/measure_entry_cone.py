#!/usr/bin/env python3
"""tools/measure_entry_cone.py (a measurement script: the oracle and the CPU-run horizon kernel as counters; not a test, not product code) — VERDICT r5 item
1(c), the entry phase under horizon tables: the rays a unit still casts are the ones near the horizon, so the entries of its tile's cut that lie wholly inside
the STEEP cone around every surfel's normal — steeper than any cast ray — could be culled from the part's record.  Counts, on BASELINE configs[2], the entries
per part after the wedge culling of k_tile_cuts and after that cone culling.  Result (profiles/r06/entry_phase_cone_culling.json): 23.0 -> 23.0 — there are
no entries up there: where the rays are not cast there is no geometry, hence no boxes, either.  Not built."""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import horizon_emu as he
from aobaker_amd import capi, scenes
from aobaker_amd.scene import FlatScene
from oracle import oracle as orc
atlas=2048
flat = FlatScene(scenes.terrain_scene(grid=708, atlas=atlas, seed=3))
opt = orc.default_options(); opt.sample_count, opt.sampler, opt.accumulate, opt.normal_fp16, opt.seed = 256, capi.SAMPLER_COSINE, capi.ACCUMULATE_FLOAT, capi.FP16_OFF, 3
tris = orc.scene_triangles(flat.desc)
surfels,_ = orc.rasterize(flat.desc, 0, opt)
nodes2 = orc.bvh_build(tris, 2, 1)[0]
n4, lt = orc.bvh4_build(nodes2, pairs=True); _, frame = orc.bvh2q_build(nodes2); n4q = orc.bvh4q_build(n4, frame)
grid = surfels.reshape(atlas, atlas); tw = atlas//8
picks = np.random.default_rng(7).choice(tw*tw, size=24, replace=False)
origin, scale = frame[:3].astype(np.float64), frame[3:].astype(np.float64)
tot = dict(cut=0, wedge=0, wedge_cone=0, parts=0)
for t_ in picks:
    sub = np.ascontiguousarray(grid[8*(t_//tw):8*(t_//tw)+8, 8*(t_%tw):8*(t_%tw)+8].reshape(-1))
    tab = he.tables(sub, n4q, lt, frame, 256, mid=3.0, small=0.14, big=0.6)
    free = np.minimum(tab[:, :16].astype(np.int64), 16)
    pos, nrm = sub["position"].astype(np.float64), sub["normal"].astype(np.float64)
    pmin, pmax = pos.min(0), pos.max(0); pull = 0.25*(pmax-pmin)
    cut = orc.tile_cut_reach(n4q, frame, (pmin+pull).astype(np.float32), (pmax-pull).astype(np.float32), pmin.astype(np.float32), pmax.astype(np.float32), 0.0, 32)
    lo = origin + (cut[:,1:4] & 0x7fff)*scale; hi = origin + ((cut[:,1:4]>>16) & 0x7fff)*scale
    E = len(cut)
    Xf = np.stack([-nrm[:,2], np.zeros(64), nrm[:,0]],1); Xf/=np.linalg.norm(Xf,axis=1,keepdims=True)
    Zf = np.cross(Xf,nrm); Zf/=np.linalg.norm(Zf,axis=1,keepdims=True)
    eps = 1e-4*32767*scale.max()
    for part in range(4):
        f0,f1 = part/4.0,(part+1)/4.0
        fmin = free[:, 4*part:4*part+4].min(1)            # per surfel: its steepest cast column in this part
        cmax = np.sqrt(np.maximum(0.0, 1.0 - fmin/16.0))  # cos(theta) of every cast ray of the surfel <= cmax
        wedge = np.zeros(E,bool); cone = np.zeros(E,bool)
        for s_ in range(64):
            P,N,X,Z = pos[s_],nrm[s_],Xf[s_],Zf[s_]
            r0,r1 = (lo-eps)-P,(hi+eps)-P
            most = lambda v: np.maximum(r0*v,r1*v).sum(1); least=lambda v: np.minimum(r0*v,r1*v).sum(1)
            ok = most(N) >= 0
            u0 = np.cos(2*np.pi*f0)*X+np.sin(2*np.pi*f0)*Z; u1 = np.cos(2*np.pi*f1)*X+np.sin(2*np.pi*f1)*Z
            ok &= (most(np.cross(u0,N))>=0) & (least(np.cross(u1,N))<=0)
            wedge |= ok
            # the box lies wholly inside the steep cone around N (every point's cos > cmax): unreachable for the cast rays
            amin = least(N/np.linalg.norm(N)); rmax = np.sqrt(np.maximum(r0**2, r1**2).sum(1))
            steep = (amin > 0) & (amin > (cmax[s_]+1e-3)*rmax)
            lists_empty = (free[s_,4*part:4*part+4] >= 16).all()
            cone |= ok & ~steep & ~lists_empty
        tot["cut"]+=E; tot["wedge"]+=wedge.sum(); tot["wedge_cone"]+=cone.sum(); tot["parts"]+=1
p=float(tot["parts"]); print({k: round(v/p,2) for k,v in tot.items() if k!="parts"})

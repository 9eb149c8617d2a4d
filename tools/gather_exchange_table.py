#!/usr/bin/env python3
"""What one gather instruction of the last stage's HASHED levels asks of the texture path under different lane arrangements (CPU only; bench
camera, 800x800, [128], fp16 rows): distinct 128-byte lines / 64-byte sectors per instruction, and the same summed over the 16 quads of
4 lanes the address unit works through -- for the shipping x-corner exchange between the wave's halves (lines, sect, qlines, qsect) and for
an exchange between neighbouring lanes (alines, asect, aqsect), with the tile's pixels in row-major, Morton or 2x2-quad lane order.
Measured beside it (profiles/r06/xswap_ab.txt): the neighbour exchange has 13 % fewer quad-line pairs and 40 % more lines per instruction
and is 3.5 % SLOWER -- the cost of a gather follows the distinct lines of the whole instruction (TA busy ~25 cycles per wave-gather at
8.5 lines; tools/ubench/gathers.hip: +2.3 cycles per line beyond 8), which the half-wave exchange already halves."""
import os, sys
import numpy as np
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle as orc
from helpers import GRIDS, oracle_cfg, synthetic_params
from sanerf_hq_amd import synth
from gather_lines_table import rows_of
steps=[128]; H=W=800; ntile=12
params=synthetic_params(steps, seed=0); cfg=oracle_cfg(orc, params, steps)
pose=synth.orbit_pose(1.0,20.0,30.0); fx,fy=synth.pinhole_intrinsics(H,W)[:2]
ro,rd=orc.generate_rays(pose,fx,fy,W/2.0,H/2.0,H,W)
rng=np.random.default_rng(0)
ty=rng.integers(0,H//8,ntile); tx=rng.integers(0,W//8,ntile)
def tile_idx(order):
    out=[]
    for y,x in zip(ty,tx):
        l=np.arange(64)
        if order=='row': px=l&7; py=l>>3
        elif order=='morton':
            px=(l&1)|((l>>1)&2)|((l>>2)&4); py=((l>>1)&1)|((l>>2)&2)|((l>>3)&4)
        elif order=='quad2x2':   # quads 2x2, quads row-major
            q=l>>2; i=l&3; px=(q&3)*2+(i&1); py=(q>>2)*2+(i>>1)
        out.append((y*8+py)*W+(x*8+px))
    return np.concatenate(out)
bound=float(cfg.bound)
def positions(idx):
    out=orc.render(cfg, ro[idx], rd[idx], debug=True)
    rb=out["real_bins0"].astype(np.float32)
    tmid=(rb[:,1:]+rb[:,:-1])/np.float32(2)
    p=ro[idx][:,None,:]+rd[idx][:,None,:]*tmid[...,None]
    z=orc.contract(p.reshape(-1,3).astype(np.float32)).reshape(p.shape)
    return (z+bound)/(2*bound)
g=GRIDS["grid"]
offs,pls=orc.grid_layout(3,g["num_levels"],g["level_dim"],2,16,g["log2_hashmap_size"],g["desired_resolution"])
L=g["num_levels"]; res=orc.level_resolutions(L,float(np.log2(pls)),16)
def dcount(a):  # a [..., lanes] -> mean distinct along last axis
    s=np.sort(a,axis=-1); return (1+(s[...,1:]!=s[...,:-1]).sum(-1))
for order in ('row','morton','quad2x2'):
    idx=tile_idx(order); x=positions(idx); T=x.shape[1]
    tot={k:0.0 for k in ('lines','sect','qlines','qsect','alines','asect','aqsect')}
    for l in range(7,16):
        r=int(res[l]); size=int(offs[l+1]-offs[l])
        rows,_,dense=rows_of(x,r,size,int(offs[l]))      # [n*64? , T, 8]
        rows=rows.reshape(ntile,64,T,8).astype(np.int64)*4     # byte offsets, fp16 rows 4 B
        # XSWAP (half-wave): instr A for pair q: lanes0-31 -> own corner 2q; lanes 32-63 -> lower lanes' corner 2q+1. instr B: lanes 0-31 -> upper's corner 2q, upper own 2q+1
        for q in range(4):
            a=np.concatenate([rows[:,:32,:,2*q],rows[:,:32,:,2*q+1]],axis=1)   # [tile,64,T]
            b=np.concatenate([rows[:,32:,:,2*q],rows[:,32:,:,2*q+1]],axis=1)
            for ins in (a,b):
                v=ins.transpose(0,2,1)   # [tile,T,64]
                tot['lines']+=dcount(v//128).mean(); tot['sect']+=dcount(v//64).mean()
                vq=v.reshape(ntile,T,16,4)
                tot['qlines']+=dcount(vq//128).sum(-1).mean(); tot['qsect']+=dcount(vq//64).sum(-1).mean()
            # adjacent-lane exchange: instr A: even lanes own corner 2q of ray l, odd lanes: corner 2q+1 of ray l-1 ; instr B: even lanes: corner 2q of ray l+1, odd: own 2q+1
            ev=rows[:,0::2]; od=rows[:,1::2]
            a2=np.stack([ev[...,2*q],ev[...,2*q+1]],axis=2).reshape(ntile,64,T)   # interleave: lane 2i -> ray 2i corner x ; lane 2i+1 -> ray 2i corner x+1
            b2=np.stack([od[...,2*q],od[...,2*q+1]],axis=2).reshape(ntile,64,T)
            for ins in (a2,b2):
                v=ins.transpose(0,2,1)
                tot['alines']+=dcount(v//128).mean(); tot['asect']+=dcount(v//64).mean()
                tot['aqsect']+=dcount(v.reshape(ntile,T,16,4)//64).sum(-1).mean()
    n=72.0
    print(order, {k: round(v/n,2) for k,v in tot.items()})

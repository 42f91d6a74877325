#!/usr/bin/env python
"""One 16 Mi-ray launch (4096 x 4096 primary rays of the atrium camera) against the oracle on every 251st ray:
the default kernel far above the benchmark's launch size."""
import sys, numpy as np
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from oracle import binding as O
from rodent_amd import abi, formats as F, raygen, scenes
path=scenes.scene_bvh("atrium")
bvh=abi.DeviceBvh.load(path,2,0)
nodes,tris=F.read_bvh(path,F.BVH2_TRI1)
eye,d,up,fov=scenes.CAMERAS["atrium"]
rays=raygen.primary_rays(eye,d,up,fov,4096,4096,0.0,5000.0)
n=len(rays); rd=abi.to_device(rays,0); hd=torch.full((n*16,),0xFF,dtype=torch.uint8,device="cuda:0")
for any_hit in (False,True):
    abi.traverse_async(bvh,rd,hd,n,any_hit,0); torch.cuda.synchronize()
    got=abi.from_device(hd,F.HIT1)
    idx=np.arange(0,n,251)
    ref,_=O.traverse(2,nodes,tris,rays[idx],any_hit=any_hit)
    if any_hit: ok=np.array_equal(got[idx]["tri_id"]>=0, ref["tri_id"]>=0)
    else: ok=got[idx].tobytes()==ref.tobytes()
    print("16Mi rays any=%d: sample of %d bit-exact: %s; hits %d"%(any_hit,len(idx),ok,(got["tri_id"]>=0).sum()))

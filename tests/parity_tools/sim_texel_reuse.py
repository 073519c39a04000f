"""Texel reuse of the plane-gradient scatter on config-2 geometry (CPU, uses the oracle's sample depths): tap hits / distinct texels
for several aggregation windows.  Backs the table in DESIGN.md section 2.1b.   python tests/parity_tools/sim_texel_reuse.py (test infrastructure: it uses the CPU oracle)"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT, os.path.join(ROOT, 'tests')]
from oracle import render_ref as R
torch.manual_seed(0)
N,res,S,F,PW=1,128,48,48,256
import math
from test_gpu_parity import _random_scene
planes, dec, o, d, nc, nf = _random_scene(0, N=N, res=res, S=S, F=F, hw=(PW,PW), scale=1.0)
# take a band of 4x4 tiles: rows 60..63, all columns -> 32 tiles
idx=[]
for ty in (15, 2):
  for tx in range(0,32,3):
    for rr in range(16):
        idx.append((ty*4+rr//4)*res + tx*4 + rr%4)
idx=torch.tensor(idx)
opts=dict(depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, clamp_mode='softplus')
st={}
R.render(planes, dec, o[:,idx], d[:,idx], opts, nc[:,idx], nf.reshape(N,res*res,F)[:,idx].reshape(-1,F), stages=st)
tc=st['depths_coarse'].reshape(-1,S).numpy(); tf=st['depths_fine'].reshape(-1,F).numpy()
oo=o[0,idx].numpy(); dd=d[0,idx].numpy()
def taps(t):  # t [R,K] -> keys [R,K,12]
    p=oo[:,None,:]+t[:,:,None]*dd[:,None,:]
    p=p*2.0
    keys=[]
    for pl,(a,b) in enumerate(((0,1),(0,2),(2,0))):
        ix=((p[...,a]+1)*PW-1)/2; iy=((p[...,b]+1)*PW-1)/2
        x0=np.floor(ix).astype(int); y0=np.floor(iy).astype(int)
        for dy in (0,1):
            for dx in (0,1):
                xx=np.clip(x0+dx,0,PW-1); yy=np.clip(y0+dy,0,PW-1)
                keys.append(pl*PW*PW+yy*PW+xx)
    return np.stack(keys,-1)
kc=taps(tc); kf=taps(np.sort(tf,1))   # fine sorted per ray
kf_uns=taps(tf)
nt=len(idx)//16
def analyse(name, per_ray_keys_list, chunk):
    tot_hits=0; tot_dist=0; per_tile=[]
    for t in range(nt):
        hits=0; dist=0
        for K in per_ray_keys_list:
            Kt=K[t*16:(t+1)*16]   # [16, n, 12]
            n=Kt.shape[1]
            for c0 in range(0,n,chunk):
                ch=Kt[:,c0:c0+chunk].reshape(-1)
                hits+=ch.size; dist+=len(np.unique(ch))
        tot_hits+=hits; tot_dist+=dist
    print(name, 'chunk',chunk,'hits',tot_hits,'flushes',tot_dist,'reduction %.2f'%(tot_hits/tot_dist))
for chunk in (4,8,16,48):
    analyse('coarse+fine(sorted)', [kc,kf], chunk)
    analyse('coarse+fine(unsorted)', [kc,kf_uns], chunk)
# merged order chunks
tall=np.concatenate([tc,tf],1); tall=np.sort(tall,1)
ka=taps(tall)
for chunk in (8,16,32,96):
    analyse('merged sorted', [ka], chunk)
# per-ray only (no cross-ray): reduction within a ray
hits=ka.reshape(len(idx),-1); print('per-ray reduction %.2f'% (hits.size/sum(len(np.unique(h)) for h in hits)))
# per plane: how much of the reduction each plane contributes at 16-rank chunks, and what keeping a plane's table across the whole ray would give
for pl, name in enumerate(('plane 0 (x,y)', 'plane 1 (x,z)', 'plane 2 (z,x)')):
    kp = ka[..., 4 * pl:4 * pl + 4]
    for chunk in (16, 96):
        flushes = 0
        for t in range(nt):
            Kt = kp[t * 16:(t + 1) * 16]
            for c0 in range(0, Kt.shape[1], chunk):
                flushes += len(np.unique(Kt[:, c0:c0 + chunk].reshape(-1)))
        print(name, 'chunk', chunk, 'hits', kp.size, 'flushes', flushes, 'reduction %.2f' % (kp.size / flushes))

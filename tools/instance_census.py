#!/usr/bin/env python3
"""How many (Gaussian, tile) instances does each membership rule keep at config 2?  CPU only: the oracle's preprocess
gives the reference's rects, centres and conics; counted are the reference's per-tile instances, the 2 x 1 super-tile
lists, both again with the rect cut down to the bounding box of the ellipse alpha >= 1/255 (what inference frames
store: preprocess.hip), and with per-tile-row spans of that ellipse (not built)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
np.seterr(all="ignore")
from gsworld_amd import scenes
from oracle import gs_oracle as go
raw = scenes.tabletop_scene("xarm6_align", seed=1)
cam = scenes.sensor_camera("xarm6_align")
means, shs, op, sc, rot = (t.numpy() for t in raw.activated())
st = go.Settings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy)
e = np.zeros(0, np.float32)
o = go.preprocess(st, means, shs, None, op.reshape(-1), sc, rot, None, cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), cam.camera_center.numpy())
vis = o["tiles_touched"] > 0
r = o["rects"][vis].astype(np.int64); c = o["means2D"][vis].astype(np.float64); q = o["conic_opacity"][vis].astype(np.float64)
V = vis.sum(); print("V", V, "ref instances", int(((r[:,2]-r[:,0])*(r[:,3]-r[:,1])).sum()))
A, B, C_, opa = q[:,0], q[:,1], q[:,2], q[:,3]
k = 2*np.log(255*opa)
det = A*C_ - B*B
safe = (A>0)&(C_>0)&(B*B<=0.999*A*C_)
hx = np.sqrt(np.maximum(k,0)*C_/det)+0.25; hy = np.sqrt(np.maximum(k,0)*A/det)+0.25
t = r.copy()
t[:,0] = np.maximum(r[:,0], np.floor((c[:,0]-hx)/16)); t[:,1] = np.maximum(r[:,1], np.floor((c[:,1]-hy)/16))
t[:,2] = np.minimum(r[:,2], np.floor((c[:,0]+hx)/16)+1); t[:,3] = np.minimum(r[:,3], np.floor((c[:,1]+hy)/16)+1)
t[~safe] = r[~safe]
area = np.maximum(t[:,2]-t[:,0],0)*np.maximum(t[:,3]-t[:,1],0); area[k<=0] = 0
print("AABB tiles", int(area.sum()))
def sup(t):  # 2x1 super-tiles
    return (np.maximum((t[:,2]+1)//2 - t[:,0]//2, 0) * np.maximum(t[:,3]-t[:,1],0))
a2 = sup(t); a2[(k<=0)|(area==0)] = 0
print("ref super", int(sup(r).sum()), "AABB super", int(a2.sum()))
# per-row spans (tile rows), and exact per-tile test, for the AABB-surviving Gaussians
rows_t = 0; rows_s = 0; exact_t = 0; exact_s = 0
idx = np.nonzero((area>0))[0]
for i in idx:
    x0,y0,x1,y1 = t[i]
    cx, cy = c[i]; a,b,cc,kk = A[i],B[i],C_[i],k[i]
    if not safe[i]:
        rows_t += (x1-x0)*(y1-y0); exact_t += (x1-x0)*(y1-y0); rows_s += ((x1+1)//2-x0//2)*(y1-y0); exact_s += ((x1+1)//2-x0//2)*(y1-y0); continue
    for ty in range(y0, y1):
        ylo, yhi = 16*ty - cy, 16*ty+15 - cy   # dy range of pixel centres (pixel - centre)
        # x-extent of ellipse a dx^2 + 2 b dx dy + c dy^2 <= k over dy in [ylo,yhi]
        dys = [ylo, yhi]
        # extreme points in x at dy = -b dx / cc -> dx = +-sqrt(k cc / det)
        xe = np.sqrt(kk*cc/(a*cc-b*b)); 
        lo, hi = np.inf, -np.inf
        for s in (-1, 1):
            dye = -b*(s*xe)/cc
            if ylo <= dye <= yhi:
                lo = min(lo, s*xe); hi = max(hi, s*xe)
        for dy in dys:
            disc = b*b*dy*dy - a*(cc*dy*dy - kk)
            if disc >= 0:
                s_ = np.sqrt(disc); lo = min(lo, (-b*dy - s_)/a); hi = max(hi, (-b*dy + s_)/a)
        if hi < lo: continue
        tx0 = max(x0, int(np.floor((cx+lo-0.25)/16))); tx1 = min(x1, int(np.floor((cx+hi+0.25)/16))+1)
        if tx1 > tx0:
            rows_t += tx1-tx0; rows_s += (tx1+1)//2 - tx0//2
print("per-row-span tiles", rows_t, "super", rows_s)

import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
import oracle
from yak_amd import capi, synth
from gpu_helpers import gpu_compute_dists
ctx = capi.Context(0)
for name in ['s0','s1']:
    depth,_,_ = getattr(synth,'scene_'+name)()
    want = oracle.compute_dists(depth, synth.INTR)
    got = gpu_compute_dists(ctx, depth, synth.INTR)
    bad = np.argwhere(got!=want)
    print(name, 'mismatches', len(bad), 'of', got.size)
    fx,fy,cx,cy=(np.float32(v) for v in synth.INTR)
    for (y,x) in bad[:8]:
        xl=(np.float32(x)-cx)*(np.float32(1)/fx); yl=(np.float32(y)-cy)*(np.float32(1)/fy)
        lam=np.sqrt(xl*xl+yl*yl+np.float32(1)).astype(np.float32)
        p=np.float32(depth[y,x])*lam
        q32=np.float32(p*np.float32(0.001))
        q64=np.float64(p)*np.float64(np.float32(0.001))
        print(y,x,'got',hex(got[y,x]),'want',hex(want[y,x]),'double-round',hex(np.float16(q32).view(np.uint16)),'single-round',hex(np.float16(q64).view(np.uint16)), float(q64))

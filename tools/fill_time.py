import sys, time, numpy as np, ctypes as C
sys.path[:0]=['/root/repo','/root/repo/body-and-organ-analysis_amd']
from boa_hip.device import Context
from boa_hip._lib import check
from scipy import ndimage
c=Context(0)
rng=np.random.default_rng(0)
shape=(154,512,512)
sm=ndimage.gaussian_filter(rng.standard_normal(shape),1.0)
for thr,name in ((0.0,'half'),(0.15,'sparse'),(-0.15,'dense')):
    m=(sm>thr)
    n=m.size
    d_m=c.from_numpy(m.astype(np.uint8)); d_i=c.alloc(n*4); d_t=c.alloc(n); d_o=c.alloc(n)
    for it in range(3):
        c.sync(); t0=time.perf_counter()
        check(c.lib.boa_fill_holes_2d(c.h,d_m.vp,shape[0],shape[1],shape[2],d_i.vp,d_t.vp,d_o.vp))
        c.sync(); t1=time.perf_counter()
    out=d_o.download(shape,np.uint8).astype(bool)
    ref=np.stack([ndimage.binary_fill_holes(m[i]) for i in range(0,shape[0],17)])
    print(name,'fill_holes %.3f ms'%((t1-t0)*1e3), 'ok', np.array_equal(out[::17],ref))
    for d in (d_m,d_i,d_t,d_o): d.free()
# pure noise
m=rng.random(shape)<0.5
d_m=c.from_numpy(m.astype(np.uint8)); n=m.size; d_i=c.alloc(n*4); d_t=c.alloc(n); d_o=c.alloc(n)
for it in range(3):
    c.sync(); t0=time.perf_counter(); check(c.lib.boa_fill_holes_2d(c.h,d_m.vp,shape[0],shape[1],shape[2],d_i.vp,d_t.vp,d_o.vp)); c.sync(); t1=time.perf_counter()
print('noise fill_holes %.3f ms'%((t1-t0)*1e3))
c.close()

import sys, ctypes as C, numpy as np, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(),'tests'))
import _libs as L
from raytracing_amd import scenes, host, gpu, scene_types as T
print("devices", gpu.lib().rt_gpu_device_count(), flush=True)
# math
k = np.arange(0, 1<<23, 7, dtype=np.uint32)
u = k.astype(np.float32) * np.float32(2.0**-23)
a = (2.0*np.pi*u.astype(np.float64)).astype(np.float32)
m = gpu.eval_math(a, None)
s = np.zeros_like(a); c = np.zeros_like(a)
L.oracle().orc_sincosf(a.ctypes.data, s.ctypes.data, c.ctypes.data, a.size)
print("sin same", L.same_bits(m['sin'], s).all(), "cos same", L.same_bits(m['cos'], c).all())
lg = np.zeros_like(u); L.oracle().orc_logf(u.ctypes.data, lg.ctypes.data, u.size)
m2 = gpu.eval_math(u, None)
print("log same", L.same_bits(m2['log'], lg).all(), "sqrt same", L.same_bits(m2['sqrt'], np.sqrt(u)).all())
rng = np.random.default_rng(1)
x = rng.standard_normal(1<<20).astype(np.float32) * np.float32(1e3); y = rng.standard_normal(1<<20).astype(np.float32)
y[:8] = [0, -0.0, np.inf, -np.inf, np.nan, 1e-45, 1e38, -1e-38]; x[:8]=[1,1,1,np.inf,1,1,1e38,1e38]
with np.errstate(all='ignore'):
    print("div same", L.same_bits(gpu.eval_math(x, y)['div'], x/y).all())
for name in ("deux_spheres","sphere_room","dog_room","three_dogs"):
    sh = scenes.CONFIGS[name]()
    W=H=64; spp=4
    t=time.time(); g = gpu.GpuScene(sh); tc=time.time()-t
    img, arr = host.new_image(W,H)
    t=time.time(); g.render_region(img, (0,0,W,H), spp, 1234, gpu.Options(flags=gpu.FLAG_DETAILED_STATS)); tr=time.time()-t
    gst = g.stats()
    img2, arr2 = host.new_image(W,H)
    st = L.Stats()
    reg = gpu.rect(0,0,W,H)
    L.oracle().orc_render_region_counter(sh.ptr, C.byref(img2), C.byref(reg), spp, 0, 1234, 1, 64, C.byref(st))
    eq = L.same_bits(arr, arr2)
    print(name, "create %.3fs render %.3fs"%(tc,tr), "counter parity:", eq.all(), (~eq).sum(), "mean", arr.mean(), arr2.mean(), flush=True)
    ost = st.as_dict()
    print("   gpu", {k:gst[k] for k in ost}, "\n   orc", ost, flush=True)
    # AES replay
    jobs=[]; img3, arr3 = host.new_image(W,H); img4, arr4 = host.new_image(W,H)
    tile=8
    for ty in range(H//tile):
        for tx in range(W//tile):
            jobs.append(L.make_job(sh, img3, (tx*tile,ty*tile,tile,tile), 2, L.serie_from_seed(100+ty*8+tx)))
    g.render_jobs(jobs)
    for j in jobs:
        j.image = C.pointer(img4)
        L.oracle().orc_render_region_aes(C.byref(j), 1, 64, None)
    eq = L.same_bits(arr3, arr4)
    print("   aes replay parity:", eq.all(), (~eq).sum(), flush=True)
    g.close()

import importlib, sys, numpy as np
sys.path.insert(0,'/root/repo')
pkg = importlib.import_module("srmeetsps-cuda_amd"); pkg.load()
for (h,w,sf,n,kind) in [(2048,2048,4,20,"full"),(1024,1024,4,40,"ellipse"),(512,512,2,23,"ellipse")]:
    sc = pkg.synth.make_scene(h,w,sf,n,seed=7,mask_kind=kind); dh = pkg.DataHandler.from_scene(sc)
    ref=None
    for it in range(12):
        ctx = pkg.Context(device_id=0); ctx.setup(dh)
        en = ctx.execute(0); z = ctx.get("z"); s = ctx.get("s"); ctx.close()
        if ref is None: ref=(list(en),z,s)
        else: assert list(en)==ref[0] and np.array_equal(z,ref[1]) and np.array_equal(s,ref[2]), (h,w,n,it)
    print(h,w,n,kind,"12 whole solves bit-identical,",len(ref[0]),"passes")

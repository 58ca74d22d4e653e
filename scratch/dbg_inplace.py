import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import test_gpu_stress as st
from tensorflowraytrace_amd import ops, _lib
DEV="cuda:0"
seed=int(sys.argv[1]) if len(sys.argv)>1 else 35
flags=_lib.COMPILE_ACTIVE | _lib.COMPILE_FINISHED | _lib.COMPILE_DEAD | _lib.COMPILE_STOPPED
sc0=st._soup(seed)
fv=sc0["P"].to(DEV)
eps=[(1e-10,1e-10,1e-10),(1e-10,1e-3,1e-7),(1e-10,0.2,-0.01)][seed%3]
base=dict(n_in=sc0["n_in"].to(DEV), n_out=sc0["n_out"].to(DEV))
for dtype in (torch.float64, torch.float32):
    r=sc0["rays"].to(DEV).to(dtype)
    order=ops.ray_order(r)
    outs={}
    for ip in (False, True):
        args=ops.Scene3DArgs(fv, sc0["cat"].int().to(DEV), cluster_order=ops.cluster_order(fv), coherent_rays=True, **base)
        args.eps=eps; args.coherent_only=True; args.in_place=ip
        outs[ip]=ops.trace3d(r[:,order.long()].contiguous(), fv, args, max_passes=4, flags=flags, new_ray_length=sc0["L"], dead_ray_length=0.5 if seed%2 else None)
    a,b=outs[True],outs[False]
    print(dtype, "counts equal", np.array_equal(a["counts"], b["counts"]))
    print(a["counts"]); print(b["counts"])
    for cls in ("finished","active","dead","stopped","unfinished"):
        for key in (cls, cls+"_id", cls+"_face"):
            if key not in a: continue
            x,y=a[key],b[key]
            if x.shape!=y.shape: print(key,"shape",x.shape,y.shape); continue
            if not torch.equal(x,y):
                d=(x!=y)
                if d.dim()==2: d=d.any(0)
                idx=d.nonzero().flatten()[:10]
                print(key,"differs at",idx.tolist(), "inplace", x[...,idx].tolist(), "perpass", y[...,idx].tolist())
    plain=ops.Scene3DArgs(fv, sc0["cat"].int().to(DEV), **base); plain.eps=eps
    ref=ops.trace3d(r, fv, plain, max_passes=4, flags=flags, new_ray_length=sc0["L"], dead_ray_length=0.5 if seed%2 else None)
    grp=ops.Scene3DArgs(fv, sc0["cat"].int().to(DEV), cluster_order=ops.cluster_order(fv), **base); grp.eps=eps
    refg=ops.trace3d(r, fv, grp, max_passes=4, flags=flags, new_ray_length=sc0["L"], dead_ray_length=0.5 if seed%2 else None)
    for nm,o in (("perpass",ops.restore_order(b,order)),("inplace",ops.restore_order(a,order)),("group",refg)):
        for cls in ("finished","active","dead","stopped"):
            x,y=o[cls+"_face"],ref[cls+"_face"]
            if not torch.equal(x,y):
                idx=(x!=y).nonzero().flatten()[:10]
                print(nm, cls,"face differs at",idx.tolist(), nm, x[idx].tolist(), "allpairs", y[idx].tolist(), "ids", ref[cls+"_id"][idx].tolist())
                j=int(idx[0]); f1,f2=int(x[j]),int(y[j])
                print(" faces", fv[f1].tolist(), fv[f2].tolist())
                print(" rays equal:", torch.equal(o[cls], ref[cls]))

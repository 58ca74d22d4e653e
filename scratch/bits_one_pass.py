"""Which components of a one-pass trace differ from the oracle, and by how many ulps."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import test_gpu_stress as T
for seed in (3, 35):
    sc = T._soup(seed)
    out = T._gpu_trace(sc, 1); ref = T._oracle_trace(sc, 1)
    for name, got, want in (("active", out["active"].cpu(), T._block(ref["active"])),
                            ("finished", out["finished"].cpu(), T._block(ref["finished"])),
                            ("child", out["unfinished"].cpu(), T._block(ref["unfinished"]))):
        g, w = got.numpy(), want.numpy()
        ulp = np.abs(g.view(np.int64) - w.view(np.int64))
        print(seed, name, "rows differing per component:", (ulp > 0).sum(1), "max ulp", ulp.max(1))
    # faces of the active rays and normals
    face = out["active_face"].cpu().long()
    P = sc["P"][face]
    from oracle import tracer
    f = tracer.faces_from_vertices(P.reshape(-1, 3), torch.arange(P.shape[0] * 3).reshape(-1, 3))
    # device normal through build_faces
    from tensorflowraytrace_amd import ops
    fv, nrm = ops.build_faces(P.reshape(-1, 3).to("cuda:0"), torch.arange(P.shape[0] * 3, dtype=torch.int32).reshape(-1, 3).to("cuda:0"))
    d = (nrm.cpu() != f["norm"]).any(1).sum()
    print(seed, "face normals differing (build_faces vs oracle):", int(d), "of", P.shape[0])

import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import apex_solver_amd as pkg
from apex_solver_amd import capi
d = pkg.synthetic.make_named(sys.argv[1] if len(sys.argv) > 1 else "final-13682", 1.0)
for rep in range(2):
    t = time.time()
    hs = capi.host_structure(d.n_cam, d.n_pt, d.cam_idx, d.pt_idx)
    print({k: round(v, 3) for k, v in hs.items() if k.startswith("s_")}, round(time.time() - t, 3), int(hs["tiles"]), int(hs["etree_levels"]))

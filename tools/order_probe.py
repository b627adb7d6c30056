import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
order = sys.argv[1]
import __graft_entry__ as ge
if order == "torch_first":
    import torch
    print("torch sees", torch.cuda.device_count(), torch.cuda.is_available())
    x = torch.ones(4, device="cuda"); print(x.sum().item())
pkg = ge.load_package()
lbm = pkg.BinaryLBM(16, 16, 16)
lbm.LBM_init_stripe(0.5); lbm.LBM_timestep(3)
print("lib ok", lbm.mass())
if order == "lib_first":
    import torch
    print("torch sees", torch.cuda.device_count())
    try:
        x = torch.ones(4, device="cuda"); print(x.sum().item())
    except Exception as e:
        print("torch cuda failed:", str(e)[:100])
import oracle_binding as ob
ref = ob.OracleLattice(16,16,16); ref.init_stripe(0.5)
for _ in range(3): ref.timestep()
f,g = lbm.populations(); print("parity", np.array_equal(f, ref.f))
maps = open("/proc/self/maps").read()
print(sorted(set(l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l or "libhsa-runtime" in l)))

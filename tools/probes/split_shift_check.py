import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch, cases
from infiniteexamodels.jl_amd.model import ExaModel
from pyoracle import OracleModel
for name in ("quadrotor_1000", "quadrotor_100", "pandemic_300x7"):
    core = cases.build_core(name); blob = core.to_blob(); om = OracleModel(blob)
    x, y = cases.eval_point_for(name, om)
    gm = ExaModel(core, device=0, blob=blob, options={"split_shift": 1, "split_small": 0})
    got = gm.jac_coord(torch.tensor(x, device="cuda")).cpu().numpy()
    print(name, "max diff", np.abs(got - om.jac_coord(x)).max())
    gm.close()

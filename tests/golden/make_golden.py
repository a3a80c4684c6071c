#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/.

The reference cannot be executed in the build container (no Julia; ExaModels.jl is not
vendored), so these vectors come from the build's own CPU oracle (oracle/iem_oracle.c)
after it has been cross-checked against torch autograd (tests/test_oracle_autodiff.py) and
the reference's solver-level constants (tests/test_known_answers.py).  They pin the
oracle against regressions and give the GPU tests a fixed target.

    python tests/golden/make_golden.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np

import cases
from pyoracle import OracleModel

CASES = ["quadrotor_5", "quadrotor_oc3_40", "quadrotor_100", "pandemic_20x3", "farmer_5", "opf_7", "ode_5x5", "test_problem_1",
         "rosenbrock", "pfun", "operator_zoo", "irregular"]


def main():
    for name in CASES:
        core = cases.build_core(name)
        om = OracleModel(core.to_blob())
        x, y = cases.eval_point_for(name, om, seed=11)
        jr, jc = om.jac_structure(1)
        hr, hc = om.hess_structure(1)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            meta=np.array([om.nvar, om.ncon, om.npar, om.nnzj, om.nnzh]), x=x, y=y, obj_weight=0.7,
            obj=om.obj(x), cons=om.cons(x), grad=om.grad(x), jac_rows=jr.astype(np.int32), jac_cols=jc.astype(np.int32),
            jac_vals=om.jac_coord(x), hess_rows=hr.astype(np.int32), hess_cols=hc.astype(np.int32),
            hess_vals=om.hess_coord(x, y, 0.7), x0=om.x0, lvar=om.lvar, uvar=om.uvar, lcon=om.lcon,
            ucon=om.ucon, theta=om.theta)
        print(name, om.nvar, om.ncon, om.nnzj, om.nnzh)


if __name__ == "__main__":
    main()

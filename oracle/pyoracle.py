"""ctypes wrapper of the CPU oracle (``oracle/liboracle.so``).

TEST INFRASTRUCTURE ONLY — imported by ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg, never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "iem_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, i64, dbl = C.c_void_p, C.c_int64, C.c_double
        pd = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
        pi = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
        L.orc_create.restype = vp
        L.orc_create.argtypes = [C.c_char_p, C.c_size_t]
        L.orc_destroy.argtypes = [vp]
        L.orc_meta.argtypes = [vp, pi]
        L.orc_template_info.argtypes = [vp, i64, pi]
        L.orc_get_array.argtypes = [vp, C.c_int, pd]
        L.orc_set_parameter.argtypes = [vp, i64, i64, pd]
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_max_threads.restype = C.c_int
        L.orc_obj.restype = dbl
        L.orc_obj.argtypes = [vp, pd]
        L.orc_cons.argtypes = [vp, pd, pd]
        L.orc_grad.argtypes = [vp, pd, pd]
        L.orc_jac_structure.argtypes = [vp, pi, pi, C.c_int]
        L.orc_jac_coord.argtypes = [vp, pd, pd]
        L.orc_hess_structure.argtypes = [vp, pi, pi, C.c_int]
        L.orc_hess_coord.argtypes = [vp, pd, pd, dbl, pd]
        L.orc_jprod.argtypes = [vp, pd, pd, pd]
        L.orc_jtprod.argtypes = [vp, pd, pd, pd]
        L.orc_hprod.argtypes = [vp, pd, pd, pd, dbl, pd]
        _LIB = L
    return _LIB


class OracleModel:
    """NLPModels-style evaluator over a blob, on the host CPU."""

    def __init__(self, blob: bytes):
        self._L = lib()
        self._h = self._L.orc_create(blob, len(blob))
        if not self._h:
            raise ValueError("oracle rejected the blob")
        meta = np.zeros(8, dtype=np.int64)
        self._L.orc_meta(self._h, meta)
        self.nvar, self.ncon, self.npar, self.nnzj, self.nnzh, mini, self.n_templates = (int(v) for v in meta[:7])
        self.minimize = bool(mini)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_destroy(self._h)
            self._h = None

    def set_threads(self, n: int):
        self._L.orc_set_threads(int(n))

    def max_threads(self) -> int:
        return int(self._L.orc_max_threads())

    def template_info(self, i: int) -> dict:
        out = np.zeros(8, dtype=np.int64)
        self._L.orc_template_info(self._h, i, out)
        keys = ("kind", "n_items", "o0", "o1", "o2", "o1step", "o2step")
        return dict(zip(keys, (int(v) for v in out[:7])))

    def _arr(self, which: int, n: int) -> np.ndarray:
        out = np.zeros(n)
        self._L.orc_get_array(self._h, which, out)
        return out

    x0 = property(lambda s: s._arr(0, s.nvar))
    lvar = property(lambda s: s._arr(1, s.nvar))
    uvar = property(lambda s: s._arr(2, s.nvar))
    lcon = property(lambda s: s._arr(3, s.ncon))
    ucon = property(lambda s: s._arr(4, s.ncon))
    y0 = property(lambda s: s._arr(5, s.ncon))
    theta = property(lambda s: s._arr(6, s.npar))

    def set_parameter(self, off: int, vals):
        v = np.ascontiguousarray(vals, dtype=np.float64)
        self._L.orc_set_parameter(self._h, off, v.shape[0], v)

    @staticmethod
    def _x(x):
        return np.ascontiguousarray(x, dtype=np.float64)

    def obj(self, x) -> float:
        return float(self._L.orc_obj(self._h, self._x(x)))

    def cons(self, x) -> np.ndarray:
        c = np.zeros(self.ncon)
        self._L.orc_cons(self._h, self._x(x), c)
        return c

    def grad(self, x) -> np.ndarray:
        g = np.zeros(self.nvar)
        self._L.orc_grad(self._h, self._x(x), g)
        return g

    def jac_structure(self, base: int = 0):
        r = np.zeros(self.nnzj, dtype=np.int64)
        c = np.zeros(self.nnzj, dtype=np.int64)
        self._L.orc_jac_structure(self._h, r, c, base)
        return r, c

    def jac_coord(self, x) -> np.ndarray:
        v = np.zeros(self.nnzj)
        self._L.orc_jac_coord(self._h, self._x(x), v)
        return v

    def jprod(self, x, v) -> np.ndarray:
        out = np.zeros(self.ncon)
        self._L.orc_jprod(self._h, self._x(x), self._x(v), out)
        return out

    def jtprod(self, x, v) -> np.ndarray:
        out = np.zeros(self.nvar)
        self._L.orc_jtprod(self._h, self._x(x), self._x(v), out)
        return out

    def hprod(self, x, y, v, obj_weight: float = 1.0) -> np.ndarray:
        out = np.zeros(self.nvar)
        self._L.orc_hprod(self._h, self._x(x), self._x(y), self._x(v), float(obj_weight), out)
        return out

    def hess_structure(self, base: int = 0):
        r = np.zeros(self.nnzh, dtype=np.int64)
        c = np.zeros(self.nnzh, dtype=np.int64)
        self._L.orc_hess_structure(self._h, r, c, base)
        return r, c

    def hess_coord(self, x, y, obj_weight: float = 1.0) -> np.ndarray:
        v = np.zeros(self.nnzh)
        self._L.orc_hess_coord(self._h, self._x(x), self._x(y), float(obj_weight), v)
        return v

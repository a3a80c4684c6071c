#!/usr/bin/env python3
"""FP64 library GEMM shapes of the hub solver's Schur products: C[w x w] -= A' Z with A, Z [K x w] (K = lanes x nQ = 1000) —
as written (transposed view of A), with A' made contiguous first, and with the product cut into lower-triangle chunks."""
import torch, time
torch.manual_seed(0)
def timed(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for K, w in ((1000, 4999), (1000, 4095), (1000, 2047), (1000, 1023), (1000, 511)):
    A = torch.randn(K, w, dtype=torch.float64, device="cuda"); Z = torch.randn(K, w, dtype=torch.float64, device="cuda")
    C = torch.zeros(w, w, dtype=torch.float64, device="cuda")
    fl = 2.0 * K * w * w
    t_tn = timed(lambda: C.sub_(A.t() @ Z))
    def nn():
        At = A.t().contiguous(); C.sub_(At @ Z)
    t_nn = timed(nn)
    t_addmm = timed(lambda: C.addmm_(A.t(), Z, alpha=-1.0))
    print(f"K {K} w {w}: A'Z as written {t_tn:.3f} ms ({fl / t_tn / 1e9:.1f} TF)   contiguous A' {t_nn:.3f} ms   addmm_ in place {t_addmm:.3f} ms ({fl / t_addmm / 1e9:.1f} TF)")

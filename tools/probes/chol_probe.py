"""Dense SPD factorisation of the hubs' Schur complement (5 000 x 5 000, FP64): what the libraries reach on this GPU."""
import torch, time
n=5000
A=torch.randn(n,n,dtype=torch.float64,device="cuda"); S=A@A.T+n*torch.eye(n,dtype=torch.float64,device="cuda")
def timed(fn,it=3):
    fn(); torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/it*1e3
print("default linalg backend:", torch.backends.cuda.preferred_linalg_library())
print("torch.linalg.cholesky_ex %.2f ms" % timed(lambda: torch.linalg.cholesky_ex(S)))
print("torch.linalg.lu_factor %.2f ms" % timed(lambda: torch.linalg.lu_factor(S)))
try:
    print("torch.linalg.ldl_factor %.2f ms" % timed(lambda: torch.linalg.ldl_factor(S)))
except Exception as e: print("ldl_factor:", str(e)[:100])
for lib in ("magma", "cusolver"):
    try:
        torch.backends.cuda.preferred_linalg_library(lib)
        print(lib, "cholesky_ex %.2f ms" % timed(lambda: torch.linalg.cholesky_ex(S)), " lu_factor %.2f ms" % timed(lambda: torch.linalg.lu_factor(S)))
    except Exception as e: print(lib, "->", str(e)[:120])
torch.backends.cuda.preferred_linalg_library("default")
# recursive Cholesky: everything below a leaf size is a library call, everything above GEMMs with explicitly inverted triangles
def tri_inv(Lkk):
    return torch.linalg.solve_triangular(Lkk, torch.eye(Lkk.shape[0], dtype=Lkk.dtype, device=Lkk.device), upper=False)
def blocked(S, NB):
    L=S.clone(); n=S.shape[0]
    for k in range(0,n,NB):
        e=min(k+NB,n)
        Lkk=torch.linalg.cholesky(L[k:e,k:e]); L[k:e,k:e]=Lkk
        if e<n:
            P=L[e:,k:e]@tri_inv(Lkk).T
            L[e:,k:e]=P
            L[e:,e:]-=P@P.T
    return L
for NB in (256,512,1024):
    print("blocked (GEMM trsm) NB", NB, "%.2f ms" % timed(lambda: blocked(S,NB)))
Lr=torch.linalg.cholesky(S); Lb=blocked(S,512)
print("err", (torch.tril(Lb)-Lr).abs().max().item())
for m in (128,256,512,1024):
    B=S[:m,:m].contiguous()
    print("size", m, "cholesky %.3f ms" % timed(lambda: torch.linalg.cholesky(B),10), " tri_inv %.3f ms" % timed(lambda: tri_inv(torch.linalg.cholesky(B)),10))
b=torch.randn(n,1,dtype=torch.float64,device="cuda")
print("cholesky_solve %.2f ms" % timed(lambda: torch.cholesky_solve(b, Lr)))
print("solve_triangular x2 %.2f ms" % timed(lambda: torch.linalg.solve_triangular(Lr.T, torch.linalg.solve_triangular(Lr, b, upper=False), upper=True)))

import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
from spadot_amd import ops
dev='cuda:0'
for m in (236, 310, 330, 600, 621, 870, 1200, 1300):
    rng=np.random.default_rng(m)
    # matrices shaped like the SVGP's Sigma_l: K_mm + c K_mn diag(w) K_nm + 1e-2 I, Gaussian kernel scale 0.1
    z=torch.as_tensor(rng.normal(size=(m,2))).to(dev); x=torch.as_tensor(rng.normal(size=(512,2))).to(dev)
    Kmm=ops.kernel_matrix(z,z); Knm=ops.kernel_matrix(x,z)
    w=torch.as_tensor(rng.uniform(0.3,3.0,size=(10,512))).to(dev)
    A=(Kmm[None]+20.0*torch.einsum("bm,lb,bn->lmn",Knm,w,Knm)+1e-2*torch.eye(m,dtype=torch.float64,device=dev)).contiguous()
    X,ld=ops.spd_inverse_logdet(A)
    err=float((A@X-torch.eye(m,dtype=torch.float64,device=dev)).abs().max())
    ldr=torch.linalg.slogdet(A)[1]; Xr=torch.linalg.inv(A)
    err=max(err, float((ld-ldr).abs().max())); print("   rel X err vs torch", float(((X-Xr).abs().max()/Xr.abs().max())), "cond", float(torch.linalg.cond(A[0])))
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): ops.spd_inverse_logdet(A)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
    t1=time.perf_counter()
    for _ in range(20): torch.linalg.inv_ex(A, check_errors=False)
    torch.cuda.synchronize(); dt2=(time.perf_counter()-t1)/20
    print(f"m={m:4d} sweep {dt*1e3:7.3f} ms  torch.inv {dt2*1e3:7.3f} ms  resid {err:.1e}")

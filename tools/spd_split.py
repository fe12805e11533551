import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
from spadot_amd import ops
dev='cuda:0'
for m in (200, 236, 260, 279, 300, 310, 400, 480, 600):
    rng=np.random.default_rng(m)
    z=torch.as_tensor(rng.normal(size=(m,2))).to(dev); x=torch.as_tensor(rng.normal(size=(512,2))).to(dev)
    Kmm=ops.kernel_matrix(z,z); Knm=ops.kernel_matrix(x,z)
    w=torch.as_tensor(rng.uniform(0.3,3.0,size=(10,512))).to(dev)
    A=(Kmm[None]+20.0*torch.einsum("bm,lb,bn->lmn",Knm,w,Knm)+1e-2*torch.eye(m,dtype=torch.float64,device=dev)).contiguous()
    row=[]
    for split in (128, 160, 200, 256, 279, 310):
        ops._SPLIT[0]=split
        g=torch.cuda.CUDAGraph()
        ops.spd_inverse_logdet(A); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            X,ld=ops.spd_inverse_logdet(A)
        g.replay(); torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize(); row.append((time.perf_counter()-t0)/20*1e3)
    print(f"m={m:4d} " + "  ".join(f"{t:6.3f}" for t in row), flush=True)

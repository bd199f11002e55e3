import os.path as osp, sys
sys.path.insert(0, osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__)))))
import torch, time
from spark_sched_sim_amd.decima import make_mlp
from spark_sched_sim_amd.train_kernels import KernelMLP
import os
if os.environ.get("FUSE_WIDE"):
    KernelMLP.FUSE_WIDE = True
dev=torch.device("cuda:0")
for dims, act_cls, kw in (((5,32,16,16),"LeakyReLU",dict(negative_slope=0.2)), ((16,32,16,16),"LeakyReLU",dict(negative_slope=0.2)), ((21,32,16,16),"LeakyReLU",dict(negative_slope=0.2)), ((53,64,64,1),"Tanh",{}), ((36,64,64,1),"Tanh",{})):
    mlp = make_mlp(dims[0],[dims[1],dims[2]],dims[3],act_cls,kw).to(dev)
    rows = 2_500_000 if dims[3]==16 else 600_000
    x = torch.randn((rows,dims[0]),device=dev,requires_grad=True)
    w = torch.randn((rows,dims[3]),device=dev)
    for name, fwd in (("fused", lambda t: mlp(t)), ("layers", lambda t: torch.nn.Sequential.forward(mlp, t))):
        for _ in range(3):
            (fwd(x)*w).sum().backward()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(10):
            y = fwd(x)
        torch.cuda.synchronize(); t1=time.perf_counter()
        for _ in range(10):
            y = fwd(x); y.backward(w)
        torch.cuda.synchronize(); t2=time.perf_counter()
        print(dims, rows, name, "fwd %.3f ms, fwd+bwd %.3f ms" % ((t1-t0)*100, (t2-t1)*100))

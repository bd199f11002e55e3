import os.path as osp, sys
sys.path.insert(0, osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__)))))
import torch, time
from spark_sched_sim_amd.decima import make_mlp
from spark_sched_sim_amd.train_kernels import mlp_forward, mlp_backward, pack_mlp, linear_wgrad
dev=torch.device("cuda:0")
def T(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)*1e3/n
for dims in ((5,32,16,16),(16,32,16,16),(21,32,16,16)):
    mlp = make_mlp(dims[0],[dims[1],dims[2]],dims[3],"LeakyReLU",dict(negative_slope=0.2)).to(dev)
    rows=2_500_000
    x=torch.randn((rows,dims[0]),device=dev); dy=torch.randn((rows,16),device=dev)
    pk=pack_mlp(mlp[0],mlp[2],mlp[4])
    a1,a2,y=mlp_forward(x,pk,dims,0,0.2)
    g1,g2,dx=mlp_backward(dy,a1,a2,pk,dims,0,0.2)
    print(dims, "fwd %.3f bwd(dx) %.3f bwd(no dx) %.3f wgrad3 %.3f wgrad2 %.3f wgrad1 %.3f" % (
        T(lambda: mlp_forward(x,pk,dims,0,0.2)), T(lambda: mlp_backward(dy,a1,a2,pk,dims,0,0.2)), T(lambda: mlp_backward(dy,a1,a2,pk,dims,0,0.2,want_dx=False)),
        T(lambda: linear_wgrad(a2,dy)), T(lambda: linear_wgrad(a1,g2)), T(lambda: linear_wgrad(x,g1))))

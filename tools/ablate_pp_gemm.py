import os, sys, statistics, torch
import os as _os; _os.environ.setdefault('VQCPC_LAB', '1')   # measurement switches live in the lab build (vqcpc_bach_amd/build.py)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vqcpc_bach_amd import hip, ops
hip.load(); hip.set_gemm_mode(1)
M=557056
for N,K in [(768,256),(256,256),(256,1024),(1024,256)]:
    a=torch.randn(M,K,device='cuda'); b=torch.randn(N,K,device='cuda'); bias=torch.randn(N,device='cuda'); out=torch.empty(M,N,device='cuda')
    ts=[]
    for r in range(6):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gemm_nt(a,b,bias=bias,out=out)
        e1.record(); torch.cuda.synchronize()
        if r: ts.append(2.0*M*N*K/(e0.elapsed_time(e1)/10*1e-3)/1e12)
    print(os.environ.get('VQCPC_PP_ABL','0'),(M,N,K),round(statistics.median(ts),1),flush=True)

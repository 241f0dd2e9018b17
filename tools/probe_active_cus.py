"""Is the bf16x6 NT GEMM bound by its schedule or by the chip's power budget?  The persistent 256-tile kernel is launched on
fewer workgroups than CUs (VQCPC_PP_GRID, measurement only) with and without its output stores (VQCPC_PP_ABL=32):
    for g in 256 128 64 32; do for a in 0 32; do VQCPC_PP_GRID=$g VQCPC_PP_ABL=$a python tools/probe_active_cus.py; done; done
Per-CU throughput rises 1.5-1.7x when a quarter of the CUs or fewer are active (the clock is no longer held at ~1.4 GHz by the
1.24 kW the full chip draws), see profiles/r03_gemm_power_limit.txt."""
import os, sys, statistics, torch
import os as _os; _os.environ.setdefault('VQCPC_LAB', '1')   # measurement switches live in the lab build (vqcpc_bach_amd/build.py)
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from vqcpc_bach_amd import hip, ops
hip.load(); hip.set_gemm_mode(1)
M=557056
for N,K in [(1024,256),(256,1024)]:
    a=torch.randn(M,K,device='cuda'); b=torch.randn(N,K,device='cuda'); bias=torch.randn(N,device='cuda'); out=torch.empty(M,N,device='cuda')
    ts=[]
    for r in range(4):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): ops.gemm_nt(a,b,bias=bias,out=out)
        e1.record(); torch.cuda.synchronize()
        if r: ts.append(2.0*M*N*K/(e0.elapsed_time(e1)/4*1e-3)/1e12)
    g=int(os.environ.get('VQCPC_PP_GRID','256'))
    print('grid',g,'abl',os.environ.get('VQCPC_PP_ABL','0'),(N,K),'TF',round(statistics.median(ts),1),'TF per CU', round(statistics.median(ts)/g,3),flush=True)

import sys, time, threading
sys.path.insert(0, '.')
import torch
from sfft_amd.plan import get_plan
from sfft_amd.utils.synthetic import make_pair
N=4096; dev=torch.device('cuda',0)
pair=make_pair(N,N,seed=1234,mask=True,sky=0.0,bkg_scale=0.05)
g={k:torch.from_numpy(v).to(dev) for k,v in pair.items()}
for nthr in (1,2,3):
    plans=[get_plan(N,N,8,2,2,True,0,slot=i) for i in range(nthr)]
    streams=[torch.cuda.Stream(dev) for _ in range(nthr)]
    outs=[(torch.empty(plans[0].NEQ,dtype=torch.float64,device=dev), torch.empty((N,N),dtype=torch.float64,device=dev)) for _ in range(nthr)]
    K=12
    def work(i, n):
        with torch.cuda.stream(streams[i]):
            for _ in range(n):
                plans[i].subtract(g['REF'],g['SCI'],g['mREF'],g['mSCI'],out_solution=outs[i][0],out_diff=outs[i][1])
    for i in range(nthr): work(i,2)
    torch.cuda.synchronize()
    t0=time.perf_counter()
    th=[threading.Thread(target=work,args=(i,K)) for i in range(nthr)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    dt=time.perf_counter()-t0
    print("threads",nthr,"pairs/s",nthr*K/dt, "same result:", bool(torch.equal(outs[0][1],outs[-1][1])))

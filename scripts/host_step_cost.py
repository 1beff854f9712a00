import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
import numpy as np, torch, torch.distributed as dist
import rome_jl_amd as R
from rome_jl_amd.distributed import PipelinedSegmentSweep
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
N=100
fg=R.loadG2o("/root/repo/tests/golden/manhattan.g2o", N=N)
cov=np.diag([1/44.6,1/399.0,1/9591.0])
fg.addVariable("ghost_prev",R.Pose2); fg.addVariable("ghost_next",R.Pose2)
fg.addFactor(["ghost_prev","x0"],R.Pose2Pose2(R.MvNormal([1.0,0,0],cov)))
fg.addFactor(["x3499","ghost_next"],R.Pose2Pose2(R.MvNormal([1.0,0,0],cov)))
R.dead_reckon_init(fg,seed=11)
dg=R.DeviceGraph(fg); dg.upload_beliefs(fg); pk=dg.packed
opts=R.make_opts(N=N,solver=1,seed=1)
vf,vt=pk.p2p2["var_from"],pk.p2p2["var_to"]
f_first=int(np.nonzero((vf==pk.index["x0"])&(vt==pk.index["x1"]))[0][0]); f_last=int(np.nonzero((vf==pk.index["x3498"])&(vt==pk.index["x3499"]))[0][0])
pipe=PipelinedSegmentSweep(dg,opts,dist,1,0,[2*f_first+1,2*f_last],pk.index["ghost_prev"],pk.index["ghost_next"],always_collective=True,depth=int(os.environ.get("ROME_PIPE_DEPTH","2")))
for _ in range(2000): pipe.step()
pipe.drain(); torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(2000): pipe.step()
th=time.perf_counter()-t
pipe.drain(); torch.cuda.synchronize()
tt=time.perf_counter()-t
print("host enqueue per step %.1f us, total per step %.1f us"%(th/2000*1e6, tt/2000*1e6))
# components
st=pipe.streams[0]
t=time.perf_counter()
for _ in range(2000):
    with torch.cuda.stream(st): pass
print("stream ctx %.1f us"%((time.perf_counter()-t)/2000*1e6))
torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(500): pipe.plans[0]()
th=(time.perf_counter()-t)/500*1e6; torch.cuda.synchronize()
print("plan launch host %.1f us"%th)
t=time.perf_counter()
ws=[]
for _ in range(500): ws.append(dist.all_gather_into_tensor(pipe.recv[0].view(-1), pipe.send[0].view(-1), async_op=True))
th=(time.perf_counter()-t)/500*1e6; torch.cuda.synchronize()
print("all_gather host %.1f us"%th)
t=time.perf_counter()
for w in ws: w.wait()
print("wait host %.1f us"%((time.perf_counter()-t)/500*1e6))
pg=dist.distributed_c10d._get_default_group()
o=pipe.recv[0].view(-1); i=pipe.send[0].view(-1)
t=time.perf_counter()
for _ in range(500): w=pg._allgather_base(o,i)
th=(time.perf_counter()-t)/500*1e6; torch.cuda.synchronize()
print("pg._allgather_base host %.1f us"%th)
dist.destroy_process_group()

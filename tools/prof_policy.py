import sys, time, torch
sys.path.insert(0,'/root/repo')
from hope_amd import policy as P, agent_glue as G
dev='cuda'
N=65536
net=P.HopeNet(P.actor_configs(use_img=False)).to(dev)
obs={'lidar':torch.randn(N,120,device=dev),'target':torch.randn(N,5,device=dev),'action_mask':torch.rand(N,42,device=dev)}
sn=G.BatchedStateNorm(device=dev); sn.update(obs)
def step():
    with torch.no_grad():
        nz=sn.normalize({'lidar':obs['lidar'],'target':obs['target']})
        o={'lidar':nz['lidar'].float(),'target':nz['target'].float(),'action_mask':obs['action_mask']}
        mean=torch.clamp(net(o),-1,1)
        a,_=G.choose_action(mean, torch.ones_like(mean), obs['action_mask'])
        sn.update(obs)
    return a
for _ in range(5): step()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); print('ms/step', (time.perf_counter()-t)/20*1e3)
def fwd_only():
    with torch.no_grad(): return net(obs)
for _ in range(5): fwd_only()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(20): fwd_only()
torch.cuda.synchronize(); print('net fwd ms', (time.perf_counter()-t)/20*1e3)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(5): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=25, max_name_column_width=60))

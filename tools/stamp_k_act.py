import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
PKG="master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd"
S=importlib.import_module(PKG); D=importlib.import_module(PKG+".ddpg")
n=65536
tab=S.tables.synthetic_table("train",98)
env=S.ShemsBatch(n,72,[tab],[S.make_config(98,0,tab.shape[0])]).use_torch_stream()
ag=D.Agent(seed=1)
env.reset_(1,episode=0)
st=env.state; ag.set_norm(st.min(0), st.max(0))
blk=torch.zeros(1024,dtype=torch.float64,device='cuda')
for t in range(5):
    ag.act_step(env, train=True, tick=t, block_reward=blk)
torch.cuda.synchronize()
v=blk.cpu().numpy().view(np.uint64)[:26].reshape(13,2).astype(np.int64)
t=v[:,0]-v[0,0]; rt=v[:,1]-v[0,1]
names=['start','stage0','L1g0+sync','chunks 0-13 end','chunk 14 end','-','-','-','-','-','loop end','epilogue red end','env step end']
for nme,a,b in zip(names,t,rt): print(f"{nme:18s} cyc {a:8d}  t_us {b/100:8.2f}")
print('clock GHz', t[12]/max(1,rt[12])*0.1)

import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
PKG="master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd"
S=importlib.import_module(PKG); D=importlib.import_module(PKG+".ddpg")
wl=D.TrainWorkload(S, torch, 4096, seed=7, updates=1)
for _ in range(30): wl.agent.replay(wl.ring)
torch.cuda.synchronize()
ws=wl.agent.ws.cpu().numpy()
BP=128; WS_SLOT0 = (9+9+2+1+1+1+1+2+1+1+2+1)*BP + 8*8*2*BP + 4*12*256
SL_P3 = 500*BP; SL_D1P = SL_P3 + 16*2*BP; SL_SIZE = SL_D1P + 8*250*BP
for slot,name in ((2,'critic bwd'),(4,'critic2 bwd (I only)'),(3,'actor bwd')):
    base = WS_SLOT0 + slot*SL_SIZE + SL_D1P + 8*250*128 - 4096
    for off,kind in ((0,'W wg0'),(64,'I wg0')):
        st = ws[base+off: base+off+20].copy().view(np.uint64).reshape(5,2).astype(np.int64)
        t=st[:,0]-st[0,0]
        print(name, kind, t.tolist())

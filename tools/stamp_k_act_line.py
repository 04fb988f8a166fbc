import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
PKG="master-thesis-deep-reinforcement-learning-ddpg-in-home-energy-management_amd"
S=importlib.import_module(PKG); D=importlib.import_module(PKG+".ddpg")
n=int(sys.argv[1]) if len(sys.argv) > 1 else 65536
tab=S.tables.synthetic_table("train",98)
env=S.ShemsBatch(n,72,[tab],[S.make_config(98,0,tab.shape[0])]).use_torch_stream()
ag=D.Agent(seed=1)
env.reset_(1,episode=0)
st=env.state; ag.set_norm(st.min(0), st.max(0))
blk=torch.zeros(1024,dtype=torch.float64,device='cuda')
res=[]
for t in range(8):
    ag.act_step(env, train=True, tick=t, block_reward=blk)
    torch.cuda.synchronize()
    v=blk.cpu().numpy().view(np.uint64)[:26].reshape(13,2).astype(np.int64)
    res.append(v.copy())
v=res[-1]
t=v[:,0]-v[0,0]; rt=v[:,1]-v[0,1]
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for k in range(20): ag.act_step(env, train=True, tick=100+k, block_reward=blk)
e1.record(); torch.cuda.synchronize()
if n < 16384 and os.environ.get('SHEMS_ACT_FORM', '-1') in ('-1', '8', '9'):      # column-group forms (k_actg): + 8 (layer 3 done) 9 (ticket taken; split form)
    print(f"n={n} k_actg form={os.environ.get('SHEMS_ACT_FORM','-1')}: stage0 {t[1]:6d}  layer1 {t[2]-t[1]:6d}  loop {t[10]-t[2]:6d} (chunk 0: {t[3]-t[2]}, chunks 1-14: {(t[4]-t[3])/14:.0f} each, chunk 15: {t[10]-t[4]})  layer3 {t[8]-t[10]:6d}  handoff {t[9]-t[8]:6d}  | tile 0's finisher: sums at {t[11]:6d}  env {t[12]-t[11]:6d} [noise {t[5]-t[11]} loads {t[6]-t[5]} step {t[7]-t[6]} stores {t[12]-t[7]}]  total {t[12]:6d} cyc = {rt[12]/100:6.2f} us  clk {t[12]/max(1,rt[12])*0.1:.3f} GHz  kernel_us {e0.elapsed_time(e1)*50:.1f}")
    raw = blk.cpu().numpy().view(np.uint64).astype(np.int64)
    print("   per-wave loop start:", [int(x - res[-1][0, 0]) for x in raw[48:56]], " loop end:", [int(x - res[-1][0, 0]) for x in raw[32:40]])
    sys.exit(0)
if n < 32768 or os.environ.get('SHEMS_ACT_FORM4', '1') != '0':      # free-running forms: the free-running form stamps 0 (start) 1 (stage 0 done) 2 (layer 1 done) 10 (loop done) 11 (layer 3) 12 (env tail)
    print(f"n={n} form={os.environ.get('SHEMS_ACT_FORM','3')}/{os.environ.get('SHEMS_ACT_FORM4','1')}: stage0 {t[1]:6d}  layer1 {t[2]-t[1]:6d}  loop {t[10]-t[2]:6d} (chunk 0: {t[3]-t[2]}, chunks 1-14: {(t[4]-t[3])/14:.0f} each, chunk 15: {t[10]-t[4]})  epi {t[11]-t[10]:6d}  env {t[12]-t[11]:6d} [noise {t[5]-t[11]} loads {t[6]-t[5]} step {t[7]-t[6]} stores {t[12]-t[7]}]  total {t[12]:6d} cyc = {rt[12]/100:6.2f} us  clk {t[12]/max(1,rt[12])*0.1:.3f} GHz  kernel_us {e0.elapsed_time(e1)*50:.1f}")
    sys.exit(0)
print(f"{os.path.basename(os.environ.get('SHEMS_HIP_LIB','default')):22s} pro {t[2]:6d}  cyc/chunk(0-13) {(t[3]-t[2])/14:8.1f}  c14 {t[4]-t[3]:6d} c15 {t[10]-t[4]:6d} epi {t[11]-t[10]:6d} env {t[12]-t[11]:6d} total_us {rt[12]/100:6.2f} clk {t[12]/max(1,rt[12])*0.1:.3f}  kernel_us {e0.elapsed_time(e1)*50:.1f}")

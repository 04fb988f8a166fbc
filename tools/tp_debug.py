import sys, os, importlib
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import util as U, ddpg_oracle as DO
import test_group_gpu as TG
torch, S, D, G, env, grp = TG._setup(L=3, E=128, cap=2400, form="throughput")
grp.store_grad = True
batch = 128
rng = np.random.default_rng(5)
host = []
for l, ag in enumerate(grp.learners):
    ag.batch = batch
    pa, pc = ag.actor.cpu().numpy().copy(), ag.critic.cpu().numpy().copy()
    pa[128000:129000] *= 30.0; pc[128250:128750] *= 30.0
    pa[2250:2500] = rng.normal(0, 0.05, 250); pc[2750:3000] = rng.normal(0, 0.05, 250)
    ag.set_params(actor=pa, critic=pc)
    ring = grp.rings[l]
    ring.done.copy_(torch.from_numpy((rng.random(ring.capacity) < 0.05).astype(np.uint8)))
    host.append(dict(s=ring.s.cpu().numpy(), s_min=ag.s_min.cpu().numpy(), s_max=ag.s_max.cpu().numpy()))
for tick in (3, 4):
    pre = [(ag.actor.cpu().numpy().copy(), ag.critic.cpu().numpy().copy()) for ag in grp.learners]
    grp.replay(tick=tick); torch.cuda.synchronize()
    for l, (ag, h) in enumerate(zip(grp.learners, host)):
        idx = DO.sample_indices(grp.rng_seed + l, tick, batch, len(grp.rings[l]))
        Lr = DO.Learner(pre[l][0], ag.critic.cpu().numpy(), h["s_min"], h["s_max"])
        s = h["s"][idx]
        ga64, _ = Lr.actor_grad(s, dtype=np.float64)
        ga = ag.grad_actor.cpu().numpy()
        for n, lo, hi in DO.blocks(9, 2):
            e = np.abs(ga[lo:hi] - ga64[lo:hi]) / np.abs(ga64[lo:hi]).max()
            bad = np.nonzero(e > 2e-6)[0]
            print(tick, l, n, "max err", e.max(), "n bad", len(bad), bad[:10])
        # pre-activations of layer 2 of the actor in float64
        sn = DO.normalize(s, h["s_min"], h["s_max"]).astype(np.float64)
        W1, b1, W2, b2, W3, b3 = DO.split(pre[l][0].astype(np.float64), 9, 2)
        print("shapes", W1.shape, W2.shape)
        h1 = np.maximum(sn @ W1.reshape(9, 250) + b1, 0) if W1.shape != (250, 9) else None
        try:
            z2 = h1 @ W2.reshape(250, 500) + b2
            k = np.argsort(np.abs(z2).ravel())[:5]
            print("smallest |pre2|:", np.abs(z2).ravel()[k], np.unravel_index(k, z2.shape))
        except Exception as ex:
            print("z2 failed", ex)

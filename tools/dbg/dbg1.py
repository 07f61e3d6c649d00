import numpy as np, torch, sys
sys.path.insert(0,'.')
from psnerf_amd import hip
g=np.load('tests/golden/stage1_composite.npz')
for S in (64,96):
    a=torch.from_numpy(g['alpha%d'%S]).cuda(); c=torch.from_numpy(g['rgb%d'%S]).cuda()
    da,dc=hip.composite_bwd(a,c,torch.from_numpy(g['c1_%d'%S]).cuda(),torch.from_numpy(g['c2_%d'%S]).cuda(),True)
    ref=g['dalpha%d'%S]
    err=np.abs(da.cpu().numpy()-ref)
    print(S,'row err',np.round(err.max(1)[:10],5), 'argmax cols', err.argmax(1)[:10])
    r=int(err.max(1).argmax()); print('worst row',r, da.cpu().numpy()[r,:6], ref[r,:6])

"""Control experiment for tests/test_convergence_gpu.py: the CPU oracle against ITSELF, initial weights perturbed by 1e-7 relative, 300 stage-2 steps across the train_fix switch -> spread of the final PSNR / loss curves of the reference arithmetic."""
import math, sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.helpers import stage2_state_dict
from psnerf_amd.synthetic import stage2_inputs
from oracle import stage2 as o2
torch.set_num_threads(8)
def psnr(a,b,m):
    a,b=a.double().reshape(-1,3),b.double().reshape(-1,3); m=m.reshape(-1).bool()
    return -10*math.log10(float(((a[m]-b[m])**2).mean()))
def run(perturb, n_steps=300, lr_scale=1.0):
    N,L,V,n_views=640,8,4,3; NL=L*n_views
    conf=o2.bear_conf()
    teacher=o2.PSNetwork(conf); teacher.load_state_dict(stage2_state_dict(conf, seed=77))
    with torch.no_grad(): teacher.visibility_net.linears[-1].bias += 0.75
    views=[]; g=torch.Generator().manual_seed(5)
    for v in range(n_views):
        inp,gt=stage2_inputs(N,L,V,seed=300+v)
        with torch.no_grad(): t_out=teacher(inp, noise={'xyz': torch.zeros(int(inp['surface_mask'].sum()),3)})
        gt={'rgb': t_out['sg_rgb_values'].detach().clone()}
        sm=inp['surface_mask'][0]; thr=t_out['vis_train'][:,sm,0].median()
        inp['vis_train_gt']=(t_out['vis_train'][...,0]>thr).float(); inp['visibility']=(t_out['visibility'][...,0]>thr).float()
        td=inp.pop('light_direction'); inp.pop('light_intensity'); views.append((inp,gt,td))
    light_init=torch.nn.functional.normalize(torch.cat([t for _,_,t in views])+0.05*torch.randn(NL,3,generator=g),dim=-1)
    sd=stage2_state_dict(conf, seed=9)
    if perturb:
        gp=torch.Generator().manual_seed(SEED)
        sd={k:(v*(1+perturb*torch.randn(v.shape,generator=gp)) if v.dtype.is_floating_point else v) for k,v in sd.items()}
    onet=o2.PSNetwork(conf); onet.load_state_dict(sd)
    ostep=o2.TrainStep(onet, conf, NL, light_init)
    start=5000-n_steps//2
    tr=ostep; tr.cur_iter=start; tr._ori=(1.0,0.05,0.01,1)
    tr.loss.sg_rgb_weight,tr.loss.albedo_smooth_weight,tr.loss.rough_smooth_weight,tr.loss.vis_weight=0,0,0,10
    tr.model.albedo_net.eval().requires_grad_(False); tr.model.rough_net.eval().requires_grad_(False)
    tr.light_para.requires_grad_(False); tr.light_inten_para.requires_grad_(False)
    def render():
        vals=[]
        with torch.no_grad():
            for v,(inp,gt,_) in enumerate(views):
                mi=dict(inp); l_slt=torch.arange(L)+L*v
                mi['light_direction']=torch.nn.functional.normalize(tr.light_para.weight.detach()[l_slt],dim=-1)
                mi['light_intensity']=tr.light_inten_para.weight.detach()[l_slt]
                out=onet(mi, noise={'xyz': torch.zeros(int(inp['surface_mask'].sum()),3)})
                vals.append(psnr(out['sg_rgb_values'],gt['rgb'],(inp['surface_mask']&inp['object_mask']).expand(L,-1)))
        return float(np.mean(vals))
    p0=render(); lo=[]
    for it in range(n_steps):
        v=it%n_views; inp,gt,_=views[v]; l_slt=torch.arange(L)+L*v
        nz=torch.randn(int(inp['surface_mask'].sum()),3,generator=g)*0.01
        ot,_=ostep.step(inp,gt,l_slt,noise={'xyz':nz})
        lo.append(float(ot['total'].detach()))
    return np.array(lo), p0, render()

res=[]
for SEED in (1,2,3,4,5):
    lo,p0,p1=run(1e-7)
    res.append((lo,p1)); print(SEED, p0, p1, lo[-10:].mean(), flush=True)
ps=np.array([r[1] for r in res]); print('psnr mean %.4f std %.4f range %.4f'%(ps.mean(), ps.std(), ps.max()-ps.min()))

import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from conftest import fold_cloth1_mask, make_cloth_case, cloth_reset_x
from test_oracle_cloth import pnp_actions
from test_cloth_gpu import Conf, _run_hip, _rel
from unidom_amd.engine.cloth_simulator import ClothSimulator
from oracle.pyoracle import ClothOracle
mask=fold_cloth1_mask()
fs=ClothSimulator(Conf(),4,lambda x,v,i,j:v,mask)
es=ClothSimulator(Conf(),4,lambda x,v,i,j:v,mask,exact=True)
def realistic(B,seed):
    rng=np.random.default_rng(seed)
    x=np.repeat(cloth_reset_x()[None],B,0).astype(np.float32)
    x[:,:,[0,2]]+=(rng.normal(size=(B,1,2))*0.05).astype(np.float32)
    v=np.zeros_like(x)
    prim=np.zeros((B,2,4),np.float32); prim[:,0]=[0.5,0.5,0.5,0.01]; prim[:,1]=[1,1,1,0.01]
    acts=[]
    for b in range(B):
        p=rng.integers(0,512); q=rng.integers(0,512)
        a6=np.concatenate([x[b,p],x[b,q]]).astype(np.float32)
        acts.append(pnp_actions(a6,prim[b,0]))
    actions=np.concatenate(acts,1).astype(np.float32)
    k=np.full(B,900,np.float32); mu=np.full(B,0.5,np.float32)
    return x,v,prim,k,mu,actions
for S in (1,5,50):
    orc=ClothOracle(mask,substeps=S)
    fs.substeps=S
    import ctypes
    # separate sims per substep count
    class C2(Conf): pass
    rng=np.random.default_rng(0)
    x,v,prim,k,mu,actions=make_cloth_case(rng,2,1)
    o=orc.rollout_fwd(x,v,prim,k,mu,actions)
    print('violent S',S,'(oracle only)')
# realistic full
orc=ClothOracle(mask)
x,v,prim,k,mu,actions=realistic(4,0)
o=orc.rollout_fwd(x,v,prim,k,mu,actions,want_lists=True,want_grasp=True,nthreads=4)
h=_run_hip(fs,x,v,prim,k,mu,actions)
e=_run_hip(es,x,v,prim,k,mu,actions)
print('realistic: exact==oracle', np.array_equal(e['x'],o['x']), 'grasp total',o['grasp'].sum(),'fast grasp equal',np.array_equal(h['grasp'],o['grasp']), 'mismatch', (h['grasp']!=o['grasp']).sum())
for t in (0,2,12,32,39):
    print(' t',t,'x rel',_rel(h['x_list'][t],o['x_list'][t]),'abs',np.abs(h['x_list'][t]-o['x_list'][t]).max(),'v rel',_rel(h['v_list'][t],o['v_list'][t]),'vmax',np.abs(o['v_list'][t]).max())
# bwd realistic
rng=np.random.default_rng(1)
g=dict(gx=rng.normal(size=x.shape).astype(np.float32),gv=rng.normal(size=x.shape).astype(np.float32),gprim=rng.normal(size=(4,2,4)).astype(np.float32))
ob=orc.rollout_bwd(x,v,prim,k,mu,actions,g['gx'],g['gv'],g['gprim'],normalize=True,nthreads=4)
hb=_run_hip(fs,x,v,prim,k,mu,actions,g=g,want_lists=False)
eb=_run_hip(es,x,v,prim,k,mu,actions,g=g,want_lists=False)
for key in ('gx','gv','gprim','gactions','gk','gmu'):
    print(key,'fast vs oracle',_rel(hb[key],ob[key]),'exact vs oracle',_rel(eb[key],ob[key]), 'mag',np.abs(ob[key]).max())
# short-window bwd on realistic mid-trajectory state: take state at t=12 as start, T=1
xs,vs,ps=o['x_list'][12],o['v_list'][12],o['prim_list'][12]
a1=actions[13:14]
ob=orc.rollout_bwd(xs,vs,ps,k,mu,a1,g['gx'],g['gv'],g['gprim'],normalize=True,nthreads=4)
hb=_run_hip(fs,xs,vs,ps,k,mu,a1,g=g,want_lists=False)
for key in ('gx','gv','gprim','gactions','gk','gmu'):
    print('T=1 mid',key,'fast vs oracle',_rel(hb[key],ob[key]), 'mag',np.abs(ob[key]).max())
of=orc.rollout_fwd(xs,vs,ps,k,mu,a1); hf=_run_hip(fs,xs,vs,ps,k,mu,a1,want_lists=False)
print('T=1 mid fwd x',_rel(hf['x'],of['x']),'v',_rel(hf['v'],of['v']))

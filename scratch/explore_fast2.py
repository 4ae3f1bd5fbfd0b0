import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from conftest import fold_cloth1_mask, make_cloth_case
from test_cloth_gpu import Conf, _run_hip, _rel
from unidom_amd.engine.cloth_simulator import ClothSimulator
from oracle.pyoracle import ClothOracle
mask=fold_cloth1_mask()
for S in (1,2,5,10,50):
  for normalize in (False,True):
    class C(Conf): substeps=S
    fs=ClothSimulator(C(),2,lambda x,v,i,j:v,mask)
    orc=ClothOracle(mask,substeps=S)
    rng=np.random.default_rng(3)
    B,T=2,2
    x,v,prim,k,mu,actions=make_cloth_case(rng,B,T,deform=0.0005,v_scale=0.01)
    P=512
    g=dict(gx=rng.normal(size=(B,P,3)).astype(np.float32),gv=rng.normal(size=(B,P,3)).astype(np.float32),gprim=rng.normal(size=(B,2,4)).astype(np.float32))
    d=lambda a: np.asarray(a,np.float64)
    o=orc.rollout_bwd(d(x),d(v),d(prim),d(k),d(mu),d(actions),d(g['gx']),d(g['gv']),d(g['gprim']),normalize=normalize)
    of=orc.rollout_fwd(x,v,prim,k,mu,actions,want_grasp=True)
    h=_run_hip(fs,x,v,prim,k,mu,actions,g=g,want_lists=False,normalize=normalize)
    print('S',S,'norm',normalize,'grasp',int(of['grasp'].sum()),'fwd x %.2e v %.2e'%(_rel(h['x'],of['x']),_rel(h['v'],of['v'])),' bwd',' '.join('%s %.2e'%(kk,_rel(h[kk],o[kk])) for kk in ('gx','gv','gprim','gactions','gk','gmu')))

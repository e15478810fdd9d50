import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from test_step_gpu import _mk
from oracle import step as S, tape as tp
gpu=torch.device('cuda:0')
ocfg,P0,cfg,tr=_mk('cifar10',64,0,'ali',None,128,True,False,gpu)
feed=S.make_feed(ocfg,np.random.default_rng(11),'ali')
Pt={k:tp.T(v.astype(np.float64)) for k,v in P0.items()}
oout=S.forward(ocfg,Pt,feed,'ali')
tr.set_feed(feed); out=tr.model.forward(tr.feed)
for which in ('gen','disc'):
    opt=out[which+'_train_op'].optimizer
    names=[p.param_name for p in opt.params]
    grads=torch.autograd.grad(out[which+'_cost'],opt.params,allow_unused=True,retain_graph=True)
    ogs=tp.grad(oout[which+'_cost'],[Pt[n] for n in names])
    for n,g,og in zip(names,grads,ogs):
        if og is None: continue
        ref=og.v; err=np.abs(g.cpu().numpy().reshape(ref.shape)-ref).max()
        print('%-5s %-28s relerr %.2e  max %.3e'%(which,n,err/np.abs(ref).max(),np.abs(ref).max()))
for k in ('fake_x','q_z','p_z'):
    a=out[k].detach().cpu().numpy(); r=oout[k].v
    print(k, 'relerr %.2e'%(np.abs(a-r).max()/np.abs(r).max()), 'max', np.abs(r).max())
for k in ('disc_fake','disc_real'):
    a=out[k].detach().cpu().numpy(); r=oout[k].v
    print(k, 'abs err %.2e'%np.abs(a-r).max(), 'max', np.abs(r).max())
# gradient wrt fake_x of gen cost
gf,=torch.autograd.grad(out['gen_cost'],[out['fake_x']],retain_graph=True)
ogf,=tp.grad(oout['gen_cost'],[oout['fake_x']])
print('d gen/d fake_x relerr %.2e'%(np.abs(gf.cpu().numpy()-ogf.v).max()/np.abs(ogf.v).max()))

import sys, numpy as np, torch, torch.nn.functional as TF
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from test_step_gpu import _mk
from oracle import step as S, tape as tp
from graphical_gan_amd import _lib
import os
gpu=torch.device('cuda:0')
fuse = os.environ.get('FUSE','1')=='1'
ocfg,P0,cfg,tr=_mk('cifar10',64,0,'ali',None,128,fuse,False,gpu)
if os.environ.get('NAIVE'): _lib.load().ggan_set_naive(1)
feed=S.make_feed(ocfg,np.random.default_rng(11),'ali')
Pt={k:tp.T(v.astype(np.float64)) for k,v in P0.items()}
oout=S.forward(ocfg,Pt,feed,'ali')
tr.set_feed(feed); out=tr.model.forward(tr.feed)
gf,=torch.autograd.grad(out['gen_cost'],[out['fake_x']],retain_graph=True)
ogf,=tp.grad(oout['gen_cost'],[oout['fake_x']])
a=gf.cpu().numpy().astype(np.float64); r=ogf.v
print('prod vs oracle: max rel %.3e  l2 rel %.3e'%(np.abs(a-r).max()/np.abs(r).max(), np.linalg.norm(a-r)/np.linalg.norm(r)))
# third opinion: torch CPU float64 on the D-fake branch
T={k:torch.tensor(v.astype(np.float64)) for k,v in P0.items() if k.startswith('Discriminator')}
x=torch.tensor(oout['fake_x'].v, requires_grad=True); z=torch.tensor(oout['p_z'].v)
def conv(x,name): return TF.conv2d(TF.pad(x,(1,2,1,2)), T[name+'.Filters'].permute(3,2,0,1), T[name+'.Biases'], stride=2)
lrelu=lambda t: torch.maximum(0.2*t,t)
o=x.view(-1,3,32,32)
for i in (1,2,3): o=lrelu(conv(o,'Discriminator.%d'%i))
zo=lrelu(z@T['Discriminator.z1.W']+T['Discriminator.z1.b'])
o=torch.cat([o.reshape(64,-1),zo],1)
o=lrelu(o@T['Discriminator.zx1.W']+T['Discriminator.zx1.b'])
d=(o@T['Discriminator.Output.W']+T['Discriminator.Output.b']).view(-1)
loss=TF.binary_cross_entropy_with_logits(d,torch.ones_like(d))
g,=torch.autograd.grad(loss,[x])
t=g.numpy()
print('torch64 vs oracle: max rel %.3e'%(np.abs(t-r).max()/np.abs(r).max()))
print('prod vs torch64: max rel %.3e'%(np.abs(a-t).max()/np.abs(t).max()))
e=np.abs(a-r); idx=np.unravel_index(np.argsort(e.ravel())[-5:], e.shape); print(idx, e[idx], r[idx])
print('frac > 1e-4*max', (e>1e-4*np.abs(r).max()).mean())

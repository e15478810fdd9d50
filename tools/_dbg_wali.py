import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from graphical_gan_amd import tflib as lib, optim, functional as F
from graphical_gan_amd.engine import Trainer
from graphical_gan_amd.models import Config
np.random.seed(0)
cfg = Config('cifar10', batch_size=6, mode='wali-gp', dim=8, dim_latent=16)
tr = Trainer(cfg, device='cuda:0', graph=False)
ring = tr.model.synthetic_ring(torch.device('cuda:0'), n=2)
tr.set_batch(ring[0])
tr._sample_noise()
out = tr.model.forward(tr.feed, 'disc', tr.model.forward_nets(tr.feed))
op = out['disc_train_op']; opt = op.optimizer
orig = F._fused_conv_backward
def spy(ctx, gy, x, w, y):
    print('fused_conv_bwd N=%d Ci=%d Co=%d needs=%s |gy|max=%.3g w_is_leaf_second=%s' % (x.shape[0], x.shape[1], gy.shape[1], ctx.needs_input_grad[:3], float(gy.abs().max()), getattr(w, 'param_name', None)))
    return orig(ctx, gy, x, w, y)
F._fused_conv_backward = spy
with F.defer_wgrad_reduce(True) as d:
    grads = opt.compute_gradients(op.cost)
    reg = F._DEFER[0]
    print('registered', sorted((hex(k), v[0], v[1]) for k, v in reg.items()))
    for p, g in zip(opt.params, grads):
        if 'Filters' in p.param_name:
            gs = g if isinstance(g, tuple) else (g,)
            print(p.param_name, [(hex(x.data_ptr()), tuple(x.shape), x.is_contiguous()) if x is not None else None for x in gs])
    F._DEFER[0] = {}

"""The gan_inference_cifar10.py (MODE='ali') training step restated with PyTorch-CPU primitives composed to TF semantics.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED (the reference cannot run here).  Two uses:
  * an independent third opinion next to the numpy tape (tests/test_oracle_cpu.py: float64, costs to 1e-12, gradients to 1e-10);
  * the CPU baseline of bench.py: oneDNN convolutions on every host core are the closest stand-in for an optimised TensorFlow
    CPU build of the reference, far faster than the numpy restatement.
TF semantics used: SAME padding of the stride-2 5x5 conv = pad (1,2) on both spatial axes (App. A.1); Deconv2D = full transposed
conv cropped [1:1+2H] (A.2); BatchNorm with biased batch variance, eps 1e-5 (A.3); TF-Adam (A.5)."""
import numpy as np
import torch
import torch.nn.functional as F


class Step(object):
    def __init__(self, cfg, params, dtype=torch.float32):
        assert cfg.dataset == 'cifar10' and not cfg.K and cfg.bn
        self.cfg, self.dtype = cfg, dtype
        self.T = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=not k.endswith(('moving_mean', 'moving_variance')))
                  for k, v in params.items()}
        self.gen_names = [n for n in self.T if ('Generator' in n or 'Extractor' in n) and self.T[n].requires_grad]
        self.disc_names = [n for n in self.T if 'Discriminator' in n]
        self.adam = {r: dict(t=0, m={}, v={}) for r in ('gen', 'disc')}

    # ---- layers -----------------------------------------------------------------------------------------------------
    def conv(self, x, name):
        T = self.T
        return F.conv2d(F.pad(x, (1, 2, 1, 2)), T[name + '.Filters'].permute(3, 2, 0, 1), T[name + '.Biases'], stride=2)

    def deconv(self, x, name):
        T, H = self.T, x.shape[2]
        y = F.conv_transpose2d(x, T[name + '.Filters'].permute(3, 2, 0, 1), stride=2)[:, :, 1:1 + 2 * H, 1:1 + 2 * H]
        return y + T[name + '.Biases'].view(1, -1, 1, 1)

    def bn(self, x, name, axes):
        T = self.T
        m = x.mean(axes, keepdim=True)
        v = ((x - m) ** 2).mean(axes, keepdim=True)
        shp = [1 if i in axes else s for i, s in enumerate(x.shape)]
        return T[name + '.scale'].view(shp) * (x - m) / torch.sqrt(v + 1e-5) + T[name + '.offset'].view(shp)

    # ---- cost graph (gan_inference_cifar10.py:133-255, 261-292) ------------------------------------------------------
    def costs(self, real_x, p_z_noise):
        c, T = self.cfg, self.T
        B = real_x.shape[0]
        lrelu = lambda x: torch.maximum(0.2 * x, x)
        real = torch.as_tensor(real_x, dtype=self.dtype).view(-1, 3, 32, 32)
        e = lrelu(self.conv(real, 'Extractor.1'))
        e = lrelu(self.bn(self.conv(e, 'Extractor.2'), 'Extractor.BN2', (0, 2, 3)))
        e = lrelu(self.bn(self.conv(e, 'Extractor.3'), 'Extractor.BN3', (0, 2, 3)))
        q_z = e.reshape(B, -1) @ T['Extractor.Output.W'] + T['Extractor.Output.b']
        p_z = torch.as_tensor(p_z_noise, dtype=self.dtype)
        g = torch.relu(self.bn(p_z @ T['Generator.Input.W'] + T['Generator.Input.b'], 'Generator.BN1', (0,)))
        g = g.view(B, -1, 4, 4)
        g = torch.relu(self.bn(self.deconv(g, 'Generator.2'), 'Generator.BN2', (0, 2, 3)))
        g = torch.relu(self.bn(self.deconv(g, 'Generator.3'), 'Generator.BN3', (0, 2, 3)))
        fake = torch.tanh(self.deconv(g, 'Generator.5'))

        def D(x, z):
            o = x
            for i in (1, 2, 3):
                o = lrelu(self.conv(o, 'Discriminator.%d' % i))
            zo = lrelu(z @ T['Discriminator.z1.W'] + T['Discriminator.z1.b'])
            o = torch.cat([o.reshape(B, -1), zo], 1)
            o = lrelu(o @ T['Discriminator.zx1.W'] + T['Discriminator.zx1.b'])
            return (o @ T['Discriminator.Output.W'] + T['Discriminator.Output.b']).view(-1)
        df, dr = D(fake, p_z), D(real, q_z)
        bce = F.binary_cross_entropy_with_logits
        gen = bce(df, torch.ones_like(df)) + bce(dr, torch.zeros_like(dr))
        disc = bce(df, torch.zeros_like(df)) + bce(dr, torch.ones_like(dr))
        return gen, disc

    # ---- one session.run ----------------------------------------------------------------------------------------------
    def step(self, which, real_x, p_z_noise, lr=2e-4, b1=0.5, b2=0.999, eps=1e-8):
        names = self.gen_names if which == 'gen' else self.disc_names
        gen, disc = self.costs(real_x, p_z_noise)
        cost = gen if which == 'gen' else disc
        grads = torch.autograd.grad(cost, [self.T[n] for n in names], allow_unused=True)
        st = self.adam[which]
        st['t'] += 1
        lr_t = lr * np.sqrt(1 - b2 ** st['t']) / (1 - b1 ** st['t'])
        with torch.no_grad():
            for n, g in zip(names, grads):
                if g is None:
                    continue
                m = st['m'].setdefault(n, torch.zeros_like(g))
                v = st['v'].setdefault(n, torch.zeros_like(g))
                m.mul_(b1).add_(g, alpha=1 - b1)
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                self.T[n].sub_(lr_t * m / (v.sqrt() + eps))
        return float(cost.detach())

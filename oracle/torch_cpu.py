"""The image scripts' training step (gan_inference_* MODE 'ali' / 'wali-gp', gmgan_inference_* MODE 'local_ep') restated with
PyTorch-CPU primitives composed to TF semantics.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED (the reference cannot run here).  Three uses:
  * an independent third opinion next to the numpy tape (tests/test_oracle_cpu.py: float64, costs to 1e-12, gradients to 1e-10);
  * the generator of the FULL-SIZE golden fixtures (tests/golden/make_golden.py): in float64 it evaluates a batch-64 step of
    every BASELINE configuration in seconds, where the numpy tape needs minutes;
  * the CPU baseline of bench.py: oneDNN convolutions on every host core are the closest stand-in for an optimised TensorFlow
    CPU build of the reference, far faster than the numpy restatement.
TF semantics used: SAME padding of the stride-2 5x5 conv = pad (1,2) on both spatial axes, (2,2) for the 7 -> 4 layer of the
MNIST nets (App. A.1); Deconv2D = full transposed conv cropped [1:1+2H] (A.2); BatchNorm with biased batch variance, eps 1e-5
(A.4); TF-Adam (A.5); Gumbel-softmax assignment of the mixture prior (A.8); the gradient penalty without an epsilon inside the
square root (A.6).  Net wiring: gan_inference_cifar10.py:133-255,261-366, gmgan_inference_cifar10.py:114-301,341-398,
gmgan_inference_face.py:108-200, gmgan_inference_mnist.py:166-300."""
import numpy as np
import torch
import torch.nn.functional as F


def _same_pad(size, k=5, s=2):
    out = -(-size // s)
    tot = max((out - 1) * s + k - size, 0)
    return tot // 2, tot - tot // 2


class Step(object):
    def __init__(self, cfg, params, dtype=torch.float32, mode='ali'):
        assert mode in ('ali', 'wali-gp', 'local_ep') and bool(cfg.K) == (mode == 'local_ep')
        self.cfg, self.dtype, self.mode = cfg, dtype, mode
        self.T = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=not k.endswith(('moving_mean', 'moving_variance')))
                  for k, v in params.items()}
        self.gen_names = [n for n in self.T if ('Generator' in n or 'Extractor' in n) and self.T[n].requires_grad]
        self.disc_names = [n for n in self.T if 'Discriminator' in n and self.T[n].requires_grad]
        self.adam = {r: dict(t=0, m={}, v={}) for r in ('gen', 'disc')}
        self.critic_iters = 5 if mode == 'wali-gp' else 1
        # tflib/objs/gan_inference.py:34-43 (wali_gp: 1e-4, .5, .9) vs the scripts' LR / BETA1 with the default beta2
        self.hp = dict(lr=1e-4, b1=0.5, b2=0.9) if mode == 'wali-gp' else dict(lr=2e-4, b1=0.5, b2=0.999)
        # margins: set to [] to record, per ReLU / LeakyReLU on a Linear-layer output, (kind, min |pre-activation| / rms(row)): how far the
        # forward pass stays from a kink where fp32 rounding could pick the other branch (tests/golden/make_golden_full.py)
        self.margins = None
        # kinks: set to {} to record, per ReLU / LeakyReLU on an image-shaped tensor, the KINK_K units nearest their kink (flat positions,
        # float64 pre-activations, the layer's rms); force: key -> (flat positions, bool "take the positive branch"): those units take the
        # given branch whatever the sign of their pre-activation -- the float64 evaluation a float32 run agrees with when rounding put
        # exactly those units on the other side of zero (tests/golden/make_golden_full.py, tests/test_golden_full_gpu.py)
        self.kinks = None
        self.force = None
        self._tag = ''

    KINK_K = 48

    def _act(self, key, x, slope):
        if self.kinks is not None and x.dim() == 4:
            with torch.no_grad():
                f = x.detach().reshape(-1).double()
                k = min(self.KINK_K, f.numel())
                _, idx = torch.topk(f.abs(), k, largest=False)
                idx = idx.sort().values
                self.kinks[key + self._tag] = dict(idx=idx.numpy().astype(np.int64), val=f[idx].numpy(), rms=float(f.pow(2).mean().sqrt()),
                                                    shape=tuple(x.shape))
        y = torch.maximum(slope * x, x) if slope else torch.relu(x)
        f = self.force.get(key + self._tag) if self.force else None
        if f is None:
            return y
        idx, pos = f
        m = (x.detach() > 0).reshape(-1).clone()
        m[torch.as_tensor(np.asarray(idx, dtype=np.int64))] = torch.as_tensor(np.asarray(pos, dtype=bool))
        m = m.view_as(x).to(x.dtype)
        return x * (m + slope * (1 - m))

    # ---- layers -----------------------------------------------------------------------------------------------------
    def conv(self, x, name):
        T = self.T
        pt, pb = _same_pad(x.shape[2])
        pl, pr = _same_pad(x.shape[3])
        return F.conv2d(F.pad(x, (pl, pr, pt, pb)), T[name + '.Filters'].permute(3, 2, 0, 1), T[name + '.Biases'], stride=2)

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

    def lin(self, x, name):
        return x @ self.T[name + '.W'] + self.T[name + '.b']

    # ---- nets -------------------------------------------------------------------------------------------------------
    def _margin(self, tag, x):
        if self.margins is not None and x.dim() == 2:
            with torch.no_grad():
                rms = x.pow(2).mean(1, keepdim=True).sqrt() + 1e-30
                self.margins.append((tag, float((x.abs() / rms).min())))

    def lrelu(self, x, key=None):
        self._margin('lrelu', x)
        if key is not None:
            return self._act(key, x, 0.2)
        return torch.maximum(0.2 * x, x)

    def Extractor(self, x):
        c = self.cfg
        e = x.reshape(-1, c.C, c.S, c.S)
        for i in range(c.nl):
            e = self.conv(e, 'Extractor.%d' % (i + 1))
            key = 'Extractor.%d' % (i + 1)
            if c.bn and i > 0:
                e = self.bn(e, 'Extractor.BN%d' % (i + 1), (0, 2, 3))
                key = 'Extractor.BN%d' % (i + 1)
            e = self.lrelu(e, key)
        return self.lin(e.reshape(-1, c.flat), 'Extractor.Output')

    def Generator(self, z):
        c = self.cfg
        g = self.lin(z, 'Generator.Input')
        if c.bn:
            g = self.bn(g, 'Generator.BN1', (0,))
        self._margin('relu', g)
        g = torch.relu(g).view(-1, c.top, 4, 4)
        names = ['2', '3', '4', '5'] if c.nl == 4 else ['2', '3', '5']
        for i, nm in enumerate(names):
            g = self.deconv(g, 'Generator.' + nm)
            if i < len(names) - 1:
                if c.bn:
                    g = self.bn(g, 'Generator.BN' + nm, (0, 2, 3))
                g = self._act('Generator.' + ('BN' if c.bn else '') + nm, g, 0.0)
                if c.dataset == 'mnist' and nm == '2':
                    g = g[:, :, :7, :7]
        g = torch.tanh(g) if c.out_act == 'tanh' else torch.sigmoid(g)
        return g.reshape(-1, c.output_dim)

    def Discriminator(self, x, z):
        c = self.cfg
        o = x.reshape(-1, c.C, c.S, c.S)
        deep = getattr(c, 'critic_deep', False)          # gan_inference_mnist.py:215-250 (oracle/nets.py Discriminator)
        for i in range(c.nl):
            o = self.conv(o, 'Discriminator.%d' % (i + 1))
            key = 'Discriminator.%d' % (i + 1)
            if deep and c.bn and i > 0:
                o = self.bn(o, 'Discriminator.BN%d' % (i + 1), (0, 2, 3))
                key = 'Discriminator.BN%d' % (i + 1)
            o = self.lrelu(o, key)
        zo = self.lrelu(self.lin(z, 'Discriminator.z1'))
        if deep:
            zo = self.lrelu(self.lin(zo, 'Discriminator.2'))
        o = torch.cat([o.reshape(-1, c.flat), zo], 1)
        o = self.lrelu(self.lin(o, 'Discriminator.zx1'))
        if deep:
            o = self.lrelu(self.lin(o, 'Discriminator.zx2'))
        return self.lin(o, 'Discriminator.Output').view(-1)

    def HyperDiscriminator(self, z, k):
        o = torch.cat([z, k], 1)
        for nm in ('HyperInput', 'Hyper2', 'Hyper3'):
            o = self.lrelu(self.lin(o, 'Discriminator.' + nm))
        return self.lin(o, 'Discriminator.HyperOutput').view(-1)

    def HyperExtractor(self, z, gumbel_u):
        c, mu = self.cfg, self.T['Generator.Hyper.Mu']
        logits = -0.5 * ((z[:, None, :] - mu[None, :, :]) ** 2).sum(2) + float(np.log(np.float32(1.0) / np.float32(c.K)))
        u = torch.as_tensor(np.asarray(gumbel_u), dtype=self.dtype)
        g = -torch.log(-torch.log(u + 1e-20) + 1e-20)
        return logits, torch.softmax((logits + g) / c.temp, 1)

    # ---- cost graph --------------------------------------------------------------------------------------------------
    def real_x(self, feed):
        c = self.cfg
        if c.dataset == 'mnist':
            return torch.as_tensor(np.asarray(feed['real_x']), dtype=self.dtype)
        xi = torch.as_tensor(np.asarray(feed['real_x_int']), dtype=self.dtype)
        if c.dataset == 'face':
            return 2 * (xi / 256. - .5) + torch.as_tensor(np.asarray(feed['dequant_u']), dtype=self.dtype)
        return 2 * (xi / 255. - .5)

    def forward(self, feed, which=None):
        """feed: oracle.step.make_feed layout.  which: 'gen' leaves the gradient penalty out (it is not part of gen_cost)."""
        c, mode = self.cfg, self.mode
        real = self.real_x(feed)
        q_z = self.Extractor(real)
        noise = torch.as_tensor(np.asarray(feed['p_z_noise']), dtype=self.dtype)
        out = dict(real_x=real, q_z=q_z)
        if c.K:
            onehot = torch.zeros(c.B, c.K, dtype=self.dtype)
            onehot[torch.arange(c.B), torch.as_tensor(np.asarray(feed['k_idx']))] = 1
            _, q_k = self.HyperExtractor(q_z, feed['gumbel_u'])
            p_z = onehot @ self.T['Generator.Hyper.Mu'] + noise
        else:
            p_z = noise
        fake = self.Generator(p_z)
        out.update(p_z=p_z, fake_x=fake)
        bce = F.binary_cross_entropy_with_logits
        def D(tag, x, z):
            self._tag = '@' + tag
            try:
                return self.Discriminator(x, z)
            finally:
                self._tag = ''
        if c.K:
            d_fake = [self.HyperDiscriminator(p_z, onehot), D('fake', fake, p_z)]
            d_real = [self.HyperDiscriminator(q_z, q_k), D('real', real, q_z)]
            gen = sum(bce(f, torch.ones_like(f)) + bce(r, torch.zeros_like(r)) for f, r in zip(d_fake, d_real)) / len(d_fake)
            disc = sum(bce(f, torch.zeros_like(f)) + bce(r, torch.ones_like(r)) for f, r in zip(d_fake, d_real)) / len(d_fake)
        else:
            d_fake, d_real = D('fake', fake, p_z), D('real', real, q_z)
            if mode == 'ali':
                gen = bce(d_fake, torch.ones_like(d_fake)) + bce(d_real, torch.zeros_like(d_real))
                disc = bce(d_fake, torch.zeros_like(d_fake)) + bce(d_real, torch.ones_like(d_real))
            else:                                   # wali-gp: tflib/objs/gan_inference.py:28-32, gan_inference_cifar10.py:353-364
                gen = -d_fake.mean() + d_real.mean()
                disc = d_fake.mean() - d_real.mean()
                if which != 'gen':
                    a = torch.as_tensor(np.asarray(feed['alpha']), dtype=self.dtype).view(-1, 1)
                    x_hat = real + a * (fake - real)
                    z_hat = q_z + a * (p_z - q_z)
                    d_hat = D('hat', x_hat, z_hat)
                    (g,) = torch.autograd.grad(d_hat.sum(), [x_hat], create_graph=True)
                    gp = 10.0 * ((torch.sqrt((g ** 2).sum(1)) - 1.0) ** 2).mean()
                    disc = disc + gp
                    out['gradient_penalty'] = gp
        out.update(disc_fake=d_fake, disc_real=d_real, gen_cost=gen, disc_cost=disc)
        return out

    def costs(self, real_x, p_z_noise):
        """(gen_cost, disc_cost) of the plain CIFAR 'ali' step from a scaled float minibatch (kept for the first cross-check)."""
        c = self.cfg
        xi = (np.asarray(real_x, dtype=np.float64) / 2 + .5) * (256. if c.dataset == 'face' else 255.)
        feed = {'real_x_int': xi, 'p_z_noise': p_z_noise, 'dequant_u': np.zeros_like(xi), 'real_x': real_x}
        out = self.forward(feed)
        return out['gen_cost'], out['disc_cost']

    def grads(self, feed, which):
        """cost value and name -> gradient (numpy, None where the cost does not reach the parameter) of one session.run"""
        names = self.gen_names if which == 'gen' else self.disc_names
        out = self.forward(feed, which)
        cost = out[which + '_cost']
        gs = torch.autograd.grad(cost, [self.T[n] for n in names], allow_unused=True)
        return out, cost, dict(zip(names, gs))

    # ---- one session.run ----------------------------------------------------------------------------------------------
    def run(self, which, feed, eps=1e-8):
        lr, b1, b2 = self.hp['lr'], self.hp['b1'], self.hp['b2']
        out, cost, grads = self.grads(feed, which)
        st = self.adam[which]
        st['t'] += 1
        lr_t = lr * np.sqrt(1 - b2 ** st['t']) / (1 - b1 ** st['t'])
        with torch.no_grad():
            for n, g in grads.items():
                if g is None:
                    continue
                m = st['m'].setdefault(n, torch.zeros_like(g))
                v = st['v'].setdefault(n, torch.zeros_like(g))
                m.mul_(b1).add_(g, alpha=1 - b1)
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                self.T[n].sub_(lr_t * m / (v.sqrt() + eps))
        return float(cost.detach())

    def step(self, which, real_x, p_z_noise, **kw):
        """the original entry point: one 'ali' session.run from a scaled float minibatch"""
        c = self.cfg
        xi = (np.asarray(real_x, dtype=np.float64) / 2 + .5) * (256. if c.dataset == 'face' else 255.)
        return self.run(which, {'real_x_int': xi, 'p_z_noise': p_z_noise, 'dequant_u': np.zeros_like(xi), 'real_x': real_x})

    def iteration(self, it, feeds):
        """gmgan_inference_cifar10.py:480-494: iteration 0 = critic steps only; 1 + CRITIC_ITERS feeds otherwise"""
        res = {}
        if it > 0:
            res['gen_cost'] = self.run('gen', next(feeds))
        for _ in range(self.critic_iters):
            res['disc_cost'] = self.run('disc', next(feeds))
        return res

"""The state-space scripts' training step (ssgan_inference_moving_mnist.py / ssgan_inference_chairs.py) restated with PyTorch-CPU primitives
composed to TF semantics -- the counterpart of oracle/torch_cpu.py for BASELINE configs[4].

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED at the TF primitives (the reference cannot run here).  Two uses:
  * an independent opinion next to the numpy tape (tests/test_oracle_cpu.py: float64, costs and every gradient against oracle/ssgan.py);
  * the CPU baseline of bench.py for the state-space rows: the SAME iteration at the SAME size on every granted host core (oneDNN
    convolutions), where the numpy tape only manages a 2-sequence sample.
Covered: MODE 'local_ep' (weighted_local_epce, the script default) and MODE 'ali' with ALI_MODE 'concat_x' / '3dcnn'; POS_MODE
'naive_mean_field' (default); OP_DYN_MODE 'res' / 'res_w'; BN flags off (default).  Net wiring: ssgan_inference_moving_mnist.py:98-349 (nets),
:352-449 (sequence critics), :510-547 (costs), :78-79 (ratios); Conv3D: tflib/ops/conv3d.py:33-48."""
import numpy as np
import torch
import torch.nn.functional as F

from .torch_cpu import _same_pad


def _pad_k(size, k, s):
    out = -(-size // s)
    tot = max((out - 1) * s + k - size, 0)
    return tot // 2, tot - tot // 2


class Step(object):
    def __init__(self, cfg, params, dtype=torch.float32):
        assert cfg.pos_mode == 'naive_mean_field' and cfg.mode in ('local_ep', 'ali')
        assert not cfg.seq_critic or cfg.ali_mode in ('concat_x', '3dcnn')
        self.cfg, self.dtype = cfg, dtype
        self.T = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in params.items()}
        self.gen_names = [n for n in self.T if 'Generator' in n] + [n for n in self.T if 'Extractor' in n]
        self.disc_names = [n for n in self.T if 'Discriminator' in n]
        self.adam = {r: dict(t=0, m={}, v={}) for r in ('gen', 'disc')}
        self.critic_iters = cfg.critic_iters

    # ---- layers ---------------------------------------------------------------------------------------------------------
    def lin(self, x, name):
        return x @ self.T[name + '.W'] + self.T[name + '.b']

    def conv(self, x, name):
        pt, pb = _same_pad(x.shape[2])
        pl, pr = _same_pad(x.shape[3])
        return F.conv2d(F.pad(x, (pl, pr, pt, pb)), self.T[name + '.Filters'].permute(3, 2, 0, 1), self.T[name + '.Biases'], stride=2)

    def deconv(self, x, name):
        H = x.shape[2]
        y = F.conv_transpose2d(x, self.T[name + '.Filters'].permute(3, 2, 0, 1), stride=2)[:, :, 1:1 + 2 * H, 1:1 + 2 * H]
        return y + self.T[name + '.Biases'].view(1, -1, 1, 1)

    def conv3d(self, x, name, sl):
        """x [B, C, L, H, W]; filter [4, 4, 4, Cin, Cout] (length, height, width); strides (sl, 2, 2); SAME"""
        w = self.T[name + '.Filters'].permute(4, 3, 0, 1, 2)
        pl = _pad_k(x.shape[2], 4, sl)
        ph = _pad_k(x.shape[3], 4, 2)
        pw = _pad_k(x.shape[4], 4, 2)
        x = F.pad(x, (pw[0], pw[1], ph[0], ph[1], pl[0], pl[1]))
        return F.conv3d(x, w, self.T[name + '.Biases'], stride=(sl, 2, 2))

    @staticmethod
    def lrelu(x):
        return torch.maximum(0.2 * x, x)

    # ---- nets -----------------------------------------------------------------------------------------------------------
    def expand_labels(self, y):
        c = self.cfg
        return y.view(c.B, 1, c.n_c).expand(c.B, c.LEN, c.n_c).reshape(c.B * c.LEN, c.n_c)

    def z_rows(self, z_g, z_l, labels):
        c = self.cfg
        zg = z_g.view(c.B, 1, c.dim_g).expand(c.B, c.LEN, c.dim_g)
        lab = self.expand_labels(labels).view(c.B, c.LEN, c.n_c)
        return torch.cat([zg, z_l.reshape(c.B, c.LEN, c.dim_l), lab], 2).reshape(c.B * c.LEN, c.dim_g + c.dim_l + c.n_c)

    def operator(self, name, a, b, res_src):
        c = self.cfg
        out = self.lrelu(self.lin(torch.cat([a, b], 1), name + '.Input'))
        out = self.lrelu(self.lin(out, name + '.1'))
        out = self.lin(out, name + '.Output')
        if c.op_dyn_mode == 'res':
            out = out + res_src
        elif c.op_dyn_mode == 'res_w':
            out = out + self.lin(res_src, name + '.ZW')
        return out

    def DynamicGenerator(self, z_l_0, epsilon):
        c = self.cfg
        zs = [z_l_0]
        for _ in range(c.LEN - 1):
            zs.append(self.operator('Generator.Dynamic', zs[-1], epsilon, zs[-1]))
        return torch.cat(zs, 1).view(c.B, c.LEN, c.dim_l)

    def Generator(self, z_g, z_l, labels):
        c = self.cfg
        out = torch.relu(self.lin(self.z_rows(z_g, z_l, labels), 'Generator.Input')).view(c.B * c.LEN, 8 * c.dim, 4, 4)
        for nm in ('2', '3', '4'):
            out = torch.relu(self.deconv(out, 'Generator.' + nm))
        return torch.tanh(self.deconv(out, 'Generator.5')).reshape(c.B, c.LEN, c.output_dim)

    def conv_stack(self, pre, x):
        for i in range(4):
            x = self.lrelu(self.conv(x, '%s.%d' % (pre, i + 1)))
        return x

    def Extractor(self, x, labels):
        c = self.cfg
        out = self.conv_stack('Extractor', x.reshape(c.B * c.LEN, c.C, 64, 64)).reshape(c.B * c.LEN, c.flat)
        return self.lin(torch.cat([out, self.expand_labels(labels)], 1), 'Extractor.Output').view(c.B, c.LEN, c.dim_l)

    def G_Extractor(self, x, labels):
        c = self.cfg
        out = self.conv_stack('Extractor.G', x.reshape(c.B, c.C * c.LEN, 64, 64)).reshape(c.B, c.flat)
        return self.lin(torch.cat([out, labels], 1), 'Extractor.G.Output')

    def Discriminator(self, x, z_g, z_l, labels):
        c = self.cfg
        out = self.conv_stack('Discriminator', x.reshape(c.B * c.LEN, c.C, 64, 64)).reshape(c.B * c.LEN, c.flat)
        z_out = self.lrelu(self.lin(self.z_rows(z_g, z_l, labels), 'Discriminator.z1'))
        out = self.lrelu(self.lin(torch.cat([out, z_out, self.expand_labels(labels)], 1), 'Discriminator.zx1'))
        return self.lin(out, 'Discriminator.Output').view(c.B * c.LEN)

    def SequenceDiscriminator(self, x, z_g, z_l, labels):
        c = self.cfg
        z = torch.cat([z_g, z_l.reshape(c.B, c.LEN * c.dim_l), labels], 1)
        if c.ali_mode == '3dcnn':
            from .ssgan import conv3d_plan
            out = x.reshape(c.B, 1, c.LEN, 64, 64)
            for i, (_, _, sl) in enumerate(conv3d_plan(c)):
                out = self.lrelu(self.conv3d(out, 'Discriminator.%d' % (i + 1), sl))
            # the reference flattens the NLHWC volume: [B, L', H', W', C] -> [B, flat]
            out = out.permute(0, 2, 3, 4, 1).reshape(c.B, c.flat)
        else:
            out = self.conv_stack('Discriminator', x.reshape(c.B, c.C * c.LEN, 64, 64)).reshape(c.B, c.flat)
        z_out = self.lrelu(self.lin(z, 'Discriminator.z1'))
        out = self.lrelu(self.lin(torch.cat([out, z_out], 1), 'Discriminator.zx1'))
        return self.lin(out, 'Discriminator.Output').view(c.B)

    def mlp_critic(self, pre, x):
        out = self.lrelu(self.lin(x, pre + '.Input'))
        out = self.lrelu(self.lin(out, pre + '.2'))
        out = self.lrelu(self.lin(out, pre + '.3'))
        return self.lin(out, pre + '.Output').view(-1)

    # ---- cost graph -------------------------------------------------------------------------------------------------------
    def forward(self, feed):
        c, dt = self.cfg, self.dtype
        T = lambda k: torch.as_tensor(np.asarray(feed[k]), dtype=dt)
        real_x = 2 * (T('real_x_unit') / c.x_div - .5)
        real_y, p_y = T('real_y'), T('p_y')
        q_z_l = self.Extractor(real_x, real_y)
        q_z_g = self.G_Extractor(real_x, real_y)
        p_z_l = self.DynamicGenerator(T('p_z_l_0'), T('epsilon'))
        p_z_g = T('p_z_g')
        fake_x = self.Generator(p_z_g, p_z_l, p_y)
        bce = F.binary_cross_entropy_with_logits
        pair = lambda f, r, lf, lr: bce(f, torch.full_like(f, lf)) + bce(r, torch.full_like(r, lr))
        if c.seq_critic:
            d_fake = self.SequenceDiscriminator(fake_x, p_z_g, p_z_l, p_y)
            d_real = self.SequenceDiscriminator(real_x, q_z_g, q_z_l, real_y)
            return dict(fake_x=fake_x, gen_cost=pair(d_fake, d_real, 1., 0.), disc_cost=pair(d_fake, d_real, 0., 1.))
        fakes, reals = [], []
        for i in range(c.LEN - 1):
            fakes.append(self.mlp_critic('Discriminator.Dynamic', torch.cat([p_z_l[:, i], p_z_l[:, i + 1]], 1)))
            reals.append(self.mlp_critic('Discriminator.Dynamic', torch.cat([q_z_l[:, i], q_z_l[:, i + 1]], 1)))
        fakes.append(self.mlp_critic('Discriminator.ZG', p_z_g))
        reals.append(self.mlp_critic('Discriminator.ZG', q_z_g))
        fakes.append(self.Discriminator(fake_x, p_z_g, p_z_l, p_y))
        reals.append(self.Discriminator(real_x, q_z_g, q_z_l, real_y))
        gen = disc = 0.0
        for f, r, ratio in zip(fakes, reals, list(c.ratio())):
            gen = float(ratio) * pair(f, r, 1., 0.) + gen
            disc = float(ratio) * pair(f, r, 0., 1.) + disc
        return dict(fake_x=fake_x, gen_cost=gen, disc_cost=disc)

    def grads(self, feed, which):
        names = self.gen_names if which == 'gen' else self.disc_names
        out = self.forward(feed)
        cost = out[which + '_cost']
        gs = torch.autograd.grad(cost, [self.T[n] for n in names], allow_unused=True)
        return out, cost, dict(zip(names, gs))

    # ---- one session.run (TF-Adam) ------------------------------------------------------------------------------
    def run(self, which, feed, eps=1e-8):
        c = self.cfg
        lr, b1, b2 = c.lr, c.beta1, c.beta2
        _, cost, grads = self.grads(feed, which)
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

    def iteration(self, it, feeds):
        """ssgan_inference_moving_mnist.py:625-681: iteration 0 = critic step only; one generator step + CRITIC_ITERS critic steps otherwise"""
        res = {}
        if it > 0:
            res['gen_cost'] = self.run('gen', next(feeds))
        for _ in range(self.critic_iters):
            res['disc_cost'] = self.run('disc', next(feeds))
        return res

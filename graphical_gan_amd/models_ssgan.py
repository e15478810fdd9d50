"""State-space GAN on moving-MNIST sequences: the net + loss wiring of ssgan_inference_moving_mnist.py against this
package's `tflib` (same layer names, so registry keys match the reference's checkpoints).

  hyper-parameters   ssgan_inference_moving_mnist.py:27-56 (MODE='local_ep', POS_MODE, OP_DYN_MODE, BN flags off)
  nets               :98-349      (ImplicitOperator, ConcatOperator, Dynamic{Generator,Extractor}, Generator, Extractor,
                                   G_Extractor, Discriminator, DynamicDiscrminator, ZGDiscrminator)
  losses             :510-547     weighted_local_epce over LEN+1 factors with ratio = [1]*(LEN-1)+[1,LEN] (:78-79)

What is done differently from a literal transcription (same arithmetic):
  * every frame net runs on all B*LEN frames at once (the reference does too) and the LEN-1 transition critics, which
    share one set of weights, are evaluated in ONE call on the stacked pairs [(LEN-1)*B, 2*dim_l] instead of LEN-1 calls;
    their equal-weight BCE terms collapse into one term (sum_i r*mean_B(.) == r*(LEN-1)*mean over the stack);
  * each of the three critics is evaluated once per step on [fake; real] (rows independent: no BatchNorm); generator steps
    hand the critics their weights without gradient edges (tflib.frozen), as TF's var_list does, and run the frame
    critic's conv data-gradient on the fake frames only (grad_rows).
"""
import os

import numpy as np
import torch

from . import functional as F
from . import tflib as lib
from .tflib.ops.act import LRELU, RELU, TANH


class SSConfig(object):
    K = 0

    def __init__(self, batch_size=50, length=16, dim=32, dim_op=256, dim_g=128, dim_l=8, n_c=10,
                 pos_mode='naive_mean_field', op_dyn_mode='res', lr=1e-4, fuse=True, channels=1, dataset='moving_mnist',
                 mode='local_ep', lamb=0.1, ali_mode='concat_x'):
        """defaults: ssgan_inference_moving_mnist.py:26-53.  channels=3, n_c=0, length=31, op_dyn_mode='res_w',
        dataset='chairs': ssgan_inference_chairs.py:28-54 (RGB frames, no class labels)."""
        self.dataset = dataset
        assert mode in ('local_ep', 'local_epce-z', 'ali', 'alice-z'), mode
        self.mode, self.lamb = mode, lamb                      # *-z: + LAMBDA * l2(real_x, G(q_z_g, q_z_l, real_y)) (:549-558)
        # MODE ali / alice-z: ONE critic on the whole sequence (ALI_MODE = 'concat_x' :407-449 | 'concat_z' :451-497 | '3dcnn' :352-405)
        self.seq_critic = mode in ('ali', 'alice-z')
        assert ali_mode in ('concat_x', 'concat_z', '3dcnn'), ali_mode
        if self.seq_critic and ali_mode == '3dcnn':   # the script hard-codes one input channel and the LEN 4 / 16 stride plans
            assert length in (4, 16) and channels == 1, (length, channels)
        self.ali_mode = ali_mode
        self.B, self.LEN, self.dim, self.dim_op = batch_size, length, dim, dim_op
        self.dim_g, self.dim_l, self.dim_t, self.n_c = dim_g, dim_l, dim_l, n_c
        self.S, self.C, self.output_dim = 64, channels, channels * 64 * 64
        self.x_div = 256.0 if dataset == 'chairs' else 1.0     # chairs frames are 0..255: real_x = 2*((x/256.)-.5) (chairs :508)
        self.flat = 4 * 4 * 8 * dim
        assert pos_mode in ('naive_mean_field', 'inverse', 'forward_inverse', 'gsp'), pos_mode
        assert op_dyn_mode in ('res', 'res_w'), op_dyn_mode
        self.pos_mode, self.op_dyn_mode = pos_mode, op_dyn_mode
        self.lr, self.beta1 = lr, 0.5
        self.critic_iters = 1
        self.fuse = fuse

    def ratio(self):
        r = np.asarray([1.0] * (self.LEN - 1) + [1, self.LEN])
        return r * 1.0 / (len(r) + self.LEN - 1)


class StateSpaceGAN(object):
    """forward_nets(feed) / forward(feed, which, nets): the interface engine.Trainer drives."""

    # every conv filter receives ONE gradient contribution per backward pass (each conv net is applied once: the critics see
    # [fake; real] as one batch), so the pack kernel may sum the filter-gradient slabs; the shared-weight Linear operators
    # (applied LEN-1 times) are summed by autograd as usual
    @property
    def single_contribution(self):
        return self.cfg.mode in ('local_ep', 'ali')          # the *-z modes apply the frame generator twice


    def __init__(self, cfg):
        self.cfg = cfg
        # chains of short launches (the transition operator's scan, the latent critics) go to a second stream beside the conv
        # stacks while the Trainer builds a single-graph step (fork_now); eager steps keep the reference's op order
        self._side = None
        self.fork_nets = not os.environ.get('GGAN_NO_FORK_NETS')
        self.fork_now = False

    # ---- engine hooks: static inputs of one session.run ---------------------------------------------------------------
    def feed_buffers(self, device):
        c = self.cfg
        z = lambda *s: torch.zeros(*s, device=device)
        # x_pair: [fake frames ; real frames], each half written in place by its producer (the frame generator's last layer, the input
        # scaling), so that the critics' [fake; real] batch is an alias instead of a 33 MB copy per step (functional.RowSlot / JoinRows)
        return dict(real_x_unit=z(c.B, c.LEN, c.output_dim), real_y=z(c.B, c.n_c), p_z_l_0=z(c.B, c.dim_l),
                    epsilon=z(c.B, c.dim_t), p_z_g=z(c.B, c.dim_g), p_y=z(c.B, c.n_c), x_pair=z(2 * c.B * c.LEN, c.output_dim))

    def sample_noise(self, feed):
        """Fresh noise of one session.run in ONE launch (functional.noise_fill_: the draw number lives on the device, so a captured
        step draws anew on every replay): p_z_l_0, epsilon, p_z_g ~ N(0,1), p_y one-hot rows with a uniform class
        (ssgan_inference_moving_mnist.py:98-101, :510-519)."""
        c = self.cfg
        if 'rng_state' not in feed:
            feed['rng_state'] = F.noise_state(feed['p_z_g'].device)
        specs = [(feed['p_z_l_0'], F.NOISE_NORMAL, 0., 1.), (feed['epsilon'], F.NOISE_NORMAL, 0., 1.), (feed['p_z_g'], F.NOISE_NORMAL, 0., 1.)]
        if c.n_c:
            specs.append((feed['p_y'], F.NOISE_ONEHOT, 0., 0.))
        F.noise_fill_(feed['rng_state'], specs)

    def set_batch(self, feed, batch):
        x, y = batch if isinstance(batch, (tuple, list)) else (batch, None)
        feed['real_x_unit'].copy_(x.reshape(feed['real_x_unit'].shape), non_blocking=True)
        if y is not None and self.cfg.n_c:
            feed['real_y'].copy_(y, non_blocking=True)

    def synthetic_ring(self, device, n=4, seed=1234):
        c, rng, ring = self.cfg, np.random.default_rng(seed), []
        for _ in range(n):
            x = torch.as_tensor(rng.random((c.B, c.LEN, c.output_dim), dtype=np.float32) * np.float32(256.0 if c.x_div > 1 else 1.0))
            y = np.zeros((c.B, c.n_c), np.float32)
            if c.n_c:
                y[np.arange(c.B), rng.integers(0, c.n_c, size=c.B)] = 1
            ring.append((x.to(device), torch.as_tensor(y).to(device)))
        return ring

    # ---- helpers --------------------------------------------------------------------------------------------------------
    def _lin(self, name, nin, nout, x, act=None):
        if act is None:
            return lib.ops.linear.Linear(name, nin, nout, x)
        if self.cfg.fuse:
            return lib.ops.linear.Linear(name, nin, nout, x, activation=act)
        return F.ActFwd.apply(lib.ops.linear.Linear(name, nin, nout, x), act, 0.2)

    def _conv(self, name, cin, cout, x, grad_rows=None):
        if self.cfg.fuse:
            return lib.ops.conv2d.Conv2D(name, cin, cout, 5, x, stride=2, activation=LRELU, grad_rows=grad_rows)
        return F.ActFwd.apply(lib.ops.conv2d.Conv2D(name, cin, cout, 5, x, stride=2, grad_rows=grad_rows), LRELU, 0.2)

    def _deconv(self, name, cin, cout, x, act, out=None):
        if self.cfg.fuse:
            return lib.ops.deconv2d.Deconv2D(name, cin, cout, 5, x, activation=act, out=out)
        return F.ActFwd.apply(lib.ops.deconv2d.Deconv2D(name, cin, cout, 5, x), act, 0.0)

    def expand_labels(self, y):
        c = self.cfg
        return y.unsqueeze(1).expand(c.B, c.LEN, c.n_c).reshape(c.B * c.LEN, c.n_c)

    def _z_rows(self, z_g, z_l, labels):
        c = self.cfg
        zg = z_g.reshape(c.B, 1, c.dim_g).expand(c.B, c.LEN, c.dim_g)
        lab = labels.reshape(c.B, 1, c.n_c).expand(c.B, c.LEN, c.n_c)
        return torch.cat([zg, z_l.reshape(c.B, c.LEN, c.dim_l), lab], -1).reshape(c.B * c.LEN, c.dim_g + c.dim_l + c.n_c)

    # ---- nets -----------------------------------------------------------------------------------------------------------
    def _operator(self, name, a, b, res_src):
        c = self.cfg
        out = self._lin(name + '.Input', a.shape[1] + b.shape[1], c.dim_op, torch.cat([a, b], 1), LRELU)
        out = self._lin(name + '.1', c.dim_op, c.dim_op, out, LRELU)
        out = self._lin(name + '.Output', c.dim_op, c.dim_l, out)
        if c.op_dyn_mode == 'res':
            return out + res_src
        return out + self._lin(name + '.ZW', c.dim_l, c.dim_l, res_src)

    def ImplicitOperator(self, z_l, epsilon, name):
        return self._operator(name, z_l, epsilon, z_l)

    def ConcatOperator(self, z_l_0, z_l_1_pre, name):
        return self._operator(name, z_l_0, z_l_1_pre, z_l_0)

    def DynamicGenerator(self, z_l_0, epsilon):
        c, zs = self.cfg, [z_l_0]
        name = 'Generator.Dynamic'
        P = lib.ops.linear.linear_params
        if (c.fuse and c.dim_op == 256 and c.dim_l <= 16 and c.dim_t <= 16 and z_l_0.is_cuda
                and not os.environ.get('GGAN_NO_DYN_SCAN')):
            # the LEN-1 applications of the shared-weight operator as ONE scan launch per direction (functional.DynScan)
            # (while the graph is being built every application draws its initial values, as in the reference: same RNG stream)
            for _ in range(c.LEN - 1 if lib.initial_values_needed(name + '.Input.W') else 1):
                w_in, b_in = P(name + '.Input', c.dim_l + c.dim_t, c.dim_op)
                w_1, b_1 = P(name + '.1', c.dim_op, c.dim_op)
                w_out, b_out = P(name + '.Output', c.dim_op, c.dim_l)
                zw, b_zw = P(name + '.ZW', c.dim_l, c.dim_l) if c.op_dyn_mode != 'res' else (None, None)
            return F.DynScan.apply(z_l_0, epsilon, w_in, b_in, w_1, b_1, w_out, b_out, zw, b_zw, c.LEN - 1, 0.2)
        for _ in range(c.LEN - 1):
            zs.append(self.ImplicitOperator(zs[-1], epsilon, 'Generator.Dynamic'))
        return torch.stack(zs, 1)

    def DynamicExtractor(self, z_pre):
        c, L = self.cfg, self.cfg.LEN
        if c.pos_mode == 'naive_mean_field':
            return z_pre
        if c.pos_mode == 'inverse':
            zs = [z_pre[:, L - 1, :]]
            for i in range(L - 1):
                zs.insert(0, self.ConcatOperator(zs[0], z_pre[:, L - i - 2, :], 'Extractor.Dynamic.Backward'))
        elif c.pos_mode == 'forward_inverse':
            zs = [z_pre[:, 0, :]]
            for i in range(L - 1):
                zs.append(self.ConcatOperator(zs[-1], z_pre[:, i + 1, :], 'Extractor.Dynamic.Forward'))
        else:   # gsp
            tmp = [z_pre[:, L - 1, :]]
            for i in range(L - 1):
                tmp.insert(0, self.ConcatOperator(tmp[0], z_pre[:, L - i - 2, :], 'Extractor.Dynamic.Backward'))
            zs = [tmp[0]]
            for i in range(L - 1):
                zs.append(self.ConcatOperator(zs[-1], tmp[i + 1], 'Extractor.Dynamic.Forward'))
        return torch.stack(zs, 1)

    def Generator(self, z_g, z_l, labels, out_slot=None):
        c, d = self.cfg, self.cfg.dim
        out = self._lin('Generator.Input', c.dim_g + c.dim_l + c.n_c, c.flat, self._z_rows(z_g, z_l, labels), RELU)
        out = out.reshape(c.B * c.LEN, 8 * d, 4, 4)
        out = self._deconv('Generator.2', 8 * d, 4 * d, out, RELU)
        out = self._deconv('Generator.3', 4 * d, 2 * d, out, RELU)
        out = self._deconv('Generator.4', 2 * d, d, out, RELU)
        out = self._deconv('Generator.5', d, c.C, out, TANH, out=out_slot)
        return out.reshape(c.B, c.LEN, c.output_dim)

    def _conv_stack(self, pre, x, cin, grad_rows=None):
        d = self.cfg.dim
        out = self._conv(pre + '.1', cin, d, x, grad_rows)
        out = self._conv(pre + '.2', d, 2 * d, out, grad_rows)
        out = self._conv(pre + '.3', 2 * d, 4 * d, out, grad_rows)
        return self._conv(pre + '.4', 4 * d, 8 * d, out, grad_rows)

    def Extractor(self, inputs, labels):
        c = self.cfg
        out = self._conv_stack('Extractor', inputs.reshape(c.B * c.LEN, c.C, 64, 64), c.C).reshape(c.B * c.LEN, c.flat)
        out = torch.cat([out, self.expand_labels(labels)], 1)
        return self._lin('Extractor.Output', c.flat + c.n_c, c.dim_l, out).reshape(c.B, c.LEN, c.dim_l)

    def G_Extractor(self, inputs, labels):
        c = self.cfg
        out = self._conv_stack('Extractor.G', inputs.reshape(c.B, c.C * c.LEN, 64, 64), c.C * c.LEN).reshape(c.B, c.flat)
        return self._lin('Extractor.G.Output', c.flat + c.n_c, c.dim_g, torch.cat([out, labels], 1))

    def Discriminator(self, x, z_g, z_l, labels):
        c = self.cfg
        return self._frame_critic(x.reshape(c.B * c.LEN, c.output_dim), self._z_rows(z_g, z_l, labels), self.expand_labels(labels))

    def _frame_critic(self, frames, z_rows, label_rows, grad_rows=None):
        """the frame critic on any number of (frame, latent row, label row) triples; grad_rows: only the leading frames carry
        a gradient (generator steps: [fake; real])"""
        c, n = self.cfg, frames.shape[0]
        out = self._conv_stack('Discriminator', frames.reshape(n, c.C, 64, 64), c.C, grad_rows).reshape(n, c.flat)
        z_out = self._lin('Discriminator.z1', c.dim_g + c.dim_l + c.n_c, 512, z_rows, LRELU)
        out = torch.cat([out, z_out, label_rows], 1)
        out = self._lin('Discriminator.zx1', c.flat + 512 + c.n_c, 512, out, LRELU)
        return self._lin('Discriminator.Output', 512, 1, out).reshape(-1)

    def SequenceDiscriminator(self, x, z_g, z_l, labels, grad_rows=None):
        """ALI_MODE = 'concat_x' (:407-449): the frames of a sequence as input channels, one logit per sequence; works on any
        number of (sequence, z_g, z_l, labels) rows (the critic step hands it [fake; real])"""
        c, n = self.cfg, x.shape[0]
        if c.ali_mode == '3dcnn':        # :352-405: the sequence as an NLHWC volume (one channel: the transpose is a reshape)
            out, cin, s24 = x.reshape(n, c.LEN, 64, 64, 1), 1, (2 if c.LEN == 16 else 1)
            for i, (cout, sl) in enumerate(((c.dim, 2), (2 * c.dim, s24), (4 * c.dim, 2), (8 * c.dim, s24))):
                out = lib.ops.conv3d.Conv3D('Discriminator.%d' % (i + 1), 4, cin, cout, 4, out, stride=2, stride_len=sl,
                                            activation=LRELU if c.fuse else None, grad_rows=grad_rows)
                out, cin = (out if c.fuse else lib.ops.act.LeakyReLU(out)), cout
            z = torch.cat([z_g, z_l.reshape(n, c.LEN * c.dim_l), labels], 1)
            z_out = self._lin('Discriminator.z1', c.dim_g + c.dim_l * c.LEN + c.n_c, 512, z, LRELU)
            out = self._lin('Discriminator.zx1', c.flat + 512, 512, (out.reshape(n, c.flat), z_out), LRELU)
            return self._lin('Discriminator.Output', 512, 1, out).reshape(-1)
        if c.ali_mode == 'concat_z':     # :451-497: per-frame conv stack + a 4x4 VALID conv to DIM_LATENT_G features per frame
            fr = None if grad_rows is None else grad_rows * c.LEN
            out = self._conv_stack('Discriminator', x.reshape(n * c.LEN, c.C, 64, 64), c.C, fr)
            out = lib.ops.conv2d.Conv2D('Discriminator.5', 8 * c.dim, c.dim_g, 4, out, stride=1, padding='VALID').reshape(n, c.LEN * c.dim_g)
            z = torch.cat([z_g, z_l.reshape(n, c.LEN * c.dim_l), labels], 1)
            z_out = self._lin('Discriminator.z1', c.dim_g + c.dim_l * c.LEN + c.n_c, 512, z, LRELU)
            out = self._lin('Discriminator.zx1', c.LEN * c.dim_g + 512 + c.n_c, 512, torch.cat([out, z_out, labels], 1), LRELU)
            return self._lin('Discriminator.Output', 512, 1, out).reshape(-1)
        out = self._conv_stack('Discriminator', x.reshape(n, c.C * c.LEN, 64, 64), c.C * c.LEN, grad_rows).reshape(n, c.flat)
        z = torch.cat([z_g, z_l.reshape(n, c.LEN * c.dim_l), labels], 1)
        z_out = self._lin('Discriminator.z1', c.dim_g + c.dim_l * c.LEN + c.n_c, 512, z, LRELU)
        out = self._lin('Discriminator.zx1', c.flat + 512, 512, (out, z_out), LRELU)
        return self._lin('Discriminator.Output', 512, 1, out).reshape(-1)

    def _forward_seq(self, feed, which, out):
        """MODE ali / alice-z (:536-538, :553-558)"""
        c = self.cfg
        real_y, p_y = feed['real_y'], feed['p_y']
        real_x, fake_x, q_z_l, q_z_g, p_z_l, p_z_g = (out[k] for k in ('real_x', 'fake_x', 'q_z_l', 'q_z_g', 'p_z_l', 'p_z_g'))
        if which == 'disc':
            fake_x, q_z_l, q_z_g, p_z_l = fake_x.detach(), q_z_l.detach(), q_z_g.detach(), p_z_l.detach()
        J = lib.objs.gan_inference
        J.ONLY[0] = which
        with (lib.frozen('Discriminator') if which == 'gen' else lib.frozen()):
            if which in ('gen', 'disc'):
                assert which == 'disc' or not real_x.requires_grad, 'grad_rows: the real sequences must be data'
                d = self.SequenceDiscriminator(F.JoinRows.apply(fake_x, real_x), torch.cat([p_z_g, q_z_g], 0), torch.cat([p_z_l, q_z_l], 0),
                                               torch.cat([p_y, real_y], 0), grad_rows=c.B if which == 'gen' else None)
                d_fake, d_real = F.SplitRows.apply(d, c.B)
            else:
                d_fake = self.SequenceDiscriminator(fake_x, p_z_g, p_z_l, p_y)
                d_real = self.SequenceDiscriminator(real_x, q_z_g, q_z_l, real_y)
        gen_params, disc_params = self._var_lists()
        if c.mode == 'alice-z':
            rec = c.lamb * lib.utils.distance.distance(real_x, self.Generator(q_z_g, q_z_l, real_y), 'l2') if which != 'disc' else None
            res = J.alice(d_fake, d_real, rec, gen_params, disc_params, lr=c.lr, beta1=c.beta1)
        else:
            res = J.ali(d_fake, d_real, gen_params, disc_params, lr=c.lr, beta1=c.beta1)
        J.ONLY[0] = None
        out.update(disc_fake=d_fake, disc_real=d_real, gen_cost=res[0], disc_cost=res[1], gen_train_op=res[2], disc_train_op=res[3])
        return out

    def _mlp_critic(self, pre, x):
        out = self._lin(pre + '.Input', x.shape[1], 512, x, LRELU)
        out = self._lin(pre + '.2', 512, 512, out, LRELU)
        out = self._lin(pre + '.3', 512, 512, out, LRELU)
        return self._lin(pre + '.Output', 512, 1, out).reshape(-1)

    def DynamicDiscrminator(self, z1, z2):
        return self._mlp_critic('Discriminator.Dynamic', torch.cat([z1, z2], 1))

    def ZGDiscrminator(self, z_g):
        return self._mlp_critic('Discriminator.ZG', z_g)

    def _pairs(self, z_l):
        """all LEN-1 (z_t, z_t+1) pairs stacked time-major: two [(LEN-1)*B, dim_l] tensors"""
        c = self.cfg
        return (z_l[:, :-1, :].transpose(0, 1).reshape((c.LEN - 1) * c.B, c.dim_l),
                z_l[:, 1:, :].transpose(0, 1).reshape((c.LEN - 1) * c.B, c.dim_l))

    def _transitions(self, z_l):
        return self.DynamicDiscrminator(*self._pairs(z_l))

    @staticmethod
    def _var_lists():
        gen = lib.params_with_name('Generator') + lib.params_with_name('Extractor')      # gen_params + ext_params (:543-545)
        return gen, lib.params_with_name('Discriminator')

    # ---- loss wiring ------------------------------------------------------------------------------------------------------
    def _fork(self, device):
        """-> the current stream when a chain of short launches may go to the second stream (a step graph is being captured),
        with that stream waiting for everything issued so far; else None"""
        if not (self.fork_nets and self.fork_now) or device.type != 'cuda':
            return None
        if self._side is None:
            self._side = F.shared_stream(device, 'side')
        cur = torch.cuda.current_stream(device)
        self._side.wait_stream(cur)
        return cur

    def forward_nets(self, feed):
        """everything that reads no critic variable (:515-527)"""
        real_y, p_y = feed['real_y'], feed['p_y']
        # the transition operator's scan is one workgroup per sequence (32 of them) walking LEN-1 steps: it runs on the second
        # stream beside the Extractor's conv stack over all frames instead of in front of the chip
        cur = self._fork(feed['p_z_g'].device)
        nf = self.cfg.B * self.cfg.LEN
        pair = feed.get('x_pair') if self.cfg.fuse else None
        slots = (F.RowSlot(pair, 0, nf), F.RowSlot(pair, nf, 2 * nf)) if pair is not None else (None, None)

        def extractor_side():
            real_x = F.Axpby.apply(feed['real_x_unit'], feed['real_x_unit'], 2.0 / self.cfg.x_div, 0.0, -1.0, slots[1])      # 2*(x/div-.5)
            return real_x, self.DynamicExtractor(self.Extractor(real_x, real_y)), self.G_Extractor(real_x, real_y)
        if cur is not None and os.environ.get('GGAN_SSGAN_NETS') != 'scan_aside':
            # (round 6) the whole Extractor family on the second stream, the scan and the frame generator on this one: the two halves of the
            # nets pass share nothing, and -- autograd keeps a backward node on the stream of its forward -- neither do their backward
            # passes: the scan's LEN-1 sequential steps per direction (32 workgroups) and the short products of the latent paths hide behind
            # the other half's conv launches in BOTH directions.  (Before: only the forward scan ran aside; in a generator step the
            # Extractor's backward -- its tail of short launches first -- queued behind the frame generator's on one stream.)
            with torch.cuda.stream(self._side):
                real_x, q_z_l, q_z_g = extractor_side()
            p_z_l = self.DynamicGenerator(feed['p_z_l_0'], feed['epsilon'])
            fake_x = self.Generator(feed['p_z_g'], p_z_l, p_y, out_slot=slots[0])
            cur.wait_stream(self._side)
        else:
            if cur is not None:
                with torch.cuda.stream(self._side):
                    p_z_l = self.DynamicGenerator(feed['p_z_l_0'], feed['epsilon'])
            real_x, q_z_l, q_z_g = extractor_side()
            if cur is not None:
                cur.wait_stream(self._side)
            else:
                p_z_l = self.DynamicGenerator(feed['p_z_l_0'], feed['epsilon'])
            fake_x = self.Generator(feed['p_z_g'], p_z_l, p_y, out_slot=slots[0])
        return dict(real_x=real_x, q_z_l=q_z_l, q_z_g=q_z_g, p_z_l=p_z_l, p_z_g=feed['p_z_g'], fake_x=fake_x)

    def forward(self, feed, which=None, nets=None):
        c = self.cfg
        out = dict(nets) if nets is not None else self.forward_nets(feed)
        if c.seq_critic:
            return self._forward_seq(feed, which, out)
        real_y, p_y = feed['real_y'], feed['p_y']
        real_x, fake_x, q_z_l, q_z_g, p_z_l, p_z_g = (out[k] for k in ('real_x', 'fake_x', 'q_z_l', 'q_z_g', 'p_z_l', 'p_z_g'))
        if which == 'disc':      # the critic step needs no gradient w.r.t. the generator/extractor outputs
            fake_x, q_z_l, q_z_g, p_z_l = fake_x.detach(), q_z_l.detach(), q_z_g.detach(), p_z_l.detach()
        J = lib.objs.gan_inference
        J.ONLY[0] = which
        with (lib.frozen('Discriminator') if which == 'gen' else lib.frozen()):
            if which in ('gen', 'disc'):
                # every critic once, on [fake; real] (rows are independent: no BatchNorm); in generator steps the conv
                # data-gradient is needed for the fake frames only
                nf = c.B * c.LEN
                assert which == 'disc' or not real_x.requires_grad, 'grad_rows: the real frames must be data'
                # the two latent critics (MLP chains of short launches) run on the second stream beside the frame critic's conv stack
                cur = self._fork(real_x.device)

                def latent_critics():
                    (af, bf), (ar, br) = self._pairs(p_z_l), self._pairs(q_z_l)
                    return (self.DynamicDiscrminator(torch.cat([af, ar], 0), torch.cat([bf, br], 0)),
                            self.ZGDiscrminator(torch.cat([p_z_g, q_z_g], 0)))
                if cur is not None:
                    with torch.cuda.stream(self._side):
                        t, zg = latent_critics()
                else:
                    t, zg = latent_critics()
                d = self._frame_critic(F.JoinRows.apply(fake_x.reshape(nf, -1), real_x.reshape(nf, -1)),
                                       torch.cat([self._z_rows(p_z_g, p_z_l, p_y), self._z_rows(q_z_g, q_z_l, real_y)], 0),
                                       torch.cat([self.expand_labels(p_y), self.expand_labels(real_y)], 0),
                                       grad_rows=nf if which == 'gen' else None)
                if cur is not None:
                    cur.wait_stream(self._side)
                (tf_, tr_), (zf, zr), (df, dr) = (F.SplitRows.apply(t, (c.LEN - 1) * c.B), F.SplitRows.apply(zg, c.B),
                                                  F.SplitRows.apply(d, nf))
                disc_fake, disc_real = [tf_, zf, df], [tr_, zr, dr]
            else:
                disc_fake = [self._transitions(p_z_l), self.ZGDiscrminator(p_z_g), self.Discriminator(fake_x, p_z_g, p_z_l, p_y)]
                disc_real = [self._transitions(q_z_l), self.ZGDiscrminator(q_z_g), self.Discriminator(real_x, q_z_g, q_z_l, real_y)]
        r = c.ratio()
        ratios = [float(r[0]) * (c.LEN - 1), float(r[c.LEN - 1]), float(r[c.LEN])]      # the LEN-1 equal transition terms as one
        gen_params, disc_params = self._var_lists()
        rec_penalty = None
        if c.mode == 'local_epce-z' and which != 'disc':
            rec_x = self.Generator(q_z_g, q_z_l, real_y)
            rec_penalty = c.lamb * lib.utils.distance.distance(real_x, rec_x, 'l2')
        res = J.weighted_local_epce(disc_fake, disc_real, ratios, gen_params, disc_params, lr=c.lr, beta1=c.beta1,
                                    rec_penalty=rec_penalty)
        J.ONLY[0] = None
        out.update(disc_fake=disc_fake, disc_real=disc_real, gen_cost=res[0], disc_cost=res[1], gen_train_op=res[4],
                   disc_train_op=res[5])
        return out

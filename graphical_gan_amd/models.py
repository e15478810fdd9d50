"""Net definitions + loss wiring of the reference's driver scripts, written against the tflib mirror.

One parametrised definition covers the six image scripts (the reference repeats it per file):
  gan_inference_cifar10.py:133-255,261-366      Generator / Extractor / Discriminator, MODE ali | alice* | wali | wali-gp | vegan | vegan-wgan-gp
  gmgan_inference_cifar10.py:114-301,341-398    + HyperGenerator / HyperExtractor / HyperDiscriminator, MODE local_ep
  gmgan_inference_mnist.py:166-300              28x28x1, crop [:,:,:7,:7], sigmoid output, float input
  gan_inference_mnist.py:122-250                the same nets, but a critic of its own (BatchNorm, two more Linear layers: Config.critic_deep)
  gmgan_inference_face.py:82-274                64x64x3, four conv stages, DIM 32, no BatchNorm, dequantisation
Layer names are the reference's, so the registry keys match.
`fuse=True` folds bias+activation into the producing conv/linear/BN kernel (identical math, fewer HBM passes).
"""
import contextlib
import os
import weakref

import numpy as np
import torch

from . import functional as F
from . import tflib as lib
from .tflib.ops.act import LRELU, RELU, TANH, SIGMOID


class Config(object):
    def __init__(self, dataset='cifar10', batch_size=64, n_coms=0, mode=None, dim=None, dim_latent=128, bn=None,
                 temp=0.1, fuse=True, lr=None, batch_critic=True):
        self.dataset, self.B, self.K, self.dim_latent, self.temp, self.fuse = dataset, batch_size, n_coms, dim_latent, temp, fuse
        if dataset == 'cifar10':
            self.C, self.S, self.dim, self.nl, self.bn, self.out_act = 3, 32, 64, 3, True, 'tanh'
        elif dataset == 'svhn':          # g(m)gan_inference_svhn.py: the CIFAR nets with BN_FLAG = False (:69-72)
            self.C, self.S, self.dim, self.nl, self.bn, self.out_act = 3, 32, 64, 3, False, 'tanh'
        elif dataset == 'mnist':
            self.C, self.S, self.dim, self.nl, self.bn, self.out_act = 1, 28, 64, 3, True, 'sigmoid'
        elif dataset == 'face':
            self.C, self.S, self.dim, self.nl, self.bn, self.out_act = 3, 64, 32, 4, False, 'tanh'
        else:
            raise ValueError(dataset)
        if dim is not None:
            self.dim = dim
        if bn is not None:
            self.bn = bn
        self.mode = mode or ('local_ep' if n_coms else 'ali')
        # reconstruction variants (gan_inference_cifar10.py:293-304, gmgan_inference_cifar10.py:399-403):
        #   alice-z: + l2(real_x, G(q_z));  alice-x: + l2(p_z, E(fake_x));  alice: both;  local_epce: gmgan + l2(real_x, G(q_z))
        #   vegan / vegan-wgan-gp (gan_inference_cifar10.py:192-222,305-322): the critic is an MLP on codes, + l2(real_x, G(q_z))
        #   vegan-mmd (:327-329, tflib/objs/mmd.py): no critic at all -- lamb * MMD^2(q_z, p_z) + l2(real_x, G(q_z)), generator steps only
        #   vegan-kl / vegan-ikl / vegan-jsd (:331-341, tflib/objs/kl_aggregated.py): no critic, a stochastic encoder (TYPE_Q = 'learn_std',
        #     :40-43,173-188) and lamb * D(aggregated posterior || N(0, I)) on Z_SAMPLES Monte-Carlo samples + l2(real_x, G(q_z))
        assert self.mode in ('ali', 'local_ep', 'wali', 'wali-gp', 'alice', 'alice-z', 'alice-x', 'local_epce', 'vegan',
                             'vegan-wgan-gp', 'vegan-mmd', 'vegan-kl', 'vegan-ikl', 'vegan-jsd')
        self.agg = self.mode in ('vegan-kl', 'vegan-ikl', 'vegan-jsd')
        self.learn_std, self.z_samples = self.agg, 100               # TYPE_Q, Z_SAMPLES (:41-43)
        self.latent_critic = self.mode in ('vegan', 'vegan-wgan-gp')
        self.lamb = 1.0                                                  # LAMBDA (gan_inference_cifar10.py:62)
        self.no_critic = self.mode == 'vegan-mmd' or self.agg
        assert not ((self.latent_critic or self.no_critic) and n_coms)
        self.top = self.dim * 2 ** (self.nl - 1)
        self.flat = 16 * self.top
        self.output_dim = self.C * self.S * self.S
        self.critic_iters = 5 if self.mode in ('wali', 'wali-gp', 'vegan', 'vegan-wgan-gp') else 1   # gan_inference_cifar10.py:53-59
        if self.no_critic:
            self.critic_iters = 0          # 'No discriminators' (:52-53)
        self.lr = lr if lr is not None else {'wali-gp': 1e-4, 'wali': 5e-5}.get(self.mode, 2e-4)
        self.beta1 = 0.5
        # critic steps evaluate the critic ONCE on [fake; real] (the critics of these scripts have no BatchNorm, so
        # rows are independent and the result is identical); generator steps keep the two branches separate because
        # only the fake branch needs the conv data-gradients
        self.batch_critic = batch_critic
        # gan_inference_mnist.py:215-250: the joint critic of THAT script has BatchNorm after conv 2 / 3 (BN_FLAG) and a second Linear
        # on the z path ('Discriminator.2': the prefix is shared with the conv layer, the keys are distinct) and on the joint path
        # ('Discriminator.zx2').  BatchNorm makes the critic's rows dependent: the fake and the real pair are two evaluations.
        self.critic_deep = dataset == 'mnist' and not n_coms and not self.latent_critic and not self.no_critic
        if self.critic_deep and self.bn:
            self.batch_critic = False


class GraphicalGAN(object):
    """Builds the per-step graph of one script: forward(feed) -> dict with gen_cost / disc_cost / train ops."""

    def __init__(self, cfg):
        self.cfg = cfg
        self._side = None                                     # second stream of forward_nets
        self._early = False                                   # begin_nets() forked it before the noise launch
        self._pending_join = None                             # [stream, event after real_x, event at the branch's end]
        self._gp_stream = None                                # the stream the gradient-penalty pass of a wali-gp critic step was issued on
        # two-stream Extractor / Generator passes: measured +3 % (ali, face, mnist), -5..7 % with the gradient penalty (more
        # cross-stream edges than overlap), so the joint-critic modes without a penalty ask for it; the Trainer
        # turns it on while it builds a single-graph step (fork_now) -- eager steps are host-bound and gain nothing
        # (round 2, after the kernels got shorter: +1.5 % with the mixture prior too; wali-gp: -6.6 % with the fork behind the noise
        #  launch and an immediate join, +1.3 % once the Extractor branch is a root of the graph and carries the critic's z path)
        self.fork_nets = (not os.environ.get('GGAN_NO_FORK_NETS')
                          and cfg.mode in ('ali', 'alice', 'alice-z', 'alice-x', 'wali', 'wali-gp', 'local_ep', 'local_epce')) or bool(os.environ.get('GGAN_FORCE_FORK_NETS'))
        self.fork_now = False

    # ---- engine hooks: the static inputs of one session.run (what the reference feeds / samples) -----------------
    @property
    def single_contribution(self):
        """every parameter receives exactly one gradient contribution per backward pass (critic evaluated once on
        [fake; real]; the wali-gp penalty re-enters the critic)"""
        # (reconstruction terms reuse G / E; the wali-gp penalty pass uses second leaves of the critic's weights, see forward)
        return bool(self.cfg.batch_critic) and (self.cfg.mode in ('ali', 'local_ep', 'wali') or
                                                (self.cfg.mode == 'wali-gp' and not os.environ.get('GGAN_NO_SECOND_LEAF')))

    def launch_hint(self, which):
        """workgroups per conv launch the step should plan for (engine.Trainer._step_body -> functional.launch_hint -> ggan_conv_geom.plan_wgs), 0 = default:
        128 in wali-gp critic steps while a step graph is built -- there the penalty pass runs beside the [fake; real] pass, and launches
        of ~128 workgroups let the two chains run on different CUs (measured -1 % of the iteration even with the generator step, which
        wants the default, planned the same way; GGAN_NO_LAUNCH_HINT)"""
        c = self.cfg
        if (c.mode == 'wali-gp' and which == 'disc' and c.batch_critic and self.fork_nets and self.fork_now
                and not os.environ.get('GGAN_NO_FORK_GP') and not os.environ.get('GGAN_NO_LAUNCH_HINT')):
            return int(os.environ.get('GGAN_GP_TARGET_WGS', '128'))
        return 0

    def cut_tensors(self, nets):
        """tensors through which EVERY gradient of the Extractor's parameters flows (None if there is no such cut: the
        reconstruction modes apply the Extractor a second time) -- lets a data-parallel generator step exchange the
        Generator's gradients while the Extractor's backward pass is still running (engine.Trainer)."""
        # (q_z_src: the Extractor's output in front of the aliases the mixture scripts hand its readers, functional.fanout)
        return [nets.get('q_z_src', nets['q_z'])] if self.cfg.mode in ('ali', 'local_ep', 'wali', 'wali-gp') else None   # (vegan: G(q_z) re-enters)

    def critic_cut(self):
        """(tensor, number of conv layers) through which every gradient of the image critic's conv stack flows in the critic step
        just built, or None (the wali-gp penalty re-enters the stack): lets a data-parallel critic step put the head's large
        weight gradient (Discriminator.zx1: 10 of the 16 MB) on the wire while the conv stack's backward pass still runs."""
        ref = getattr(self, '_critic_features', None)
        t = ref() if ref is not None else None
        if t is None or not t.requires_grad or self.cfg.mode not in ('ali', 'local_ep', 'wali') or not self.cfg.batch_critic:
            return None
        return t, self.cfg.nl

    def feed_buffers(self, device):
        c, B, feed = self.cfg, self.cfg.B, {}
        if c.dataset == 'mnist':
            # float images are fed as they are: the feed buffer IS the second half of the critic's [fake_x ; real_x] pair
            feed['x_pair'] = torch.zeros(2 * B, c.output_dim, device=device)
            feed['real_x'] = feed['x_pair'][B:]
        else:
            feed['real_x_int'] = torch.zeros(B, c.output_dim, dtype=torch.int32, device=device)
        if c.dataset == 'face':
            feed['dequant_u'] = torch.zeros(B, c.output_dim, device=device)
        # latent pair [p_z ; q_z] of the batched critic: the noise IS the first half (K == 0), the Extractor writes the second
        feed['z_pair'] = torch.zeros(2 * B, c.dim_latent, device=device)
        feed['p_z_noise'] = feed['z_pair'][:B] if not c.K else torch.zeros(B, c.dim_latent, device=device)
        if c.K:
            # assignment pair [one-hot prior draw ; q_k] of the batched critic (same idea as z_pair)
            feed['k_pair'] = torch.zeros(2 * B, c.K, device=device)
            feed['k_onehot'] = feed['k_pair'][:B]
            feed['gumbel_u'] = torch.zeros(B, c.K, device=device)
        if c.mode in ('wali-gp', 'vegan-wgan-gp'):
            feed['alpha'] = torch.zeros(B, 1, device=device)
        if c.latent_critic:          # the Gaussian noise layers of the latent critic: one set of draws per critic call
            for tag in ('f', 'r') + (('h',) if c.mode == 'vegan-wgan-gp' else ()):
                for i, w in enumerate((c.dim_latent, 1024, 512, 256)):
                    feed['dn_%s%d' % (tag, i)] = torch.zeros(B, w, device=device)
        if c.agg:                    # the encoder's eps and the Monte-Carlo draws of tflib/objs/kl_aggregated.py (component one-hots, eps, prior samples)
            feed['q_eps'] = torch.zeros(B, c.dim_latent, device=device)
            feed['kl_k'] = torch.zeros(c.z_samples, B, device=device)
            feed['kl_eps'] = torch.zeros(c.z_samples, c.dim_latent, device=device)
            feed['kl_zp'] = torch.zeros(c.z_samples, c.dim_latent, device=device)
        return feed

    def sample_noise(self, f):
        """Fresh noise for one session.run, drawn on device in ONE launch (graph-capturable: the generator's draw number lives in
        device memory, functional.noise_fill_): p_z ~ N(0,1), k ~ Cat(1/K) as one-hot rows, Gumbel U, GP alpha, dequantisation
        noise, the latent critic's noise layers (gmgan_inference_cifar10.py:115-120,344-346)."""
        c = self.cfg
        if 'rng_state' not in f:
            f['rng_state'] = F.noise_state(f['p_z_noise'].device)
        specs = [(f['p_z_noise'], F.NOISE_NORMAL, 0., 1.)]
        if c.K:
            specs += [(f['k_onehot'], F.NOISE_ONEHOT, 0., 0.), (f['gumbel_u'], F.NOISE_UNIFORM, 0., 1.)]
        if 'alpha' in f:
            specs.append((f['alpha'], F.NOISE_UNIFORM, 0., 1.))
        specs += [(f[k], F.NOISE_NORMAL, 0., 1.) for k in sorted(f) if k.startswith('dn_')]
        if c.agg:
            specs += [(f['q_eps'], F.NOISE_NORMAL, 0., 1.), (f['kl_k'], F.NOISE_ONEHOT, 0., 0.), (f['kl_eps'], F.NOISE_NORMAL, 0., 1.),
                      (f['kl_zp'], F.NOISE_NORMAL, 0., 1.)]
        if c.dataset == 'face':
            specs.append((f['dequant_u'], F.NOISE_UNIFORM, 0., 1. / 128))
        F.noise_fill_(f['rng_state'], specs)

    def set_batch(self, feed, batch):
        feed['real_x' if self.cfg.dataset == 'mnist' else 'real_x_int'].copy_(batch, non_blocking=True)

    def synthetic_ring(self, device, n=8, seed=1234):
        """Device-resident ring of pre-staged synthetic minibatches: uniform integers 0..255 (int32, as the
        reference's placeholder) or U[0,1) floats for MNIST."""
        c, rng, ring = self.cfg, np.random.default_rng(seed), []
        for _ in range(n):
            if c.dataset == 'mnist':
                b = torch.as_tensor(rng.random((c.B, c.output_dim), dtype=np.float32))
            else:
                b = torch.as_tensor(rng.integers(0, 256, size=(c.B, c.output_dim)).astype(np.int32))
            ring.append(b.to(device))
        return ring


    # ---- small helpers: an op followed by its pointwise, fused or not -------------------------------------
    def _conv(self, name, cin, cout, x, act, grad_rows=None):
        if self.cfg.fuse:
            return lib.ops.conv2d.Conv2D(name, cin, cout, 5, x, stride=2, activation=act, grad_rows=grad_rows)
        return F.ActFwd.apply(lib.ops.conv2d.Conv2D(name, cin, cout, 5, x, stride=2, grad_rows=grad_rows), act, 0.2)

    def _lin(self, name, nin, nout, x, act):
        if self.cfg.fuse:
            return lib.ops.linear.Linear(name, nin, nout, x, activation=act)
        return F.ActFwd.apply(lib.ops.linear.Linear(name, nin, nout, x), act, 0.2)

    def _bn(self, name, axes, x, act):
        if self.cfg.fuse:
            return lib.ops.batchnorm.Batchnorm(name, axes, x, activation=act)
        return F.ActFwd.apply(lib.ops.batchnorm.Batchnorm(name, axes, x), act, 0.2)

    # ---- nets ---------------------------------------------------------------------------------------
    def Generator(self, noise, out_slot=None):
        """out_slot: optional functional.RowSlot for the generated images (only honoured with fused epilogues)"""
        c = self.cfg
        if c.bn and c.fuse:
            out = lib.ops.linear.LinearBatchnormRows('Generator.Input', c.dim_latent, c.flat, noise, 'Generator.BN1', activation=RELU)
        elif c.bn:
            out = lib.ops.linear.Linear('Generator.Input', c.dim_latent, c.flat, noise)
            out = self._bn('Generator.BN1', [0], out, RELU)
        else:
            out = self._lin('Generator.Input', c.dim_latent, c.flat, noise, RELU)
        slot = out_slot
        out = out.reshape(-1, c.top, 4, 4)
        names = ['2', '3', '4', '5'] if c.nl == 4 else ['2', '3', '5']
        ch = c.top
        for i, nm in enumerate(names):
            last = i == len(names) - 1
            cout = c.C if last else ch // 2
            final_act = TANH if c.out_act == 'tanh' else SIGMOID
            if last:
                if c.fuse:
                    out = lib.ops.deconv2d.Deconv2D('Generator.' + nm, ch, cout, 5, out, activation=final_act, out=slot)
                else:
                    out = F.ActFwd.apply(lib.ops.deconv2d.Deconv2D('Generator.' + nm, ch, cout, 5, out), final_act, 0.0)
            elif c.bn:
                out = lib.ops.deconv2d.Deconv2D('Generator.' + nm, ch, cout, 5, out)
                out = self._bn('Generator.BN' + nm, [0, 2, 3], out, RELU)
            elif c.fuse:
                out = lib.ops.deconv2d.Deconv2D('Generator.' + nm, ch, cout, 5, out, activation=RELU)
            else:
                out = F.relu(lib.ops.deconv2d.Deconv2D('Generator.' + nm, ch, cout, 5, out))
            if c.dataset == 'mnist' and nm == '2':
                out = out[:, :, :7, :7].contiguous()                     # gmgan_inference_mnist.py:179
            ch = cout
        return out.reshape(-1, c.output_dim)

    def Extractor(self, inputs, out_slot=None, eps=None, after_first=None):
        """eps given (TYPE_Q = 'learn_std', gan_inference_cifar10.py:173-188): returns (mean + eps * std, mean, std), std = exp(Linear
        'Extractor.Std').  inputs may be a functional.PendingCast (the minibatch still int32 in the device ring: the first layer scales
        it while staging); after_first: called once the first layer is issued (real_x exists from there on)."""
        c = self.cfg
        out = inputs.reshape(-1, c.C, c.S, c.S)
        ch = c.C
        for i in range(c.nl):
            cout = c.dim * 2 ** i
            name = 'Extractor.%d' % (i + 1)
            if c.bn and i > 0:
                out = lib.ops.conv2d.Conv2D(name, ch, cout, 5, out, stride=2)
                out = self._bn('Extractor.BN%d' % (i + 1), [0, 2, 3], out, LRELU)
            else:
                out = self._conv(name, ch, cout, out, LRELU)
            if i == 0 and after_first is not None:
                after_first()
            ch = cout
        out = out.reshape(-1, c.flat)
        if eps is not None:
            log_std = lib.ops.linear.Linear('Extractor.Std', c.flat, c.dim_latent, out)
            mean = lib.ops.linear.Linear('Extractor.Output', c.flat, c.dim_latent, out)
            z, std = F.Reparam.apply(mean, log_std, eps)
            return z, mean, std
        return lib.ops.linear.Linear('Extractor.Output', c.flat, c.dim_latent, out, out=out_slot)

    def Discriminator(self, x, z, grad_rows=None, twice=False, before_z=None, z_out=None):
        """grad_rows: only the first grad_rows images of x carry a gradient (generator steps: [fake; real]); twice: the result
        will be differentiated twice (gradient-penalty pass): plain layer composition instead of the fused critic tail"""
        c = self.cfg
        out = x.reshape(-1, c.C, c.S, c.S)
        ch = c.C
        for i in range(c.nl):
            cout = c.dim * 2 ** i
            name = 'Discriminator.%d' % (i + 1)
            if c.critic_deep and c.bn and i > 0:              # gan_inference_mnist.py:223-230
                out = lib.ops.conv2d.Conv2D(name, ch, cout, 5, out, stride=2, grad_rows=grad_rows)
                out = self._bn('Discriminator.BN%d' % (i + 1), [0, 2, 3], out, LRELU)
            else:
                out = self._conv(name, ch, cout, out, LRELU, grad_rows)   # dropout == identity
            ch = cout
        out = out.reshape(-1, c.flat)
        # (Trainer.critic_cut: every gradient of the conv stack's parameters flows through here.  A WEAK reference: the tape keeps the
        #  tensor alive until its backward pass; a strong one would carry the previous step's tape into the next graph capture, which
        #  ROCm answers with a crash in hipStreamEndCapture -- INTEGRATION.md 3d)
        self._critic_features = weakref.ref(out)
        if before_z is not None:
            before_z()               # (forward_nets: z's second half may still be in flight on the other stream)
        if z_out is None:            # (else: _critic already ran the z path on the second stream)
            z_out = self._lin('Discriminator.z1', c.dim_latent, 512, z, LRELU)
        if c.critic_deep:            # gan_inference_mnist.py:236-248: z1 -> '2' | concat -> zx1 -> zx2 -> Output
            z_out = self._lin('Discriminator.2', 512, 512, z_out, LRELU)
            out = self._lin('Discriminator.zx1', c.flat + 512, 512, (out, z_out), LRELU)
            if c.fuse and not os.environ.get('GGAN_NO_HEAD_FUSION'):
                return lib.ops.linear.LinearLReLULinear('Discriminator.zx2', 512, 512, 'Discriminator.Output', out, differentiable=twice)
            out = self._lin('Discriminator.zx2', 512, 512, out, LRELU)
            return lib.ops.linear.Linear('Discriminator.Output', 512, 1, out).reshape(-1)
        if c.fuse and not os.environ.get('GGAN_NO_HEAD_FUSION'):
            # Linear on concat([out, z_out], 1) + LeakyReLU + the 512 -> 1 Output layer as one op
            return lib.ops.linear.LinearLReLULinear('Discriminator.zx1', c.flat + 512, 512, 'Discriminator.Output', (out, z_out),
                                                    differentiable=twice)
        out = self._lin('Discriminator.zx1', c.flat + 512, 512, (out, z_out), LRELU)     # Linear on concat([out, z_out], 1)
        out = lib.ops.linear.Linear('Discriminator.Output', 512, 1, out)
        return out.reshape(-1)

    def LatentDiscriminator(self, z, noise):
        """gan_inference_cifar10.py:192-222: the vegan critic on codes; Gaussian noise (std .3 on the input, .5 after the first
        three hidden layers) comes in as N(0,1) draws and is scaled here."""
        c = self.cfg
        out = F.Axpby.apply(z, noise[0], 1.0, 0.3, 0.0)
        for i, (nm, nin, nout) in enumerate([('Input', c.dim_latent, 1024), ('2', 1024, 512), ('3', 512, 256), ('4', 256, 256)]):
            if c.bn:
                out = lib.ops.linear.Linear('Discriminator.' + nm, nin, nout, out)
                out = self._bn('Discriminator.BN%d' % (i + 1), [0], out, LRELU)
            else:
                out = self._lin('Discriminator.' + nm, nin, nout, out, LRELU)
            if i < 3:
                out = F.Axpby.apply(out, noise[i + 1], 1.0, 0.5, 0.0)
        return lib.ops.linear.Linear('Discriminator.Output', 256, 1, out).reshape(-1)

    def _forward_latent(self, feed, which, out):
        """MODE vegan / vegan-wgan-gp (gan_inference_cifar10.py:272-275,305-322)"""
        c = self.cfg
        real_x, q_z, p_z = out['real_x'], out['q_z'], out['p_z']
        J = lib.objs.gan_inference
        J.ONLY[0] = which
        noise = lambda tag: [feed['dn_%s%d' % (tag, i)] for i in range(4)]
        det = (lambda t: t.detach()) if which == 'disc' else (lambda t: t)
        with (lib.frozen('Discriminator') if which == 'gen' else lib.frozen()):
            d_real = self.LatentDiscriminator(p_z, noise('r'))
            d_fake = self.LatentDiscriminator(det(q_z), noise('f'))
            gp = None
            if c.mode == 'vegan-wgan-gp' and which != 'gen':
                z_hat = F.RowLerp.apply(p_z, det(q_z), feed['alpha'])      # p_z + alpha*(q_z - p_z)
                if not z_hat.requires_grad:
                    z_hat.requires_grad_(True)
                d_hat = self.LatentDiscriminator(z_hat, noise('h'))
                with F.data_grad_only():
                    (g,) = torch.autograd.grad(d_hat, [z_hat], grad_outputs=F.cached_const(1.0, d_hat.shape, d_hat.device), create_graph=True)
                gp = F.GradPenalty.apply(g, 10.0)
        rec = 1. * lib.utils.distance.distance(real_x, self.Generator(q_z), 'l2') if which != 'disc' else None
        gen_params, disc_params = self._var_lists()
        if c.mode == 'vegan':
            res = J.vegan(d_fake, d_real, rec, gen_params, disc_params, c.lamb, lr=c.lr, beta1=c.beta1)
        else:
            res = J.vegan_wgan_gp(d_fake, d_real, rec, gp, gen_params, disc_params, c.lamb, lr=c.lr, beta1=c.beta1)
        J.ONLY[0] = None
        out.update(disc_fake=d_fake, disc_real=d_real, rec_penalty=rec, gradient_penalty=gp, gen_cost=res[0], disc_cost=res[1],
                   gen_train_op=res[2], disc_train_op=res[3])
        return out

    def HyperDiscriminator(self, z, k):
        c = self.cfg
        out = self._lin('Discriminator.HyperInput', c.dim_latent + c.K, 512, (z, k), LRELU)     # Linear on concat([z, k], 1)
        out = self._lin('Discriminator.Hyper2', 512, 512, out, LRELU)
        if c.fuse and not os.environ.get('GGAN_NO_HEAD_FUSION'):
            return lib.ops.linear.LinearLReLULinear('Discriminator.Hyper3', 512, 512, 'Discriminator.HyperOutput', out)
        out = self._lin('Discriminator.Hyper3', 512, 512, out, LRELU)
        out = lib.ops.linear.Linear('Discriminator.HyperOutput', 512, 1, out)
        return out.reshape(-1)

    def _mu(self, use=None):
        """the component means; use = 0 (HyperGenerator) / 1 (HyperExtractor) inside a nets pass: that net's alias of them, so that the two
        gradient contributions meet in one functional.Fanout node"""
        c = self.cfg
        fan = getattr(self, '_mu_fan', None)
        if fan is not None and use is not None:
            return fan[use]
        return lib.param('Generator.Hyper.Mu', np.random.normal(size=(c.K, c.dim_latent)).astype('float32'))

    def HyperGenerator(self, hyper_k, hyper_noise, out_slot=None):
        """gmgan_inference_cifar10.py:150-153: onehot(k) @ Mu + eps."""
        mu = self._mu(0)
        if self.cfg.fuse and F.MixMean.usable(hyper_k, mu, hyper_noise) and not os.environ.get('GGAN_NO_MIX_MEAN'):
            return F.MixMean.apply(hyper_k, mu, hyper_noise, out_slot)           # one pointwise launch (ggan_mix_mean)
        return F.Axpby.apply(F.Gemm.apply(hyper_k, mu, None, False, False, F.ACT_NONE, 0.0), hyper_noise, 1.0, 1.0, 0.0, out_slot)

    def HyperExtractor(self, latent_z, gumbel_u, out_slot=None):
        """gmgan_inference_cifar10.py:156-173 (MODE_K='CONCRETE'): component logits and the Gumbel-softmax assignment, one
        launch per direction (ggan_gmm_latent_*; the TF graph spends a dozen [B,K] / [B,K,D] pointwise ops on it)."""
        c = self.cfg
        return F.GmmLatent.apply(latent_z, self._mu(1), gumbel_u, float(np.log(np.float32(1.0) / np.float32(c.K))), c.temp, out_slot)

    @staticmethod
    def _var_lists():
        """gmgan_inference_cifar10.py:381-383 -- substring match over the registry, after every net exists."""
        return (lib.params_with_name('Generator') + lib.params_with_name('Extractor'),
                lib.params_with_name('Discriminator'))

    # ---- loss wiring ------------------------------------------------------------------------------------
    def real_x(self, feed, out_slot=None, defer=False):
        """defer: with a device ring, a functional.PendingCast -- the Extractor's first layer scales the minibatch while it stages it"""
        c = self.cfg
        if c.dataset == 'mnist':
            return feed['real_x']
        ring = feed.get('ring')            # Trainer.use_ring: minibatches pre-staged in HBM, walked by the optimizers' step counts
        defer = defer and c.fuse and not os.environ.get('GGAN_NO_CAST_FUSION')
        if c.dataset == 'face':
            return lib.ops.act.cast_scale(feed['real_x_int'], 256., 2., noise=feed['dequant_u'], out=out_slot, ring=ring, defer=defer)
        return lib.ops.act.cast_scale(feed['real_x_int'], 255., 2., out=out_slot, ring=ring, defer=defer)

    def begin_nets(self, feed):
        """Called by the Trainer BEFORE the noise launch of a step.  When the two nets passes are going to run as parallel
        branches and the Extractor pass needs no noise, the second stream is forked here, with nothing in front of it."""
        c = self.cfg
        self._early = False
        if not (self.fork_nets and self.fork_now) or (c.K and os.environ.get('GGAN_NO_EARLY_FORK_K')) or c.agg or c.dataset == 'mnist' or os.environ.get('GGAN_NO_EARLY_FORK'):
            return
        dev = feed['p_z_noise'].device
        if dev.type != 'cuda' or not (c.batch_critic and 'z_pair' in feed):
            return
        if self._side is None:
            self._side = F.shared_stream(dev, 'side')
        self._side.wait_stream(torch.cuda.current_stream(dev))
        self._early = True

    def join_side(self, x_only=False):
        """make the current stream wait for the second stream's real_x (x_only) or for everything it produced"""
        pj = self._pending_join
        if pj is None:
            return
        cur, ev_x, ev_end = pj
        if x_only:
            if ev_x is not None:
                cur.wait_event(ev_x)
                pj[1] = None
            return
        cur.wait_event(ev_end)
        self._pending_join = None

    def forward_nets(self, feed):
        """Extractor and Generator passes: everything of a session.run that does not read a critic variable."""
        c, B = self.cfg, self.cfg.B
        # the batched critic reads [fake_x ; real_x] and [p_z ; q_z]: the producers write straight into the two halves of one
        # buffer each, so those concatenations are aliases instead of copy kernels (functional.RowSlot / JoinRows)
        xs = zs = [None, None]
        if c.batch_critic and 'z_pair' in feed:
            xp = feed['x_pair'] if 'x_pair' in feed else torch.empty((2 * B, c.output_dim), dtype=torch.float32,
                                                                     device=feed['z_pair'].device)
            xs = [F.RowSlot(xp, 0, B), F.RowSlot(xp, B, 2 * B)]
            zs = [F.RowSlot(feed['z_pair'], 0, B), F.RowSlot(feed['z_pair'], B, 2 * B)]
        # the Generator pass does not depend on the Extractor pass: it is issued on a second HIP stream (forked here, joined
        # before anything reads fake_x), inside a step graph as two parallel branches.  The passes are chains of short,
        # latency-bound launches that leave most of the chip idle between them; side by side they fill each other's gaps
        # (+2.8 % on the headline step).  autograd runs each pass's backward on the stream of its forward, so the two backward
        # chains of a generator step overlap the same way.  (Requested by the Trainer for single-graph steps only.)
        # (mixture scripts: tensors two branches of the step read hand each branch an alias -- functional.fanout -- so that their gradient
        #  contributions are summed by this library's launch in one place)
        self._mu_fan = F.fanout(self._mu(), 2) if (c.K and c.fuse) else None
        try:
            return self._forward_nets_k(feed, c, B, xs, zs)
        finally:
            self._mu_fan = None

    def _forward_nets_k(self, feed, c, B, xs, zs):
        p_z = self.HyperGenerator(feed['k_onehot'], feed['p_z_noise'], zs[0]) if c.K else feed['p_z_noise']
        fork = self.fork_nets and self.fork_now and p_z.is_cuda
        nets_target = int(os.environ.get('GGAN_NETS_TARGET_WGS', '128'))
        if fork and nets_target > 0:
            # The two passes run side by side on two streams, and in a generator step so do their backward passes.  A conv launch planned
            # for ~one workgroup per CU (the default, right for a launch that has the chip to itself) makes two such chains time-slice every
            # CU; planned for 128 workgroups each (forward, data and filter gradients: functional.target_workgroups is remembered by the
            # layer for its backward) they run on different CUs: 1.125 -> 1.09 ms per CIFAR iteration.  GGAN_NETS_TARGET_WGS=0: default plan.
            with F.target_workgroups(nets_target):
                return self._forward_nets(feed, c, B, xs, zs, p_z, fork)
        return self._forward_nets(feed, c, B, xs, zs, p_z, fork)

    def _forward_nets(self, feed, c, B, xs, zs, p_z, fork):
        early, self._early = self._early, False
        p_z, p_z_gen = F.fanout(p_z, 2) if (c.K and c.fuse) else (p_z, p_z)          # (critics | Generator)
        fan_q = (lambda q: F.fanout(q, 2)) if (c.K and c.fuse) else (lambda q: (q, q))     # (critics | HyperExtractor)
        if fork and early:
            # begin_nets() forked before the noise launch: the Extractor pass (which reads no noise) is a ROOT branch of the step
            # graph on the second stream, the Generator pass follows the noise launch on this one.  The branches are joined
            # where their results are first read (join_side): real_x before the critic's first layer, q_z before its z path --
            # a cross-queue dependency that completed long ago costs nothing, one that completes last costs ~11 us of idle chip
            cur = torch.cuda.current_stream(p_z.device)
            ev_noise = torch.cuda.Event()
            ev_noise.record(cur)                      # (p_z exists on this stream from here on: the critic's z path reads it there)
            self._noise_event = ev_noise
            with torch.cuda.stream(self._side):
                if c.dataset == 'face' or c.K:
                    self._side.wait_event(ev_noise)       # (dequantisation noise of the 64x64 scripts / Gumbel noise of the mixture scripts)
                real_x = self.real_x(feed, xs[1], defer=True)
                ev_x = torch.cuda.Event()
                q_src = self.Extractor(real_x, zs[1], after_first=lambda: ev_x.record(self._side))
                q_z, q_z_h = fan_q(q_src)
                if isinstance(real_x, F.PendingCast):
                    real_x = real_x.out
                out = dict(real_x=real_x, q_z=q_z, p_z=p_z, q_z_src=q_src)
                if c.K:
                    ks = F.RowSlot(feed['k_pair'], B, 2 * B) if (c.batch_critic and 'k_pair' in feed) else None
                    _, out['q_k'] = self.HyperExtractor(q_z_h, feed['gumbel_u'], ks)
                ev_end = torch.cuda.Event()
                ev_end.record(self._side)
            self._pending_join = [cur, ev_x, ev_end]
            out['fake_x'] = self.Generator(p_z_gen, xs[0])
            return out
        if fork:
            cur = torch.cuda.current_stream(p_z.device)
            if self._side is None:
                self._side = F.shared_stream(p_z.device, 'side')
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                fake_x = self.Generator(p_z_gen, xs[0])
        real_x = self.real_x(feed, xs[1], defer=not c.agg)
        if c.agg:                      # (no critic, no fake_x: TF prunes the Generator(p_z) branch these modes never fetch)
            q_z, q_mean, q_std = self.Extractor(real_x, eps=feed['q_eps'])
            return dict(real_x=real_x, q_z=q_z, q_z_mean=q_mean, q_z_std=q_std, p_z=p_z)
        q_src = self.Extractor(real_x, zs[1])
        q_z, q_z_h = fan_q(q_src)
        if isinstance(real_x, F.PendingCast):
            real_x = real_x.out
        out = dict(real_x=real_x, q_z=q_z, p_z=p_z, q_z_src=q_src)
        if c.K:
            ks = F.RowSlot(feed['k_pair'], B, 2 * B) if (c.batch_critic and 'k_pair' in feed) else None
            _, q_k = self.HyperExtractor(q_z_h, feed['gumbel_u'], ks)
            out['q_k'] = q_k
        if fork:
            cur.wait_stream(self._side)
        else:
            fake_x = self.Generator(p_z_gen, xs[0])
        out['fake_x'] = fake_x
        return out

    def forward(self, feed, which=None, nets=None):
        """which='gen'|'disc' builds only what that session.run fetches (TF prunes the rest: the gradient penalty
        is not part of gen_cost); None builds everything.  nets: a forward_nets() result to continue from."""
        c = self.cfg
        out = dict(nets) if nets is not None else self.forward_nets(feed)
        defer = c.batch_critic and which in ('gen', 'disc') and not (c.latent_critic or c.agg or c.mode == 'vegan-mmd')
        if not defer:
            self.join_side()
        if c.latent_critic:
            return self._forward_latent(feed, which, out)
        if c.agg:
            rec = 1. * lib.utils.distance.distance(out['real_x'], self.Generator(out['q_z']), 'l2')
            gen_params, _ = self._var_lists()
            K = lib.objs.kl_aggregated
            args = (out['q_z_mean'], out['q_z_std'], None, None, rec, gen_params, c.z_samples)
            kw = dict(lr=c.lr, beta1=c.beta1, draws=(feed['kl_k'], feed['kl_eps'], feed['kl_zp']))
            if c.mode == 'vegan-ikl':
                gen_cost, gen_op = K.vegan_ikl(*args, c.dim_latent, c.lamb, **kw)
            else:
                gen_cost, gen_op = (K.vegan_kl if c.mode == 'vegan-kl' else K.vegan_jsd)(*args, c.B, c.dim_latent, c.lamb, **kw)
            out.update(rec_penalty=rec, gen_cost=gen_cost, gen_train_op=gen_op, disc_cost=None, disc_train_op=None)
            return out
        if c.mode == 'vegan-mmd':
            rec = 1. * lib.utils.distance.distance(out['real_x'], self.Generator(out['q_z']), 'l2')
            gen_params, _ = self._var_lists()
            gen_cost, gen_op = lib.objs.mmd.vegan_mmd(out['q_z'], out['p_z'], rec, gen_params, c.B, c.lamb, lr=c.lr, beta1=c.beta1)
            out.update(rec_penalty=rec, gen_cost=gen_cost, gen_train_op=gen_op, disc_cost=None, disc_train_op=None)
            return out
        real_x, q_z, p_z, fake_x = out['real_x'], out['q_z'], out['p_z'], out['fake_x']
        if c.K:
            onehot, q_k = feed['k_onehot'], out['q_k']
        J = lib.objs.gan_inference
        J.ONLY[0] = which            # TF prunes the cost a session.run does not fetch; so do we
        batched = c.batch_critic and which in ('gen', 'disc')
        # a generator step owns only Generator/Extractor variables: hand the critic its weights without gradient edges
        # wali-gp critic steps: the penalty pass (third critic pass on the interpolates, its data gradient, later their second
        # derivatives: a chain of ~45 launches that shares nothing but weights with the [fake; real] pass) goes to a stream of its own
        # while a step graph is built -- issued first, so that it runs beside the main pass in both directions (autograd keeps every
        # backward node on the stream of its forward)
        gp_early = None
        # (round 6) the penalty's VALUE joins the cost late: the critic head that is going to write the cost with its backward launch (head_hint)
        # leaves the penalty term out, the main stream is NOT joined with the penalty stream in front of the cost, and the Trainer adds the
        # term behind the backward pass (functional.LATE_EXT / add_late_terms) -- the [fake; real] pass's backward then starts where its
        # forward ends instead of where the penalty chain's first-order phase ends (headline -1.5 %; GGAN_NO_LATE_PENALTY: as before)
        late_gp = (c.mode == 'wali-gp' and which == 'disc' and batched and self.fork_nets and self.fork_now and real_x.is_cuda
                   and not os.environ.get('GGAN_NO_FORK_GP') and getattr(self, 'head_hint', False) and c.fuse and not c.K
                   and not os.environ.get('GGAN_NO_HEAD_HINT') and not os.environ.get('GGAN_NO_LATE_PENALTY'))
        if (c.mode == 'wali-gp' and which == 'disc' and batched and self.fork_nets and self.fork_now and real_x.is_cuda
                and not os.environ.get('GGAN_NO_FORK_GP')):
            # (the SECOND stream, behind the Extractor branch whose q_z it reads -- not a third one: every further stream of the process
            #  shifts how the graphs' branches map onto the hardware queues, and the workload run next in the same process measured
            #  2.6 % slower.  The critic's short z path, which otherwise rides on that stream, stays on the main one in these steps.)
            cur = torch.cuda.current_stream(real_x.device)
            if self._side is None:
                self._side = F.shared_stream(real_x.device, 'side')
            self._gp_stream = self._side
            self._gp_stream.wait_stream(cur)                          # fake_x, p_z (Generator branch)
            with torch.cuda.stream(self._gp_stream):
                F.LATE_EXT[0] = late_gp
                try:
                    gp_early = self._penalty(J, batched, real_x, fake_x, q_z, p_z, feed)
                finally:
                    F.LATE_EXT[0] = False
        # MODE ali on the batched critic: the cost of this step is known before the critic runs -- sigmoid cross-entropy of its [fake; real]
        # logits with labels (1, 0) in a generator step, (0, 1) in a critic step (tflib/objs/gan_inference.py:47-79) -- and the caller
        # (engine.Trainer, head_hint) runs the backward at once: the critic head leaves that cost's gradient behind with its forward
        hint, hkind = None, 'bce'
        if (getattr(self, 'head_hint', False) and batched and c.fuse and real_x.is_cuda and not os.environ.get('GGAN_NO_HEAD_HINT')):
            if c.mode == 'ali' and not c.K:
                fl, rl = (1.0, 0.0) if which == 'gen' else (0.0, 1.0)
                hint = [(c.B, fl, 1.0), (c.B, rl, 1.0)]
            elif c.mode == 'local_ep' and c.K:
                # the mixture scripts: two critic heads (on codes, on (x, z) pairs), each evaluated on [fake; real]; their four terms weigh 1/2
                # each (tflib/objs/gan_inference.py:81-119).  Both heads read the same hint: each one's own two terms
                fl, rl = (1.0, 0.0) if which == 'gen' else (0.0, 1.0)
                hint = [(c.B, fl, 0.5), (c.B, rl, 0.5)]
            elif c.mode == 'wali-gp' and not c.K:
                # the Wasserstein costs (tflib/objs/gan_inference.py:28-45): -mean(fake) + mean(real) for the generator step, the opposite
                # signs (+ the one-element penalty, which only enters the cost's value) for the critic step
                sg = -1.0 if which == 'gen' else 1.0
                hint, hkind = [(c.B, 0.0, sg), (c.B, 0.0, -sg)], 'mean'
        with (lib.frozen('Discriminator') if which == 'gen' else lib.frozen()), F.head_bce_hint(hint, hkind):
            d_fake, d_real = self._critic(batched, real_x, q_z, p_z, fake_x, onehot if c.K else None, q_k if c.K else None,
                                          detach=which == 'disc')
        if gp_early is not None and not late_gp:
            torch.cuda.current_stream(real_x.device).wait_stream(self._gp_stream)
        gen_params, disc_params = self._var_lists()
        rec_penalty = None
        if which != 'disc' and c.mode in ('alice', 'alice-z', 'alice-x', 'local_epce'):
            D = lib.utils.distance.distance
            if c.mode != 'alice-x':
                rec_penalty = 1. * D(real_x, self.Generator(q_z), 'l2')
            if c.mode in ('alice', 'alice-x'):
                rz = 1. * D(p_z, self.Extractor(fake_x), 'l2')
                rec_penalty = rz if rec_penalty is None else rec_penalty + rz
            out['rec_penalty'] = rec_penalty
        if c.mode == 'local_epce':
            res = J.local_epce(d_fake, d_real, rec_penalty, gen_params, disc_params, lr=c.lr, beta1=c.beta1)
        elif c.mode.startswith('alice'):
            res = J.alice(d_fake, d_real, rec_penalty, gen_params, disc_params, lr=c.lr, beta1=c.beta1)
        elif c.K:
            res = J.local_ep(d_fake, d_real, gen_params, disc_params, lr=c.lr, beta1=c.beta1)
        elif c.mode == 'wali':           # RMSProp, critic weights clipped inside the critic's update kernel
            r = J.wali(d_fake, d_real, gen_params, disc_params)
            res = (r[0], r[1], r[3], r[4])
        elif c.mode == 'wali-gp':
            if which == 'gen':
                gp = None                # not part of gen_cost; TF prunes the third critic pass
            else:
                # (the penalty pass reaches the critic's weights through second autograd leaves: the optimizer sums the two
                #  gradient contributions of every weight where it packs the bucket, not with an addition launch per weight)
                gp = gp_early if gp_early is not None else self._penalty(J, batched, real_x, fake_x, q_z, p_z, feed)
            F.LATE_EXT[0] = late_gp and gp is gp_early and gp is not None
            try:
                res = J.wali_gp(d_fake, d_real, gp, gen_params, disc_params)
            finally:
                F.LATE_EXT[0] = False
            out['gradient_penalty'] = gp
        else:
            res = J.ali(d_fake, d_real, gen_params, disc_params, lr=c.lr, beta1=c.beta1)
        J.ONLY[0] = None
        out.update(disc_fake=d_fake, disc_real=d_real, gen_cost=res[0], disc_cost=res[1],
                   gen_train_op=res[2], disc_train_op=res[3])
        return out

    def _penalty(self, J, batched, real_x, fake_x, q_z, p_z, feed):
        # (the penalty pass reaches the critic's weights through second autograd leaves: the optimizer sums the two
        #  gradient contributions of every weight where it packs the bucket, not with an addition launch per weight)
        with (lib.second_leaf() if (batched and not os.environ.get('GGAN_NO_SECOND_LEAF')) else lib.frozen()):
            return J.gradient_penalty(lambda xx, zz: self.Discriminator(xx, zz, twice=True), real_x, fake_x.detach() if batched else fake_x,
                                      q_z.detach() if batched else q_z, p_z.detach() if batched else p_z, feed['alpha'])

    def _critic(self, batched, real_x, q_z, p_z, fake_x, onehot, q_k, detach=True):
        """critic logits of the fake and the real pair.  batched: the critic is evaluated ONCE on [fake; real] (its rows
        are independent: no BatchNorm in the critics).  detach=True (critic steps): no gradient w.r.t. the
        generator/extractor outputs (TF prunes those paths too); detach=False (generator steps): gradients flow to fake_x,
        p_z, q_z, q_k, and the conv stack's data-gradient is needed for the fake half only (grad_rows)."""
        c = self.cfg
        if not batched:
            if c.K:
                return ([self.HyperDiscriminator(p_z, onehot), self.Discriminator(fake_x, p_z)],
                        [self.HyperDiscriminator(q_z, q_k), self.Discriminator(real_x, q_z)])
            return self.Discriminator(fake_x, p_z), self.Discriminator(real_x, q_z)
        if detach:
            fake_x, p_z, q_z = fake_x.detach(), p_z.detach(), q_z.detach()
            q_k = q_k.detach() if c.K else None
        B = fake_x.shape[0]
        # grad_rows leaves the real rows of the image gradient unwritten: legal only while the real half is data
        assert detach or not real_x.requires_grad, 'batched critic with grad_rows: real_x must not require a gradient'
        self.join_side(x_only=True)
        x_cat, z_cat = F.JoinRows.apply(fake_x, real_x), F.JoinRows.apply(p_z, q_z)
        z_cat, z_cat_h = F.fanout(z_cat, 2) if (c.K and c.fuse) else (z_cat, z_cat)          # (joint critic | mixture critic)
        z_out = None
        pj = self._pending_join
        gp_on_side = self._gp_stream is not None and self.cfg.mode == 'wali-gp' and detach and self.fork_now
        if pj is not None and not os.environ.get('GGAN_NO_Z_PATH_FORK') and not gp_on_side:
            # the Extractor pass is still running on the second stream: the critic's z path (a Linear on [p_z ; q_z]: two short
            # launches, and two more in the backward pass) follows it THERE, off this stream's chain of conv launches; the join
            # before the critic's tail then waits for z_out instead of q_z
            with torch.cuda.stream(self._side):
                self._side.wait_event(self._noise_event)
                z_out = self._lin('Discriminator.z1', c.dim_latent, 512, z_cat, LRELU)
                ev = torch.cuda.Event()
                ev.record(self._side)
                pj[2] = ev
        fork_h = bool(c.K) and self.fork_nets and self.fork_now and x_cat.is_cuda and not os.environ.get('GGAN_NO_FORK_HYPER')
        if fork_h:
            # the mixture critic on (z, k) is a chain of ~12 short launches per direction that reads nothing of the image critic: it
            # runs on the second stream beside the conv stack (autograd keeps each pass's backward on the stream of its forward)
            cur = torch.cuda.current_stream(x_cat.device)
            if pj is None:          # (else the second stream still carries the Extractor branch and the z path: the mixture critic follows them)
                self.join_side()
                self._side.wait_stream(cur)
            ev_z = torch.cuda.Event()
            with torch.cuda.stream(self._side):
                if z_out is None and not os.environ.get('GGAN_NO_Z_PATH_FORK'):
                    z_out = self._lin('Discriminator.z1', c.dim_latent, 512, z_cat, LRELU)      # (the image critic's z path too)
                    ev_z.record(self._side)
                else:
                    ev_z = None
                h = self.HyperDiscriminator(z_cat_h, F.JoinRows.apply(onehot, q_k))
            before = (lambda: cur.wait_event(ev_z)) if ev_z is not None else self.join_side
            d = self.Discriminator(x_cat, z_cat, grad_rows=None if detach else B, before_z=before, z_out=z_out)
        else:
            d = self.Discriminator(x_cat, z_cat, grad_rows=None if detach else B, before_z=self.join_side, z_out=z_out)
        if fork_h:
            cur.wait_stream(self._side)
        elif c.K:
            h = self.HyperDiscriminator(z_cat_h, F.JoinRows.apply(onehot, q_k))
        if c.K:
            (hf, hr), (df, dr) = F.SplitRows.apply(h, B), F.SplitRows.apply(d, B)
            return [hf, df], [hr, dr]
        return F.SplitRows.apply(d, B)

"""Net definitions of the reference's driver scripts, restated on the oracle tape.

TEST INFRASTRUCTURE (see oracle/__init__.py: composition pinned by tests/golden/reference_trace.json, TF primitives unpinned).

Follows
  gan_inference_cifar10.py:133-255      (Generator / Extractor / Discriminator, 32x32x3)
  gmgan_inference_cifar10.py:114-301    (+ HyperGenerator / HyperExtractor / HyperDiscriminator)
  gmgan_inference_mnist.py:166-300      (28x28x1: crop [:,:,:7,:7], sigmoid output, float input)
  gmgan_inference_face.py:82-200        (64x64x3: 4 conv layers, DIM 32, no BatchNorm)
tf.layers.dropout is called without training=True everywhere => identity.
Parameter names/shapes are the lib.param keys of SURVEY.md Appendix C.
"""
import numpy as np
from . import tape as tp


class Cfg(object):
    def __init__(self, dataset='cifar10', batch_size=64, n_coms=0, dim=None, dim_latent=128,
                 bn=None, mode_k='CONCRETE', temp=0.1, latent_critic=False, learn_std=False, z_samples=100, script=None, critic=True):
        self.dataset = dataset
        self.B = batch_size
        self.K = n_coms                      # 0 => plain gan_inference_* (no GMM prior)
        self.dim_latent = dim_latent
        self.temp = temp
        self.mode_k = mode_k
        self.learn_std, self.z_samples = learn_std, z_samples   # MODE vegan-kl/ikl/jsd: TYPE_Q = 'learn_std', Z_SAMPLES (gan_inference_cifar10.py:40-43)
        self.latent_critic = latent_critic     # MODE vegan / vegan-wgan-gp: the critic sees codes only (gan_inference_cifar10.py:192-222)
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
        self.critic = critic                   # False: MODE vegan-mmd / vegan-kl / -ikl / -jsd define no Discriminator (gan_inference_cifar10.py:224-225)
        # gan_inference_mnist.py:215-250: the joint critic of THAT script has BatchNorm after conv 2 / 3 (BN_FLAG) and a second
        # Linear on the z path ('Discriminator.2', sharing its prefix with the conv layer) and on the joint path ('Discriminator.zx2');
        # gmgan_inference_mnist.py's critic is the plain one
        self.critic_deep = dataset == 'mnist' and not n_coms and not latent_critic
        self.top = self.dim * 2 ** (self.nl - 1)      # channels at the 4x4 stage
        self.flat = 16 * self.top
        self.output_dim = self.C * self.S * self.S


# --------------------------------------------------------------------------------------
# initialisers -- consumed in the order the reference scripts create params
# --------------------------------------------------------------------------------------
def _uniform(rng, stdev, size):
    return rng.uniform(-stdev * np.sqrt(3), stdev * np.sqrt(3), size=size).astype('float32')


def conv_init(rng, cin, cout, k=5, stride=2):
    """tflib/ops/conv2d.py:62-86 (he_init=True)."""
    fan_in = cin * k ** 2
    fan_out = cout * k ** 2 / (stride ** 2)
    return _uniform(rng, np.sqrt(4. / (fan_in + fan_out)), (k, k, cin, cout))


def deconv_init(rng, cin, cout, k=5, stride=2):
    """tflib/ops/deconv2d.py:51-71 (he_init=True); layout [k,k,out,in]."""
    fan_in = cin * k ** 2 / (stride ** 2)
    fan_out = cout * k ** 2
    return _uniform(rng, np.sqrt(4. / (fan_in + fan_out)), (k, k, cout, cin))


def linear_init(rng, nin, nout):
    """tflib/ops/linear.py:55-60 (initialization=None -> glorot branch)."""
    return _uniform(rng, np.sqrt(2. / (nin + nout)), (nin, nout))


def init_params(cfg, seed=0):
    """name -> float32 array, reference shapes/layouts."""
    rng = np.random.RandomState(seed)
    P = {}
    d, nl = cfg.dim, cfg.nl
    chans = [cfg.C] + [d * 2 ** i for i in range(nl)]            # e.g. 3,64,128,256
    names_dec = ['2', '3', '4', '5'] if nl == 4 else ['2', '3', '5']

    def lin(name, nin, nout):
        P[name + '.W'] = linear_init(rng, nin, nout)
        P[name + '.b'] = np.zeros(nout, 'float32')

    def bn(name, c, fused=True):
        shp = (c,) if fused else (1, c)
        P[name + '.offset'] = np.zeros(shp, 'float32')
        P[name + '.scale'] = np.ones(shp, 'float32')
        if fused:
            P[name + '.moving_mean'] = np.zeros(shp, 'float32')
            P[name + '.moving_variance'] = np.ones(shp, 'float32')

    if cfg.K:
        P['Generator.Hyper.Mu'] = rng.normal(size=(cfg.K, cfg.dim_latent)).astype('float32')
    # Extractor
    for i in range(nl):
        P['Extractor.%d.Filters' % (i + 1)] = conv_init(rng, chans[i], chans[i + 1])
        P['Extractor.%d.Biases' % (i + 1)] = np.zeros(chans[i + 1], 'float32')
        if cfg.bn and i > 0:
            bn('Extractor.BN%d' % (i + 1), chans[i + 1])
    if getattr(cfg, 'learn_std', False):
        lin('Extractor.Std', cfg.flat, cfg.dim_latent)            # created before Extractor.Output (gan_inference_cifar10.py:174,181)
    lin('Extractor.Output', cfg.flat, cfg.dim_latent)
    # Generator
    lin('Generator.Input', cfg.dim_latent, cfg.flat)
    if cfg.bn:
        bn('Generator.BN1', cfg.flat, fused=False)
    dch = chans[::-1]                                           # 256,128,64,3
    for i, nm in enumerate(names_dec):
        P['Generator.%s.Filters' % nm] = deconv_init(rng, dch[i], dch[i + 1])
        P['Generator.%s.Biases' % nm] = np.zeros(dch[i + 1], 'float32')
        if cfg.bn and i < len(names_dec) - 1:
            bn('Generator.BN%s' % nm, dch[i + 1])
    # Discriminator(s)
    if getattr(cfg, 'latent_critic', False):
        for i, (nm, nin, nout) in enumerate([('Input', cfg.dim_latent, 1024), ('2', 1024, 512), ('3', 512, 256), ('4', 256, 256)]):
            lin('Discriminator.' + nm, nin, nout)
            if cfg.bn:
                bn('Discriminator.BN%d' % (i + 1), nout, fused=False)
        lin('Discriminator.Output', 256, 1)
        return P
    if not getattr(cfg, 'critic', True):
        return P
    deep = getattr(cfg, 'critic_deep', False)
    for i in range(nl):
        P['Discriminator.%d.Filters' % (i + 1)] = conv_init(rng, chans[i], chans[i + 1])
        P['Discriminator.%d.Biases' % (i + 1)] = np.zeros(chans[i + 1], 'float32')
        if deep and cfg.bn and i > 0:
            bn('Discriminator.BN%d' % (i + 1), chans[i + 1])
    lin('Discriminator.z1', cfg.dim_latent, 512)
    if deep:
        lin('Discriminator.2', 512, 512)
    lin('Discriminator.zx1', cfg.flat + 512, 512)
    if deep:
        lin('Discriminator.zx2', 512, 512)
    lin('Discriminator.Output', 512, 1)
    if cfg.K:
        lin('Discriminator.HyperInput', cfg.dim_latent + cfg.K, 512)
        lin('Discriminator.Hyper2', 512, 512)
        lin('Discriminator.Hyper3', 512, 512)
        lin('Discriminator.HyperOutput', 512, 1)
    return P


def params_with_name(P, sub):
    """lib.params_with_name (tflib/__init__.py:35-36): substring match."""
    return [n for n in P if sub in n]


def trainable(names):
    return [n for n in names if not (n.endswith('.moving_mean') or n.endswith('.moving_variance'))]


# --------------------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------------------
def Linear(P, name, x):
    return tp.add(tp.matmul(x, P[name + '.W']), P[name + '.b'])


def Conv2D(P, name, x, stride=2):
    b = tp.reshape(P[name + '.Biases'], (1, -1, 1, 1))
    return tp.add(tp.conv2d(x, P[name + '.Filters'], stride, 'SAME'), b)


def Deconv2D(P, name, x):
    b = tp.reshape(P[name + '.Biases'], (1, -1, 1, 1))
    return tp.add(tp.deconv2d(x, P[name + '.Filters'], 2, 'SAME'), b)


def Batchnorm(P, name, axes, x):
    return tp.batchnorm_train(x, P[name + '.scale'], P[name + '.offset'], axes, 1e-5)


# --------------------------------------------------------------------------------------
# nets
# --------------------------------------------------------------------------------------
def Generator(cfg, P, noise):
    out = Linear(P, 'Generator.Input', noise)
    if cfg.bn:
        out = Batchnorm(P, 'Generator.BN1', [0], out)
    out = tp.relu(out)
    out = tp.reshape(out, (-1, cfg.top, 4, 4))
    names = ['2', '3', '4', '5'] if cfg.nl == 4 else ['2', '3', '5']
    for i, nm in enumerate(names):
        out = Deconv2D(P, 'Generator.' + nm, out)
        if i < len(names) - 1:
            if cfg.bn:
                out = Batchnorm(P, 'Generator.BN' + nm, [0, 2, 3], out)
            out = tp.relu(out)
            if cfg.dataset == 'mnist' and nm == '2':
                out = tp.slice_axis(tp.slice_axis(out, 2, 0, 7), 3, 0, 7)   # output[:,:,:7,:7]
    out = tp.tanh(out) if cfg.out_act == 'tanh' else tp.sigmoid(out)
    return tp.reshape(out, (-1, cfg.output_dim))


def Extractor(cfg, P, x, eps=None):
    out = tp.reshape(x, (-1, cfg.C, cfg.S, cfg.S))
    for i in range(cfg.nl):
        out = Conv2D(P, 'Extractor.%d' % (i + 1), out)
        if cfg.bn and i > 0:
            out = Batchnorm(P, 'Extractor.BN%d' % (i + 1), [0, 2, 3], out)
        out = tp.leaky_relu(out)
    out = tp.reshape(out, (-1, cfg.flat))
    if eps is not None:        # TYPE_Q = 'learn_std' (gan_inference_cifar10.py:173-188): q_z = mean + eps * exp(Linear 'Extractor.Std')
        std = tp.exp(Linear(P, 'Extractor.Std', out))
        mean = Linear(P, 'Extractor.Output', out)
        return tp.add(mean, tp.mul(eps, std)), mean, std
    return Linear(P, 'Extractor.Output', out)


def Discriminator(cfg, P, x, z):
    """gan_inference_cifar10.py:226-255 / gmgan_inference_cifar10.py:270-301 (plain); gan_inference_mnist.py:215-250 (cfg.critic_deep:
    BatchNorm after conv 2 / 3 under BN_FLAG, 'Discriminator.2' Linear 512->512 on the z path, 'Discriminator.zx2' on the joint path)."""
    deep = getattr(cfg, 'critic_deep', False)
    out = tp.reshape(x, (-1, cfg.C, cfg.S, cfg.S))
    for i in range(cfg.nl):
        out = Conv2D(P, 'Discriminator.%d' % (i + 1), out)
        if deep and cfg.bn and i > 0:
            out = Batchnorm(P, 'Discriminator.BN%d' % (i + 1), [0, 2, 3], out)
        out = tp.leaky_relu(out)
    out = tp.reshape(out, (-1, cfg.flat))
    zo = tp.leaky_relu(Linear(P, 'Discriminator.z1', z))
    if deep:
        zo = tp.leaky_relu(Linear(P, 'Discriminator.2', zo))
    out = tp.concat([out, zo], 1)
    out = tp.leaky_relu(Linear(P, 'Discriminator.zx1', out))
    if deep:
        out = tp.leaky_relu(Linear(P, 'Discriminator.zx2', out))
    return tp.reshape(Linear(P, 'Discriminator.Output', out), (-1,))


def LatentDiscriminator(cfg, P, z, noise):
    """gan_inference_cifar10.py:192-222 (MODE vegan / vegan-wgan-gp): an MLP on codes with Gaussian noise layers (std .3 on the
    input, .5 after the first three hidden layers).  noise: the four N(0,1) draws of this call, scaled here."""
    dt = z.v.dtype
    out = tp.add(z, tp.T(np.asarray(noise[0], dtype=dt) * dt.type(.3)))
    for i, nm in enumerate(['Input', '2', '3', '4']):
        out = Linear(P, 'Discriminator.' + nm, out)
        if cfg.bn:
            out = Batchnorm(P, 'Discriminator.BN%d' % (i + 1), [0], out)
        out = tp.leaky_relu(out)
        if i < 3:
            out = tp.add(out, tp.T(np.asarray(noise[i + 1], dtype=dt) * dt.type(.5)))
    return tp.reshape(Linear(P, 'Discriminator.Output', out), (-1,))


def HyperDiscriminator(cfg, P, z, k):
    out = tp.concat([z, k], 1)
    for nm in ('HyperInput', 'Hyper2', 'Hyper3'):
        out = tp.leaky_relu(Linear(P, 'Discriminator.' + nm, out))
    return tp.reshape(Linear(P, 'Discriminator.HyperOutput', out), (-1,))


def HyperGenerator(cfg, P, onehot_k, noise):
    """gmgan_inference_cifar10.py:150-153."""
    return tp.add(tp.matmul(onehot_k, P['Generator.Hyper.Mu']), noise)


def HyperExtractor(cfg, P, z, gumbel_u):
    """gmgan_inference_cifar10.py:156-173, MODE_K='CONCRETE' (:81), TEMP (:85-86)."""
    mu = P['Generator.Hyper.Mu']
    B, D = z.shape
    diff = tp.add(tp.reshape(z, (B, 1, D)), tp.neg(tp.reshape(mu, (1, cfg.K, D))))
    logits = tp.add(tp.scale(tp.reduce_sum(tp.square(diff), (2,)), -0.5),
                    float(np.log(np.float32(1.0) / np.float32(cfg.K))))
    u = np.asarray(gumbel_u, dtype=z.v.dtype)
    g = -np.log(-np.log(u + 1e-20) + 1e-20)                  # sample_gumbel, :117-120
    k = tp.softmax(tp.scale(tp.add(logits, tp.T(g)), 1.0 / cfg.temp))
    return logits, k

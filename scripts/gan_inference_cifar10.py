"""Counterpart of the reference's gan_inference_cifar10.py for this package's tflib: the same UPPERCASE hyper-parameter block
(gan_inference_cifar10.py:39-62), nets and step order; runs on one MI355X.  `python scripts/gan_inference_cifar10.py [ITERS]`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphical_gan_amd import run
from graphical_gan_amd.models import Config

DATASET = 'cifar10'
MODE = 'ali'  # ali, alice, alice-z, alice-x, wali, wali-gp, vegan, vegan-wgan-gp, vegan-mmd, vegan-kl, vegan-ikl, vegan-jsd

if MODE in ('vegan', 'vegan-wgan-gp', 'vegan-kl', 'vegan-jsd', 'vegan-ikl'):   # gan_inference_cifar10.py: the code-space objectives
    BN_FLAG, DIM_LATENT = False, 8
else:
    BN_FLAG, DIM_LATENT = True, 128
Z_SAMPLES = 100  # MC samples of D(q(z) || p(z)) (vegan-kl / -ikl / -jsd)
BATCH_SIZE = 64
CRITIC_ITERS = 0 if MODE in ("vegan-mmd", "vegan-kl", "vegan-ikl", "vegan-jsd") else (5 if MODE in ("wali", "wali-gp", "vegan", "vegan-wgan-gp") else 1)
LR = {"wali-gp": 1e-4, "wali": 5e-5}.get(MODE, 2e-4)  # the wali objectives ignore the scripts' LR (gan_inference.py:4,28)
BETA1 = .5
ITERS = 200000  # number of iterations to train
DATA_DIR = os.environ.get('GGAN_DATA_DIR', '')
OUT_DIR = os.environ.get('GGAN_OUT_DIR', '')
SAVE_EVERY = 10000
LOG_EVERY = 100

if len(sys.argv) > 1:
    ITERS = int(sys.argv[1])
SETTINGS = {k: v for k, v in dict(globals()).items() if k.isupper() and k != 'SETTINGS'}
cfg = Config(DATASET, batch_size=BATCH_SIZE, n_coms=0, mode=MODE, dim_latent=DIM_LATENT, lr=LR, bn=BN_FLAG)
cfg.z_samples = Z_SAMPLES
run.train(SETTINGS, cfg)

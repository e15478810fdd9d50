"""Counterpart of the reference's gan_inference_face.py for this package's tflib: the same UPPERCASE hyper-parameter block
(gan_inference_face.py:30-50), nets and step order; runs on one MI355X.  `python scripts/gan_inference_face.py [ITERS]`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphical_gan_amd import run
from graphical_gan_amd.models import Config

DATASET = 'face'
MODE = 'ali'  # ali, alice, alice-z, alice-x, wali, wali-gp, vegan, vegan-wgan-gp, vegan-mmd, vegan-kl, vegan-ikl, vegan-jsd

DIM_LATENT = 128  # latent dimension
BATCH_SIZE = 128
CRITIC_ITERS = 0 if MODE in ("vegan-mmd", "vegan-kl", "vegan-ikl", "vegan-jsd") else (5 if MODE in ("wali", "wali-gp", "vegan", "vegan-wgan-gp") else 1)
LR = {"wali-gp": 1e-4, "wali": 5e-5}.get(MODE, 2e-4)  # the wali objectives ignore the scripts' LR (gan_inference.py:4,28)
BETA1 = .5
ITERS = 100000  # number of iterations to train
DATA_DIR = os.environ.get('GGAN_DATA_DIR', '')
OUT_DIR = os.environ.get('GGAN_OUT_DIR', '')
SAVE_EVERY = 10000
LOG_EVERY = 100

if len(sys.argv) > 1:
    ITERS = int(sys.argv[1])
SETTINGS = {k: v for k, v in dict(globals()).items() if k.isupper() and k != 'SETTINGS'}
cfg = Config(DATASET, batch_size=BATCH_SIZE, n_coms=0, mode=MODE, dim_latent=DIM_LATENT, lr=LR)
run.train(SETTINGS, cfg)

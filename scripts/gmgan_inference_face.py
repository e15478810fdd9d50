"""Counterpart of the reference's gmgan_inference_face.py for this package's tflib: the same UPPERCASE hyper-parameter block
(gmgan_inference_face.py:30-52), nets and step order; runs on one MI355X.  `python scripts/gmgan_inference_face.py [ITERS]`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphical_gan_amd import run
from graphical_gan_amd.models import Config

DATASET = 'face'
MODE = 'local_ep'  # local_ep, local_epce
N_COMS = 100  # mixture components of the latent prior
DIM_LATENT = 128  # latent dimension
BATCH_SIZE = 128
CRITIC_ITERS = 1
LR = 2e-4
BETA1 = .5
ITERS = 100000  # number of iterations to train
DATA_DIR = os.environ.get('GGAN_DATA_DIR', '')
OUT_DIR = os.environ.get('GGAN_OUT_DIR', '')
SAVE_EVERY = 10000
LOG_EVERY = 100

if len(sys.argv) > 1:
    ITERS = int(sys.argv[1])
SETTINGS = {k: v for k, v in dict(globals()).items() if k.isupper() and k != 'SETTINGS'}
cfg = Config(DATASET, batch_size=BATCH_SIZE, n_coms=N_COMS, mode=MODE, dim_latent=DIM_LATENT, lr=LR)
run.train(SETTINGS, cfg)

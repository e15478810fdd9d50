"""Counterpart of the reference's ssgan_inference_chairs.py for this package's tflib: the same UPPERCASE hyper-parameter block
(ssgan_inference_chairs.py:28-55), nets and step order; runs on one MI355X.  `python scripts/ssgan_inference_chairs.py [ITERS]`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphical_gan_amd import run
from graphical_gan_amd.models_ssgan import SSConfig, StateSpaceGAN

DATASET = 'chairs'  # rotating chairs: 31 RGB views of 64x64 per object, no class labels
MODE = 'local_ep'  # local_ep, local_epce-z, ali, alice-z
ALI_MODE = 'concat_x'  # concat_x, concat_z ('3dcnn' is one-channel, LEN 4 / 16 only: the moving-MNIST script)
POS_MODE = 'naive_mean_field'  # gsp, naive_mean_field, inverse, forward_inverse
OP_DYN_MODE = 'res_w'  # res, res_w
DIM_LATENT_G = 128  # global latent variable
DIM_LATENT_L = 8  # local latent variable
DIM = 32  # model size of frame generator
DIM_OP = 256  # model size of the dynamic operator
LEN = 31  # data length
N_C = 0  # (no labels)
CHANNELS = 3
LAMBDA = 0.1  # reconstruction weight (local_epce-z)
LR = 1e-4
BATCH_SIZE = 50
CRITIC_ITERS = 1
ITERS = 40000
DATA_DIR = os.environ.get('GGAN_DATA_DIR', '')
OUT_DIR = os.environ.get('GGAN_OUT_DIR', '')
SAVE_EVERY = 10000
LOG_EVERY = 100

if len(sys.argv) > 1:
    ITERS = int(sys.argv[1])
SETTINGS = {k: v for k, v in dict(globals()).items() if k.isupper() and k != 'SETTINGS'}
cfg = SSConfig(batch_size=BATCH_SIZE, length=LEN, dim=DIM, dim_op=DIM_OP, dim_g=DIM_LATENT_G, dim_l=DIM_LATENT_L, n_c=N_C,
               pos_mode=POS_MODE, op_dyn_mode=OP_DYN_MODE, lr=LR, mode=MODE, lamb=LAMBDA, ali_mode=ALI_MODE, channels=CHANNELS, dataset=DATASET)
run.train(SETTINGS, cfg, model=StateSpaceGAN(cfg))

"""Counterpart of the reference's gan_inference_mnist.py for this package's tflib: the same UPPERCASE hyper-parameter block
(gan_inference_mnist.py:31-70; `run.reference_block` holds it as data and derives the MODE-dependent constants as the script does), nets and
step order; runs on one MI355X.  `python scripts/gan_inference_mnist.py [ITERS]`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphical_gan_amd import run

MODE = 'ali'  # ali, alice, alice-z, alice-x, wali, wali-gp, vegan, vegan-wgan-gp, vegan-mmd, vegan-kl, vegan-ikl, vegan-jsd
SETTINGS = run.reference_block(__file__, MODE=MODE)
# edit the block here, e.g. SETTINGS['N_COMS'] = 10 -- or pass it to reference_block, which then derives N_VIS etc. from it
SETTINGS.update(DATA_DIR=os.environ.get('GGAN_DATA_DIR', ''), OUT_DIR=os.environ.get('GGAN_OUT_DIR', ''), SAVE_EVERY=10000, LOG_EVERY=100)
if len(sys.argv) > 1:
    SETTINGS['ITERS'] = int(sys.argv[1])
globals().update(SETTINGS)          # BATCH_SIZE, DIM, DIM_LATENT, CRITIC_ITERS, ... as module constants, as in the reference
run.train(SETTINGS, run.config(SETTINGS))

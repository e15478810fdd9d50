"""tflib/objs/mmd.py: the MMD objective of MODE vegan-mmd (gan_inference_cifar10.py:327-329).  `mix_rbf_mmd2` is one fused kernel
per direction (ggan_mix_rbf_mmd2_*); `vegan_mmd` returns (gen_cost, gen_train_op) -- there is no critic in this mode."""
from ... import functional as F
from ...optim import TrainOp, get_optimizer

SIGMAS = [2., 5., 10., 20., 40., 80.]


def mix_rbf_mmd2(X, Y, sigmas=SIGMAS, wts=None, biased=True):
    """tflib/objs/mmd.py:65-67 (the scripts only use the biased estimator)"""
    if not biased:
        raise NotImplementedError('unbiased MMD estimator (mmd.py:53-61) is not used by any script and is not built')
    return F.MixRbfMmd2.apply(X, Y, tuple(sigmas), tuple(wts) if wts is not None else None)


def vegan_mmd(q_z, p_z, rec_penalty, gen_params, batch_size, lamb, lr=2e-4, beta1=.5):
    """tflib/objs/mmd.py:69-80"""
    gen_cost = mix_rbf_mmd2(q_z, p_z) * float(lamb)
    gen_cost = gen_cost + rec_penalty
    gen_opt = get_optimizer('gen', gen_params, lr=lr, beta1=beta1, beta2=0.999)
    return gen_cost, TrainOp(gen_opt, gen_cost)

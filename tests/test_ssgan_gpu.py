"""-m gpu: the state-space GAN step (ssgan_inference_moving_mnist.py, MODE='local_ep') on the HIP path vs its float64
oracle restatement (oracle/ssgan.py), same injected weights / sequences / noise.  The product model evaluates the LEN-1
transition critics in one stacked call; the oracle follows the script literally (one call per time step), so this also
checks that collapsing the equal-weight BCE terms is exact.  Tolerances as tests/test_step_gpu.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(gpu, pos_mode, op_dyn_mode, graph, B=2, L=3, dim=4, channels=1, n_c=10, mode='local_ep', ali_mode='concat_x'):
    from graphical_gan_amd import tflib as lib, optim
    from graphical_gan_amd.engine import Trainer
    from graphical_gan_amd.models_ssgan import SSConfig, StateSpaceGAN
    from oracle import ssgan as O
    kw = dict(batch_size=B, length=L, dim=dim, dim_op=16, dim_g=8, dim_l=4, pos_mode=pos_mode, op_dyn_mode=op_dyn_mode, channels=channels,
              n_c=n_c, mode=mode, ali_mode=ali_mode)
    ocfg = O.Cfg(**kw)
    P0 = O.init_params(ocfg, seed=0)
    rng = np.random.default_rng(5)
    for k in P0:
        if k.endswith('.b') or k.endswith('.Biases'):
            P0[k] = (0.1 * rng.standard_normal(P0[k].shape)).astype(np.float32)
    optim.reset_optimizers()
    lib.delete_all_params()
    cfg = SSConfig(dataset='chairs' if channels == 3 else 'moving_mnist', **kw)
    tr = Trainer(cfg, device=gpu, graph=graph, inject_noise=True, model=StateSpaceGAN(cfg))
    tr.load_params(P0)
    return ocfg, P0, cfg, tr


@pytest.mark.parametrize('pos_mode,op_dyn_mode,channels,n_c,mode', [
    ('naive_mean_field', 'res', 1, 10, 'local_ep'), ('gsp', 'res_w', 1, 10, 'local_ep'), ('inverse', 'res', 1, 10, 'local_ep'),
    ('naive_mean_field', 'res_w', 3, 0, 'local_ep'),            # ssgan_inference_chairs.py
    ('naive_mean_field', 'res', 1, 10, 'local_epce-z'),         # + LAMBDA * l2(real_x, rec_x)
    ('naive_mean_field', 'res', 1, 10, 'ali'), ('gsp', 'res', 1, 10, 'alice-z'),       # one critic on the whole sequence (concat_x)
    ('naive_mean_field', 'res', 1, 10, 'ali:concat_z'),
    ('naive_mean_field', 'res', 1, 10, 'ali:3dcnn'), ('gsp', 'res', 1, 10, 'alice-z:3dcnn')])    # Conv3D critic (LEN 4)
def test_ssgan_first_step_costs_and_grads(gpu, pos_mode, op_dyn_mode, channels, n_c, mode):
    import torch
    from oracle import ssgan as O, tape as tp
    mode, _, ali_mode = mode.partition(':')
    ocfg, P0, cfg, tr = _mk(gpu, pos_mode, op_dyn_mode, False, channels=channels, n_c=n_c, mode=mode, ali_mode=ali_mode or 'concat_x',
                            **(dict(L=4) if ali_mode == '3dcnn' else {}))
    feed = O.make_feed(ocfg, np.random.default_rng(3))
    Pt = {k: tp.T(v.astype(np.float64)) for k, v in P0.items()}
    oout = O.forward(ocfg, Pt, feed)
    tr.set_feed(feed)
    for which in ('gen', 'disc'):
        out = tr.model.forward(tr.feed, which)
        oc, c = float(oout[which + '_cost'].v), float(out[which + '_cost'].detach())
        assert abs(c - oc) <= 1e-5 * max(1.0, abs(oc)), (which, c, oc)
        opt = out[which + '_train_op'].optimizer
        grads = torch.autograd.grad(out[which + '_cost'], opt.params, allow_unused=True)
        names = [p.param_name for p in opt.params]
        ogs = tp.grad(oout[which + '_cost'], [Pt[n] for n in names])
        gmax = max(float(np.abs(g.v).max()) for g in ogs if g is not None)
        for n, g, og in zip(names, grads, ogs):
            if og is None:
                assert g is None or float(g.abs().max()) == 0.0, n
                continue
            assert g is not None, n
            err = float(np.abs(g.cpu().numpy().astype(np.float64) - og.v).max())
            # (floor: bias gradients of the critic heads are near-cancelling sums of O(gmax) terms)
            assert err <= 1e-4 * max(float(np.abs(og.v).max()), 5e-3 * gmax), (which, n, err)
    # the frame generator's output itself
    fx = tr.model.forward_nets(tr.feed)['fake_x'].detach().cpu().numpy()
    assert np.abs(fx - oout['fake_x'].v).max() <= 1e-5


@pytest.mark.parametrize('graph,chairs,mode', [(False, False, 'local_ep'), (True, False, 'local_ep'), (True, True, 'local_ep'),
                                               (True, False, 'local_epce-z'), (True, False, 'alice-z'), (True, False, 'ali:3dcnn')],
                         ids=['eager', 'hipgraph', 'hipgraph-chairs', 'hipgraph-epce-z', 'hipgraph-alice-z', 'hipgraph-ali-3dcnn'])
def test_ssgan_trajectory(gpu, graph, chairs, mode):
    """4 iterations (critic step, then gen + critic) with TF-Adam on both sides: costs and every weight."""
    from oracle import ssgan as O
    mode, _, ali_mode = mode.partition(':')
    kw = dict(channels=3, n_c=0) if chairs else (dict(ali_mode=ali_mode, L=4) if ali_mode else {})
    ocfg, P0, cfg, tr = _mk(gpu, 'naive_mean_field', 'res_w' if chairs else 'res', graph, mode=mode, **kw)
    rng = np.random.default_rng(9)
    feeds = [O.make_feed(ocfg, rng) for _ in range(8)]
    otr = O.Trainer(ocfg, P0, np.float64)
    fo, fg = iter(feeds), iter(feeds)
    for it in range(4):
        ro = otr.iteration(it, fo)
        rg = tr.iteration(it, fg)
        for k in ro:
            assert abs(float(rg[k]) - ro[k]) <= 1e-3 * max(1.0, abs(ro[k])), (it, k, float(rg[k]), ro[k])
    P = tr.get_params()
    skip = O.used_names(ocfg)
    for n, v in otr.P.items():
        if any(s in n for s in skip):
            continue
        d = np.abs(P[n].astype(np.float64) - v).max()
        assert d <= 2e-3 * max(1.0, np.abs(v).max()), (n, d)

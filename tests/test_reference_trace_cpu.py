"""The restatement (oracle/nets.py, objs.py, step.py, ssgan.py) replayed against what the REFERENCE'S OWN scripts did.

tests/golden/reference_trace.json and param_manifest.json were recorded in the build container by running the reference's
driver scripts and tflib code under the TF1 API shim (tests/golden/make_reference_trace.py, oracle/tf1_shim.py): parameter
names / shapes, optimizer hyper-parameters and var_lists, the order of the train loop's session.run calls, the minibatch and the
random nodes each of them consumed, and digests of costs, critic logits, every gradient and the final weights.  Weights,
minibatches and noise are functions of keys (oracle/reftrace.py), so the restatement is run here on the identical inputs in
float64 and has to reproduce every number.  A missing layer, a wrong loss composition, a wrong var_list or Adam setting, a
different step order all show up here; the arithmetic of the TF primitives themselves is shared by both sides and stays unpinned.
"""
import json
import os

import numpy as np
import pytest

from oracle import nets as N, step as S, objs as J, reftrace as RT, tape as tp

HERE = os.path.dirname(os.path.abspath(__file__))
TRACE = json.load(open(os.path.join(HERE, 'golden', 'reference_trace.json')))
MANIFEST = json.load(open(os.path.join(HERE, 'golden', 'param_manifest.json')))

IMG_KEYS = sorted(k for k in TRACE if not k.startswith('ssgan'))
SS_KEYS = sorted(k for k in TRACE if k.startswith('ssgan'))
CODE_MODES = ('vegan', 'vegan-wgan-gp', 'vegan-kl', 'vegan-ikl', 'vegan-jsd')


def image_cfg(key, consts):
    """oracle.nets.Cfg for (script, MODE) with the constants the trace / manifest ran with."""
    script, mode = key.split(':')
    ds = script.split('_')[-1]
    gm = script.startswith('gmgan')
    kw = dict(batch_size=consts['BATCH_SIZE'], n_coms=consts.get('N_COMS', 0) if gm else 0, dim=consts.get('DIM', consts.get('DIM_G')))
    if mode in CODE_MODES and ds != 'face':          # the scripts' own settings for the code-space objectives (gan_inference_cifar10.py:72-77)
        kw.update(dim_latent=8, bn=False)
    kw['latent_critic'] = mode in ('vegan', 'vegan-wgan-gp')
    kw['learn_std'] = mode in S.AGG_MODES
    kw['z_samples'] = consts.get('Z_SAMPLES', 100)
    kw['critic'] = mode not in ('vegan-mmd',) + S.AGG_MODES
    return N.Cfg(ds, script=script.rsplit('_', 1)[0], **kw), mode


def roles_for(cfg, mode, random_nodes):
    """node id -> (feed field, post-processing) from the creation order of the script's random nodes (see the scripts: the
    order in which tf.random_* calls appear on the path of each cost)."""
    B, dl = cfg.B, cfg.dim_latent
    want = {}
    if mode in S.AGG_MODES:
        want[('normal', (B, dl))] = ['q_eps', 'p_z_noise']
        Z = cfg.z_samples
        want[('categorical', (Z,))] = ['kl_k_idx']
        want[('normal', (Z, dl))] = {'vegan-kl': ['kl_eps'], 'vegan-ikl': ['kl_zp'], 'vegan-jsd': ['kl_eps', 'kl_zp']}[mode]
    elif mode in ('vegan', 'vegan-wgan-gp'):
        tags = ['r', 'f'] + (['h'] if mode == 'vegan-wgan-gp' else [])
        want[('normal', (B, dl))] = ['p_z_noise'] + ['dn_%s0' % t for t in tags]
        for i, w in ((1, 1024), (2, 512), (3, 256)):
            want[('normal', (B, w))] = ['dn_%s%d' % (t, i) for t in tags]
    else:
        want[('normal', (B, dl))] = ['p_z_noise']
    if mode in ('wali-gp', 'vegan-wgan-gp'):
        want[('uniform', (B, 1))] = ['alpha']
    if cfg.K:
        want[('uniform', (B, cfg.K))] = ['gumbel_u']
        want[('categorical', (B,))] = ['k_idx']
    if cfg.dataset == 'face':
        want[('uniform', (B, cfg.output_dim))] = ['dequant_u']
    roles, seen = {}, {}
    for nid, kind, shape in random_nodes:
        sig = (kind, tuple(shape))
        i = seen.get(sig, 0)
        seen[sig] = i + 1
        if sig in want and i < len(want[sig]):
            roles[nid] = (want[sig][i], kind, tuple(shape))
    missing = [(s, n) for s, names in want.items() for n in names if n not in [r[0] for r in roles.values()]]
    return roles, missing


def make_feed(cfg, mode, trace, run, roles):
    f = {}
    data = [x for x in run['feeds'] if x['stream']]
    assert len(data) == 1
    d = data[0]
    batch = RT.det_batch(d['stream'], d['index'], (d['spec'][0], tuple(d['spec'][1])))
    f['real_x' if d['spec'][0] == 'unit' else 'real_x_int'] = batch
    for nid, (name, kind, shape) in roles.items():
        classes = {'k_idx': cfg.K, 'kl_k_idx': cfg.B}.get(name)
        v = RT.det_noise(run['run'], nid, kind, shape, classes)
        if name == 'dequant_u':
            v = (v.astype(np.float64) / 128.0)
        if name == 'kl_k_idx':
            k = np.zeros((cfg.z_samples, cfg.B), np.float32)
            k[np.arange(cfg.z_samples), v] = 1
            name, v = 'kl_k', k
        f[name] = v
    if mode in S.AGG_MODES:                          # (the restatement reads every draw of the family; the unused ones are zeros)
        f.setdefault('kl_k', np.zeros((cfg.z_samples, cfg.B), np.float32))
        for n in ('kl_eps', 'kl_zp'):
            f.setdefault(n, np.zeros((cfg.z_samples, cfg.dim_latent), np.float32))
    return f, d


def close(a, b, rel=1e-9, abs_=1e-12):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= abs_ + rel * np.maximum(np.abs(a), np.abs(b)))


@pytest.mark.parametrize('key', sorted(MANIFEST))
def test_param_names_and_shapes_match_the_reference_scripts(key):
    """oracle init_params keys / shapes == the reference's lib.param calls at the scripts' own sizes; so are the optimizers'
    var_lists (trainable variables only: minimize drops the moving statistics, whose gradient is None)."""
    m = MANIFEST[key]
    script, mode = key.split(':')
    if script.startswith('ssgan'):
        from oracle import ssgan as OSS
        c = m['constants']
        ocfg = OSS.Cfg(batch_size=2, length=c['LEN'], n_c=c.get('N_C', 0), channels=3 if 'chairs' in script else 1,
                       op_dyn_mode='res_w' if 'chairs' in script else 'res', mode=mode)
        P = OSS.init_params(ocfg, 0)
    else:
        consts = dict(m['constants'], BATCH_SIZE=2)
        cfg, mode = image_cfg(key, consts)
        P = N.init_params(cfg, 0)
    ours = {k: list(v.shape) for k, v in P.items()}
    assert ours == m['params'], (sorted(set(ours) ^ set(m['params'])), [k for k in ours if k in m['params'] and ours[k] != m['params'][k]])


@pytest.mark.parametrize('key', IMG_KEYS)
def test_restatement_reproduces_the_reference_run(key):
    t = TRACE[key]
    cfg, mode = image_cfg(key, dict(t['constants'], **t.get('script_constants', {})))
    P0 = {n: RT.det_weight(n, shp) for n, shp in t['params'].items()}
    ours = N.init_params(cfg, 0)
    assert {k: list(v.shape) for k, v in ours.items()} == t['params']
    tr = S.Trainer(cfg, P0, mode, np.float64)
    # optimizers: kind, hyper-parameters, var_list, CRITIC_ITERS as the reference built them
    ref_opts = t['optimizers']
    for opt, ro in zip((tr.gen_opt, tr.disc_opt), ref_opts):
        assert sorted(opt.names) == sorted(n for n in ro['var_list'] if not n.endswith(('moving_mean', 'moving_variance'))), key
        if ro['kind'] == 'adam':
            assert isinstance(opt, J.Adam) and (opt.lr, opt.b1, opt.b2, opt.eps) == (ro['hp']['lr'], ro['hp']['beta1'], ro['hp']['beta2'], ro['hp']['eps'])
        else:
            assert isinstance(opt, J.RMSProp) and (opt.lr, opt.decay, opt.eps) == (ro['hp']['lr'], ro['hp']['decay'], ro['hp']['eps'])
    assert tr.critic_iters == t['critic_iters']
    roles, missing = roles_for(cfg, mode, t['random_nodes'])
    assert not missing, missing
    # the loop: iteration 0 runs the critic only; every session.run takes the next minibatch of the stream and fresh noise
    runs = [r for r in t['runs'] if r['train']]
    expect, it = [], 0
    while len(expect) < len(runs):
        if it > 0:
            expect.append(0)
        expect += [1] * tr.critic_iters
        it += 1
    if len(ref_opts) == 1:
        expect = [0] * len(runs)
    assert [r['train'][0]['optimizer'] for r in runs] == expect[:len(runs)]
    assert [[x for x in r['feeds'] if x['stream']][0]['index'] for r in runs] == list(range(len(runs)))
    for r in runs:
        feed, d = make_feed(cfg, mode, t, r, roles)
        assert close(RT.digest('feed%d' % d['placeholder'], feed.get('real_x', feed.get('real_x_int'))), d['digest'])
        # every random node the reference evaluated in this run has a role here (nothing it drew is ignored)
        assert set(x[0] for x in r['draws']) <= set(roles), (key, r['draws'])
        rec = r['train'][0]
        which = 'gen' if (rec['optimizer'] == 0) else 'disc'
        cost, grads, out = tr._run(feed, which)
        assert close(cost, rec['cost']), (key, r['run'], which, cost, rec['cost'])
        for name, dg in r['kept'].items():
            base, _, idx = name.partition('[')
            v = out.get(base)
            if v is None:
                continue
            if idx:
                v = v[int(idx[:-1])]
            assert close(RT.digest(name, v.v), dg, 1e-8), (key, r['run'], name)
        opt = tr.gen_opt if which == 'gen' else tr.disc_opt
        for n in opt.names:
            g = grads[n]
            if rec['grads'][n] is None:
                assert g is None or not np.any(g), (key, n)
                continue
            assert g is not None, (key, r['run'], n)
            assert close(RT.digest(n, g), rec['grads'][n], 1e-7, 1e-11), (key, r['run'], which, n, RT.digest(n, g)[:2], rec['grads'][n][:2])
        opt.apply(tr.P, grads)
    for n, dg in t['final'].items():
        assert close(RT.digest(n, tr.P[n], RT.FINAL_SAMPLES), dg, 1e-8, 1e-10), (key, 'final', n)


def ss_case(key):
    """(oracle cfg, constructor keywords, runs with a train op, feeds of those runs) of a state-space trace: the minibatch stream,
    the binarised labels (ssgan_inference_moving_mnist.py:81-85) and every random node's draw, by its role"""
    from oracle import ssgan as OSS
    t = TRACE[key]
    c = dict(t['script_constants'], **t['constants'])
    chairs = 'chairs' in key
    kw = dict(batch_size=c['BATCH_SIZE'], length=c['LEN'], dim=c['DIM'], dim_op=c['DIM_OP'], dim_g=c['DIM_LATENT_G'], dim_l=c['DIM_LATENT_L'],
              n_c=c.get('N_C', 0), channels=3 if chairs else 1, op_dyn_mode='res_w' if chairs else 'res', mode=c['MODE'])
    cfg = OSS.Cfg(**kw)
    B = cfg.B
    order = {('normal', (B, cfg.dim_l)): ['p_z_l_0', 'epsilon'], ('normal', (B, cfg.dim_g)): ['p_z_g'], ('categorical', (B,)): ['p_y_idx']}
    roles, seen = {}, {}
    for nid, kind, shape in t['random_nodes']:
        sig = (kind, tuple(shape))
        i = seen.get(sig, 0)
        seen[sig] = i + 1
        if sig in order and i < len(order[sig]):
            roles[nid] = (order[sig][i], kind, tuple(shape))
    runs = [r for r in t['runs'] if r['train']]
    feeds = []
    for j, r in enumerate(runs):
        d = [x for x in r['feeds'] if x['stream']][0]
        assert d['index'] == j
        x = RT.det_batch(d['stream'], d['index'], (d['spec'][0], tuple(d['spec'][1])))
        feed = {'real_x_unit': x.astype(np.float64)}
        y = np.zeros((B, cfg.n_c), np.float32)
        if cfg.n_c:
            y[np.arange(B), RT.det_batch(d['stream'] + '/y', d['index'], ('label', B, cfg.n_c))] = 1
        feed['real_y'] = y
        feed['p_y'] = np.zeros((B, cfg.n_c), np.float32)
        assert set(x[0] for x in r['draws']) <= set(roles), (key, r['draws'])
        for nid, (name, kind, shape) in roles.items():
            v = RT.det_noise(r['run'], nid, kind, shape, cfg.n_c or None)
            if name == 'p_y_idx':
                feed['p_y'][np.arange(B), v] = 1
            else:
                feed[name] = v
        feeds.append(feed)
    return cfg, kw, runs, feeds


@pytest.mark.parametrize('key', SS_KEYS)
def test_state_space_restatement_reproduces_the_reference_run(key):
    """ssgan_inference_moving_mnist.py / ssgan_inference_chairs.py (MODE local_ep) at reduced width: the per-time-step factors,
    the weighted objective, the two Adam instances and the loop, against oracle/ssgan.py."""
    from oracle import ssgan as OSS
    t = TRACE[key]
    cfg, _, runs, feeds = ss_case(key)
    assert {k: list(v.shape) for k, v in OSS.init_params(cfg, 0).items()} == t['params']
    tr = OSS.Trainer(cfg, {n: RT.det_weight(n, shp) for n, shp in t['params'].items()}, np.float64)
    for opt, ro in zip((tr.gen_opt, tr.disc_opt), t['optimizers']):
        assert sorted(opt.names) == sorted(ro['var_list']) and ro['kind'] == 'adam'
        assert (opt.lr, opt.b1, opt.b2, opt.eps) == (ro['hp']['lr'], ro['hp']['beta1'], ro['hp']['beta2'], ro['hp']['eps'])
    assert [r['train'][0]['optimizer'] for r in runs] == [1, 0, 1][:len(runs)]          # iteration 0: critic only
    for r, feed in zip(runs, feeds):
        rec = r['train'][0]
        which = 'gen' if rec['optimizer'] == 0 else 'disc'
        cost, grads, out = tr._run(feed, which)
        assert close(cost, rec['cost']), (key, r['run'], which, cost, rec['cost'])
        for name, dg in r['kept'].items():
            base, _, idx = name.partition('[')
            v = out[base][int(idx[:-1])] if idx else out[base]
            assert close(RT.digest(name, v.v), dg, 1e-8), (key, r['run'], name)
        opt = tr.gen_opt if which == 'gen' else tr.disc_opt
        for n in opt.names:
            assert close(RT.digest(n, grads[n]), rec['grads'][n], 1e-7, 1e-11), (key, r['run'], which, n)
        opt.apply(tr.P, grads)
    for n, dg in t['final'].items():
        assert close(RT.digest(n, tr.P[n], RT.FINAL_SAMPLES), dg, 1e-8, 1e-10), (key, 'final', n)

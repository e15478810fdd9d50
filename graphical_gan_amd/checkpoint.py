"""Checkpoint save AND restore keyed by the registry's parameter names -- the reference only ever saves
(`saver.save`, gmgan_inference_cifar10.py:465,548-549).  One .npz: every registry entry under its name (so the keys/shapes
are those of the reference's variables), plus the two Adam states (`adam/<role>/<param name>/m|v`, `adam/<role>/step`).

restore() may be called before the first training step: optimizers are created lazily by the first session.run, so the Adam
state is parked and picked up when the optimizer of that role is built."""
import numpy as np
import torch

from . import optim
from . import tflib as lib


def _roles():
    roles = {}
    for key, opt in optim._optimizers.items():
        if key[0] in roles:
            raise RuntimeError('two live optimizers for role %r: a checkpoint keys optimizer state by role '
                               '(optim.reset_optimizers() drops stale ones)' % (key[0],))
        roles[key[0]] = opt
    return roles


def _npz(path):
    """np.savez appends '.npz' to a path that lacks it; save and restore agree on the same file name"""
    path = str(path)
    return path if path.endswith('.npz') else path + '.npz'


def save(path, trainer=None, data_source=None):
    path = _npz(path)
    if trainer is not None:
        trainer.flush()
    torch.cuda.synchronize()
    out = {n: p.detach().cpu().numpy() for n, p in lib.named_params().items()}
    for role, opt in _roles().items():
        out['adam/%s/step' % role] = opt.step.cpu().numpy()
        for p, (o, n) in zip(opt.params, opt.slots):
            out['adam/%s/%s/m' % (role, p.param_name)] = opt.m[o:o + n].cpu().numpy().reshape(tuple(p.shape))
            out['adam/%s/%s/v' % (role, p.param_name)] = opt.v[o:o + n].cpu().numpy().reshape(tuple(p.shape))
    if data_source is not None:
        out['meta/data_source'] = np.asarray(str(data_source))
    np.savez(path, **out)
    return sorted(out)


def restore(path, trainer):
    """Loads parameters (creating registry entries as needed) and the Adam states."""
    z = np.load(_npz(path))
    params = {k: z[k] for k in z.files if not k.startswith(('adam/', 'meta/'))}
    trainer.load_params(params)
    state = {}
    for k in z.files:
        if k.startswith('adam/'):
            _, role, rest = k.split('/', 2)
            state.setdefault(role, {})[rest] = z[k]
    live = _roles()
    for role, st in state.items():
        if role in live:
            optim.load_adam_state(live[role], st)
        else:
            optim._pending_state[role] = st
    return sorted(params)

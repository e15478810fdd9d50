"""TF-flavoured Adam over flat parameter buffers + the data-parallel gradient exchange.

One AdamOptimizer owns one var_list (tflib/objs/gan_inference.py:108-117 creates two per script: generator
side and critic side).  On construction the parameters are re-homed into ONE flat fp32 buffer (each tensor
keeps its identity and shape; its storage becomes a 256-byte aligned view), so that
  * the update is a single HBM-bound kernel (28 B/param) instead of ~30 small ones,
  * the data-parallel exchange is ONE RCCL all-reduce per optimizer step over the flat gradient bucket
    (12.5 MB generator side / 16.25 MB critic side on CIFAR -- SURVEY.md 8e), issued right after backward.
The step counter lives in device memory so a captured HIP graph advances it on replay.
"""
import os as _os

import torch
import torch.distributed as dist

from . import functional as F

_ALIGN = 64  # floats (256 B): keeps every view float4-aligned for the MFMA filter loads and Adam


def layout_slots(sizes, align=_ALIGN):
    """[(offset, numel)] of tensors packed into one flat buffer, each start aligned; -> (slots, total)."""
    slots, off = [], 0
    for n in sizes:
        slots.append((off, int(n)))
        off += (int(n) + align - 1) // align * align
    return slots, off


# Issue-order log of a step's gradient exchange (tests, tools/dp_one_rank_check.py): when a list is installed here, the optimizer appends
# what it enqueues -- ('pack', lo, hi) gradient ranges written into the flat bucket (parameter indices), ('exchange', lo, hi, async) the
# all-reduce of flat[lo:hi] handed to the communicator, ('update',) the Adam launch -- and the engine adds ('backward', what) / ('wait',)
# marks.  What overlaps what is decided by this order (the exchange runs on the communicator's stream from the point it is issued).
EXCHANGE_LOG = [None]


def _xlog(*ev):
    if EXCHANGE_LOG[0] is not None:
        EXCHANGE_LOG[0].append(tuple(ev))


class GradBucket(object):
    """The data-parallel exchange step: ONE sum-all-reduce of a flat gradient buffer per optimizer step
    (SURVEY.md 8e).  Every replica holds the full model and an equal-size local minibatch, so
    sum/world == the gradient of the global-batch mean cost; `scale` (= 1/world) is folded into the Adam
    kernel's gradient read instead of a separate pass.  Works on any torch.distributed backend
    (RCCL on the GPUs; the CPU tests drive it with gloo)."""

    def __init__(self, flat, group=None):
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.scale = 1.0 / self.world

    def all_reduce(self, async_op=False, lo=0, hi=None):
        """async_op: returns the torch.distributed work handle (None when there is nothing to exchange); the caller
        calls .wait() before the bucket is read -- lets the exchange overlap kernels issued in between.
        lo/hi: exchange only flat[lo:hi] (a sub-bucket whose gradients are complete before the rest)."""
        if _os.environ.get('GGAN_SKIP_ALLREDUCE'):        # measurement only (bench.py: a step without its exchange; replicas drift apart)
            return None
        if self.world > 1 or (_os.environ.get('GGAN_FORCE_ALLREDUCE') and dist.is_available() and dist.is_initialized()):
            _xlog('exchange', int(lo), int(hi) if hi is not None else int(self.flat.numel()), bool(async_op))
            buf = self.flat if (lo == 0 and hi is None) else self.flat[lo:hi]
            from . import rccl
            comm = rccl.get(create=False) if buf.is_cuda else None
            if comm is not None:
                # RCCL bound directly: an enqueue on the communicator's own stream (a parallel branch of the step graph while one is
                # being captured), no process-group watchdog behind it (rccl.py)
                return comm.all_reduce_(buf, async_op=async_op)
            if buf.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError('a torch.distributed collective cannot be captured into a step graph (engine.Trainer only captures '
                                   'exchanges issued through rccl.Communicator)')
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            return work if async_op else None
        return None


class AdamOptimizer(object):
    def __init__(self, params, lr=2e-4, beta1=0.5, beta2=0.999, eps=1e-8):
        params = [p for p in params if p.requires_grad]       # TF minimize drops (None, var) pairs
        seen, uniq = set(), []
        for p in params:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.params = uniq
        self.lr, self.beta1, self.beta2, self.eps = float(lr), float(beta1), float(beta2), float(eps)
        if not self.params:
            raise ValueError('No variables to optimize.')
        dev = self.params[0].device
        for p in self.params:
            if getattr(p, '_flat_owner', None) is not None:
                raise NotImplementedError('parameter %s already belongs to another optimizer' %
                                          getattr(p, 'param_name', '?'))
        self.slots, off = layout_slots([p.numel() for p in self.params])
        self.total = off
        self.theta = torch.zeros(off, dtype=torch.float32, device=dev)
        self.m = torch.zeros_like(self.theta)
        self.v = torch.zeros_like(self.theta)
        self.g = torch.zeros_like(self.theta)
        self.step = torch.zeros(1, dtype=torch.int32, device=dev)      # advanced by the pack kernel of each step
        with torch.no_grad():
            for p, (o, n) in zip(self.params, self.slots):
                view = self.theta[o:o + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p._flat_owner = self
        self.bucket = GradBucket(self.g)
        self._one = None
        self._arrive, self._updated = None, False
        self.world = self.bucket.world

    # -- one optimizer step ---------------------------------------------------------------------------
    def compute_gradients(self, cost):
        """-> one entry per parameter: its gradient, or a pair (gradient, gradient through the parameter's second leaf) when a
        second pass of the step used tflib.second_leaf (summed by pack)"""
        from . import tflib as lib
        if self._one is None or self._one.shape != cost.shape:
            self._one = F.unit_seed(cost)            # persistent d(cost)/d(cost) seed (no fill launch per step)
        extra = [lib.second_leaf_for(p) for p in self.params]
        idx = [i for i, e in enumerate(extra) if e is not None]
        if not idx:
            return torch.autograd.grad(cost, self.params, grad_outputs=self._one, allow_unused=True)
        g = torch.autograd.grad(cost, list(self.params) + [extra[i] for i in idx], grad_outputs=self._one, allow_unused=True)
        out = list(g[:len(self.params)])
        for j, i in enumerate(idx):
            g2 = g[len(self.params) + j]
            if g2 is not None:
                out[i] = (out[i], g2)
        return out

    def can_fuse_update(self):
        """the update may ride in the pack launch: a single replica (nothing happens to the packed gradient before update()),
        plain Adam, one pack launch"""
        return (type(self) is AdamOptimizer and self.world == 1 and len(self.params) <= F._lib.PACK_MAX
                and not _os.environ.get('GGAN_FORCE_ALLREDUCE') and not _os.environ.get('GGAN_NO_PACK_ADAM'))

    def pack(self, grads, fuse_update=False):
        """fuse_update: the caller goes on to all_reduce() (a no-op on one replica) and update(); where can_fuse_update() allows
        it, the update is applied by the pack launch and the following update() call does nothing"""
        cc = lambda g: None if g is None else (g if g.is_contiguous() else g.contiguous())
        gs = [(cc(g[0]), cc(g[1])) if isinstance(g, tuple) else cc(g) for g in grads]
        _xlog('pack', 0, len(self.params))
        self._updated = False            # (a fused pack whose update() call never came must not swallow the next one)
        if fuse_update and self.can_fuse_update():
            if self._arrive is None:
                self._arrive = torch.zeros(F._lib.PACK_ARRIVE_INTS, dtype=torch.int32, device=self.theta.device)
            F.pack_(gs, self.slots, self.g, adam=(self.theta, self.m, self.v, self.step, self._arrive, self.lr, self.beta1, self.beta2,
                                                  self.eps, self.bucket.scale))
            self._updated = True
            return gs
        F.pack_(gs, self.slots, self.g, bump=self.step)     # also advances the step counter (read by update())
        return gs  # keep alive until the kernel ran (stream-ordered)

    def all_reduce(self, async_op=False, lo=0, hi=None):
        """Sum the flat gradient bucket (or its slice [lo, hi)) over the data-parallel replicas (RCCL over xGMI)."""
        return self.bucket.all_reduce(async_op, lo, hi)

    def split_at(self, predicate):
        """k such that params[:k] all satisfy predicate and params[k:] do not (None if the parameters are not ordered that
        way), with the flat-buffer offset where params[k] starts -- the boundary of a two-bucket gradient exchange."""
        flags = [bool(predicate(p)) for p in self.params]
        k = sum(flags)
        if k == 0 or k == len(flags) or not all(flags[:k]) or any(flags[k:]):
            return None
        return k, self.slots[k][0]

    def pack_subset(self, grads, lo, hi, bump):
        _xlog('pack', int(lo), int(hi))
        self._updated = False
        gs = [None if g is None else (g if g.is_contiguous() else g.contiguous()) for g in grads]
        F.pack_(gs, self.slots[lo:hi], self.g, bump=self.step if bump else None)
        return gs

    def update(self):
        if self._updated:            # applied by the pack launch (pack(fuse_update=True))
            self._updated = False
            _xlog('update', 'in the pack launch')
            return
        _xlog('update')
        F.adam_step_(self.theta, self.g, self.m, self.v, self.step, self.lr, self.beta1, self.beta2, self.eps,
                     self.bucket.scale, counted=True)

    def apply_gradients(self, grads):
        keep = self.pack(grads, fuse_update=True)
        self.all_reduce()
        self.update()
        return keep

    def minimize(self, cost):
        return self.apply_gradients(self.compute_gradients(cost))

    def state_dict(self):
        return dict(m=self.m.clone(), v=self.v.clone(), step=self.step.clone())


class RMSPropOptimizer(AdamOptimizer):
    """tf.train.RMSPropOptimizer(learning_rate) with TF's defaults (decay .9, momentum 0, epsilon 1e-10, ms starts at 1) over
    the same flat buffers; `clip`: the wali objective's weight clipping folded into the update kernel."""

    def __init__(self, params, lr=5e-5, decay=0.9, eps=1e-10, clip=None):
        AdamOptimizer.__init__(self, params, lr=lr)
        self.decay, self.eps, self.clip = float(decay), float(eps), clip
        self.m.fill_(1.0)                    # `ms` slot (TF initialises it with ones); self.v is unused

    def update(self):
        F.rmsprop_step_(self.theta, self.g, self.m, self.lr, self.decay, self.eps, self.bucket.scale, self.clip)


class TrainOp(object):
    """What the reference's `*_train_op` is: running it applies one Adam step for `cost`."""

    def __init__(self, optimizer, cost):
        self.optimizer, self.cost = optimizer, cost

    def run(self):
        self.optimizer.minimize(self.cost)

    __call__ = run


_optimizers = {}


def get_optimizer(role, params, **hp):
    """The reference builds its optimizers once at graph-construction time; the eager counterpart calls the
    objective every step, so optimizers are cached by (role, parameter identity, hyper-parameters)."""
    key = (role, tuple(id(p) for p in params if p.requires_grad), tuple(sorted(hp.items())))
    opt = _optimizers.get(key)
    if opt is None:
        kind = hp.pop('kind', 'adam')
        opt = RMSPropOptimizer(params, **hp) if kind == 'rmsprop' else AdamOptimizer(params, **hp)
        _optimizers[key] = opt
        if role in _pending_state:                      # a checkpoint restored before this optimizer existed
            load_adam_state(opt, _pending_state.pop(role))
    return opt


_pending_state = {}


STATE_EPOCH = [0]       # bumped whenever an optimizer's step counter is overwritten from outside (engine.Trainer re-anchors its ring feed)


def load_adam_state(opt, state):
    """state: {'step': int32[1], '<param name>/m': array, '<param name>/v': array} (checkpoint.py)"""
    STATE_EPOCH[0] += 1
    with torch.no_grad():
        opt.step.copy_(torch.as_tensor(state['step']).to(opt.step.device).reshape(opt.step.shape))
        for p, (o, n) in zip(opt.params, opt.slots):
            for which, buf in (('m', opt.m), ('v', opt.v)):
                a = state.get('%s/%s' % (p.param_name, which))
                if a is not None:
                    buf[o:o + n].copy_(torch.as_tensor(a, dtype=torch.float32).reshape(-1))


def reset_optimizers(keep_params=True):
    """Drop every optimizer (their flat theta / m / v / g buffers go with them).  keep_params: the parameters move back into
    storage of their own; False when the registry is being emptied anyway (tflib.delete_all_params)."""
    for opt in _optimizers.values():
        for p in opt.params:
            if keep_params:
                p.data = p.data.clone()
            p._flat_owner = None
    _optimizers.clear()
    _pending_state.clear()
    F.UNIT_SEEDS.clear()

"""optimizer and noise kernels: Adam / RMSProp steps, packed updates, device noise."""
import ctypes as C
import torch
from .. import _lib
from .._lib import check  # noqa: F401
from ._core import _L, _p, _stream, _DEFER  # noqa: F401


# ---------------------------------------------------------------------------------------------------
# optimiser primitives (no autograd)
# ---------------------------------------------------------------------------------------------------
def adam_step_(theta, g, m, v, step, lr, beta1, beta2, eps=1e-8, grad_scale=1.0, counted=False):
    """counted=True: `step` was already advanced to this update's ordinal (pack_(..., bump=step))."""
    n = theta.numel()
    assert g.numel() == n and m.numel() == n and v.numel() == n and step.dtype == torch.int32
    L = _L()
    if counted:
        check(L.ggan_adam_step_counted(_p(theta), _p(g), _p(m), _p(v), n, _p(step), lr, beta1, beta2, eps, grad_scale,
                                       _stream()), 'ggan_adam_step_counted')
        return
    check(L.ggan_adam_step(_p(theta), _p(g), _p(m), _p(v), n, _p(step), lr, beta1, beta2, eps, grad_scale, _stream()),
          'ggan_adam_step')
    check(L.ggan_adam_advance(_p(step), _stream()), 'ggan_adam_advance')


def rmsprop_step_(theta, g, ms, lr, decay=0.9, eps=1e-10, grad_scale=1.0, clip=None):
    lo, hi = (float('-inf'), float('inf')) if clip is None else (float(clip[0]), float(clip[1]))
    check(_L().ggan_rmsprop_step(_p(theta), _p(g), _p(ms), theta.numel(), lr, decay, eps, grad_scale, lo, hi, _stream()),
          'ggan_rmsprop_step')


NOISE_NORMAL, NOISE_UNIFORM, NOISE_ONEHOT = 0, 1, 2


def noise_state(device, seed=None):
    """{seed, draw number, arrival counter} of ggan_noise_fill as an int64[3] device tensor (seed: torch.initial_seed())"""
    seed = torch.initial_seed() if seed is None else seed
    return torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF, 0, 0], dtype=torch.int64, device=device)


def noise_fill_(state, specs):
    """One launch for all the noise of a step.  specs: list of (tensor, kind, a, b) -- NOISE_NORMAL: a + b*N(0,1); NOISE_UNIFORM:
    [a, b); NOISE_ONEHOT: rows of the 2-D tensor become one-hot with a uniformly drawn index.  In place; graph-capturable (the
    draw number advances on the device)."""
    assert state.dtype == torch.int64 and state.numel() == 3 and state.is_cuda
    n = len(specs)
    for t, _, _, _ in specs:
        assert t.is_contiguous() and t.dtype == torch.float32 and t.device == state.device
    dsts = (C.c_void_p * n)(*[t.data_ptr() for t, _, _, _ in specs])
    sizes = (C.c_size_t * n)(*[t.numel() for t, _, _, _ in specs])
    kinds = (C.c_int * n)(*[int(k) for _, k, _, _ in specs])
    a = (C.c_float * n)(*[float(x) for _, _, x, _ in specs])
    b = (C.c_float * n)(*[float(x) for _, _, _, x in specs])
    widths = (C.c_int * n)(*[int(t.shape[-1]) if k == NOISE_ONEHOT else 0 for t, k, _, _ in specs])
    check(_L().ggan_noise_fill(dsts, sizes, kinds, a, b, widths, n, _p(state), _stream()), 'ggan_noise_fill')


def pack_(tensors, offsets, flat, bump=None, adam=None):
    """flat[offsets[i] : offsets[i]+n_i] = tensors[i] (None -> zeros); tensors registered by `defer_wgrad_reduce` are
    summed over their split-K slabs on the way.  An entry may be a pair (t, t2): two gradient contributions of one parameter
    (either may be None), summed here.  bump: int32 device counter incremented once (the Adam step ordinal).
    adam = (theta, m, v, step, arrive, lr, beta1, beta2, eps, grad_scale): the Adam update rides in the same launch
    (ggan_pack_adam; at most PACK_MAX tensors; `step` takes the place of bump)."""
    L = _L()
    reg = _DEFER[0]
    assert adam is None or len(tensors) <= _lib.PACK_MAX
    for i0 in range(0, len(tensors), _lib.PACK_MAX):
        chunk = [(t if isinstance(t, tuple) else (t, None)) for t in tensors[i0:i0 + _lib.PACK_MAX]]
        chunk = [((b, None) if a is None else (a, b)) for a, b in chunk]        # a lone second contribution is the first
        n = len(chunk)
        sizes = (C.c_size_t * n)(*[int(s) for s in [o[1] for o in offsets[i0:i0 + n]]])
        offs = (C.c_size_t * n)(*[int(o[0]) for o in offsets[i0:i0 + n]])

        def table(col):
            ts = [c[col] for c in chunk]
            srcs = (C.c_void_p * n)(*[t.data_ptr() if t is not None else 0 for t in ts])
            info = [reg.pop(t.data_ptr(), None) if (reg and t is not None) else None for t in ts]
            parts = (C.c_int * n)(*[(e[0] if e else 1) for e in info])
            strides = (C.c_size_t * n)(*[(e[1] if e else 0) for e in info])
            return srcs, parts, strides
        s1, p1, st1 = table(0)
        if adam is not None:
            s2, p2, st2 = table(1) if any(c[1] is not None for c in chunk) else (None, None, None)
            theta, m, v, step, arrive, lr, b1, b2, eps, gscale = adam
            check(L.ggan_pack_adam(s1, sizes, offs, p1, st1, s2, p2, st2, n, _p(flat), _p(theta), _p(m), _p(v), _p(step), _p(arrive),
                                   lr, b1, b2, eps, gscale, _stream()), 'ggan_pack_adam')
        elif any(c[1] is not None for c in chunk):
            s2, p2, st2 = table(1)
            check(L.ggan_pack_parts2(s1, sizes, offs, p1, st1, s2, p2, st2, n, _p(flat), _p(bump if i0 == 0 else None), _stream()),
                  'ggan_pack_parts2')
        else:
            check(L.ggan_pack_parts(s1, sizes, offs, p1, st1, n, _p(flat), _p(bump if i0 == 0 else None), _stream()),
                  'ggan_pack_parts')

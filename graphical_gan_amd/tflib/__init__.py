"""MI355X counterpart of the reference's `tflib` package (tflib/__init__.py): the name-keyed parameter
registry that gives the driver scripts weight sharing, with torch device tensors in place of tf.Variable.

Same functions, argument order and behaviour as tflib/__init__.py:9-47,100-120:
  param(name, *args, **kwargs)   first call creates the parameter from the numpy initial value, every later
                                 call with the same name returns the SAME object (that is how Generator is
                                 instantiated several times with shared weights);
  params_with_name(substr)       substring match over the registry keys (builds the optimiser var lists);
  delete_all_params / alias_params / delete_param_aliases / print_model_settings*.
"""
import numpy as np
import torch

_params = {}
_param_aliases = {}
_device = [None]
# activation taps (off unless a list is put here): every layer whose epilogue is a ReLU / LeakyReLU on an image-shaped tensor appends
# (layer name, output) in call order -- tests read the sign pattern of the units nearest their kink (tests/test_golden_full_gpu.py)
TAPS = [None]


def tap(name, out):
    if TAPS[0] is not None and out.dim() == 4:
        from ..functional import _SITE
        TAPS[0].append((_SITE['scope'], name, out.detach()))       # (scope: the step being built -- 'gen0', 'disc3', ... -- or None)
    return out


def drop_taps(prefixes):
    """forget the taps of steps about to be built again (engine.Trainer: the eager rehearsal's tensors in front of the capture's)"""
    if TAPS[0] is not None:
        TAPS[0][:] = [t for t in TAPS[0] if not (t[0] or '').startswith(tuple(prefixes))]


def set_device(device):
    """Device new parameters are created on (default: the current HIP device).  The registry itself is host
    logic and also works with 'cpu' tensors (used by the CPU test-suite); every op refuses CPU tensors."""
    _device[0] = torch.device(device)


def get_device():
    if _device[0] is None:
        if not torch.cuda.is_available():
            raise RuntimeError('graphical_gan_amd.tflib needs a HIP device (no CPU path); none is visible')
        _device[0] = torch.device('cuda', torch.cuda.current_device())
    return _device[0]


def param(name, *args, **kwargs):
    """tflib/__init__.py:9-33.  args[0] is the initial value (numpy array); kwargs: trainable=False."""
    if name not in _params:
        value = args[0] if args else kwargs['initial_value']
        trainable = kwargs.get('trainable', True)
        t = torch.as_tensor(np.asarray(value), dtype=torch.float32).to(get_device()).contiguous()
        p = torch.nn.Parameter(t, requires_grad=bool(trainable))
        p.param = True
        p.param_name = name
        _params[name] = p
    result = _params[name]
    while id(result) in _param_aliases:
        result = _param_aliases[id(result)]
    if _frozen and any(f in name for f in _frozen):
        return result.detach()
    if _second[0] is not None and result.requires_grad:
        return _second_leaf_of(name, result)
    return result


# A parameter that two passes of ONE step reach (the critic's weights: its main pass and its gradient-penalty pass) would get
# two gradient contributions that the tape adds with one launch per parameter.  Inside `with second_leaf():` the ops receive a
# second autograd leaf on the SAME storage instead; the optimizer asks the tape for both gradients and sums them where it packs
# the bucket (optim.AdamOptimizer.compute_gradients / functional.pack_).  Host logic only: values are shared, nothing is copied.
_second = [None]
_second_leaves = {}


class second_leaf(object):
    def __enter__(self):
        self.prev, _second[0] = _second[0], True
        return self

    def __exit__(self, *a):
        _second[0] = self.prev


_second_count = [0]


def second_leaf_count():
    """how many times a second leaf has been handed out so far (engine.Trainer: did THIS step use any?)"""
    return _second_count[0]


def _second_leaf_of(name, p):
    _second_count[0] += 1
    t = _second_leaves.get(name)
    if t is None or t.data_ptr() != p.data_ptr() or t.shape != p.shape:       # (re-homed into an optimizer's flat buffer since)
        t = p.detach().requires_grad_(True)
        t.param_name = name
        _second_leaves[name] = t
    return t


def second_leaf_for(p):
    """the second leaf last handed out for registry parameter p, if any (a leaf the current tape does not contain simply gets no
    gradient; one created before an optimizer re-homed p is still the leaf the tape of THAT forward pass holds)"""
    return _second_leaves.get(getattr(p, 'param_name', None))


_frozen = []


class frozen(object):
    """with frozen('Discriminator'): ...   parameters whose name contains the substring are handed to the ops without
    a gradient edge.  Eager counterpart of the reference's `var_list=` (tflib/objs/gan_inference.py:108-117): TF prunes
    the gradients of variables a train op does not own; an eager tape must be told before the forward pass, otherwise
    it computes (and throws away) the critic's weight gradients during every generator step."""

    def __init__(self, *substrings):
        self.subs = list(substrings)

    def __enter__(self):
        _frozen.extend(self.subs)
        return self

    def __exit__(self, *a):
        for s_ in self.subs:
            _frozen.remove(s_)


def params_with_name(name):
    return [p for n, p in _params.items() if name in n]


def named_params():
    return dict(_params)


def delete_all_params():
    """tflib/__init__.py:44-45.  The optimizers built over these parameters own flat device buffers (theta, m, v, g) and mark
    the parameters as theirs: registry and optimizer lifetimes stay in sync."""
    from .. import optim
    optim.reset_optimizers(keep_params=False)
    _params.clear()
    _second_leaves.clear()
    _build_phase[0] = True


# The reference draws every initial value at GRAPH-BUILD time, on every layer call (even when the parameter exists:
# conv2d.py:75-88) -- that is how its scripts consume numpy RNG state.  An eager framework calls the layer functions again on
# every step, where TensorFlow only replays the built graph and draws nothing.  A Trainer marks that boundary: once a generator
# and a critic step have been built, layer calls that find their parameters skip the draw (a 4608x512 uniform draw per
# Linear call otherwise: 45 of the 48 ms of an eager iteration).  Until then, and for any direct use of the ops, every call
# draws, as in the reference.
_build_phase = [True]


def end_build_phase():
    _build_phase[0] = False


def initial_values_needed(*names):
    """False only after end_build_phase() for parameters that all exist already"""
    return _build_phase[0] or any(n not in _params for n in names)


def alias_params(replace_dict):
    for old, new in replace_dict.items():
        _param_aliases[id(old)] = new


def delete_param_aliases():
    _param_aliases.clear()


def _settings(locals_):
    all_vars = [(k, v) for (k, v) in locals_.items()
                if (k.isupper() and k != 'T' and k != 'SETTINGS' and k != 'ALL_SETTINGS')]
    return sorted(all_vars, key=lambda x: x[0])


def print_model_settings(locals_):
    print("Uppercase local vars:")
    for var_name, var_value in _settings(locals_):
        print("\t{}: {}".format(var_name, var_value))


def print_model_settings_to_file(locals_, logfile):
    print("Uppercase local vars:")
    for var_name, var_value in _settings(locals_):
        print("\t{}: {}".format(var_name, var_value))
        with open(logfile, 'a') as f:
            f.write("\t{}: {}".format(var_name, var_value))


def print_model_settings_dict(settings):
    print("Settings dict:")
    for var_name, var_value in sorted(settings.items(), key=lambda x: x[0]):
        print("\t{}: {}".format(var_name, var_value))


from . import ops, objs, plot, utils  # noqa: E402,F401
from . import mnist, cifar10, svhn, celebA, simple_moving_mnist, chairs, save_images  # noqa: E402,F401

#!/usr/bin/env python
"""Run the reference's OWN driver scripts and tflib code (from /root/reference, BUILD CONTAINER ONLY) under the TF1 API shim
(oracle/tf1_shim.py) and commit what they do as small fixtures:

  tests/golden/param_manifest.json       (script, MODE) -> {lib.param name: shape} at the scripts' own sizes, plus each
                                         optimizer's kind / hyper-parameters / var_list and CRITIC_ITERS
  tests/golden/reference_trace.json      for every (script, MODE) at reduced width (DIM etc. patched, see CASES): the order of
                                         session.run calls of the reference's train loop, which minibatch each was fed, which
                                         random nodes it drew, the fetched cost, critic logits, and per-parameter gradient
                                         digests, and the parameter digests after the last run (oracle/reftrace.py)

Nothing of the reference is copied: its Python 2 sources are read where they lie, converted in memory (lib2to3 + two AST
rewrites for `/` on ints and `is` on string literals), executed, and only numbers / names leave.  What this pins and what it
does not: oracle/tf1_shim.py's docstring.

  python tests/golden/make_reference_trace.py [--only gan_inference_mnist:ali] [--manifest-only]
"""
import argparse
import ast
import importlib.abc
import importlib.util
import io
import json
import os
import re
import sys
import tempfile
import time
import types
from unittest import mock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'

from oracle import tf1_shim as shim          # noqa: E402
from oracle import reftrace as RT            # noqa: E402


# ---------------------------------------------------------------------------------------------------------------
# Python 2 -> 3 in memory
# ---------------------------------------------------------------------------------------------------------------
def _py2div(a, b):
    ints = (int, np.integer)
    if isinstance(a, ints) and isinstance(b, ints) and not isinstance(a, bool) and not isinstance(b, bool):
        return a // b
    return a / b


class _Py2Semantics(ast.NodeTransformer):
    def visit_BinOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            return ast.copy_location(ast.Call(func=ast.Name(id='_py2div', ctx=ast.Load()), args=[node.left, node.right], keywords=[]), node)
        return node

    def visit_AugAssign(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            load = ast.parse(ast.unparse(node.target), mode='eval').body
            return ast.copy_location(ast.Assign(targets=[node.target], value=ast.Call(func=ast.Name(id='_py2div', ctx=ast.Load()),
                                                                                         args=[load, node.value], keywords=[])), node)
        return node

    def visit_Compare(self, node):
        self.generic_visit(node)
        # `MODE is 'vae'`: identity of interned identifier-like literals == equality in CPython 2
        if any(isinstance(c, ast.Constant) and isinstance(c.value, str) for c in node.comparators):
            node.ops = [ast.Eq() if isinstance(o, ast.Is) else ast.NotEq() if isinstance(o, ast.IsNot) else o for o in node.ops]
        return node


_TOOL = None


def to_py3(src, filename):
    global _TOOL
    if _TOOL is None:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            from lib2to3 import refactor
        _TOOL = refactor.RefactoringTool(refactor.get_fixers_from_package('lib2to3.fixes'))
    if not src.endswith('\n'):
        src += '\n'
    src3 = str(_TOOL.refactor_string(src, filename))
    tree = _Py2Semantics().visit(ast.parse(src3, filename))
    ast.fix_missing_locations(tree)
    return compile(tree, filename, 'exec')


class _RefTflib(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """import tflib[.ops.x | .objs.x | .utils.x] -> the reference's file, converted; everything else of tflib (loaders, plotting,
    image dumps, Inception score, t-SNE scatter) is outside the hot path and stubbed."""
    REAL = ('tflib', 'tflib.ops', 'tflib.objs', 'tflib.utils')

    def find_spec(self, name, path=None, target=None):
        if name != 'tflib' and not name.startswith('tflib.'):
            return None
        rel = name.split('.')
        base = os.path.join(REF, *rel)
        if name in self.REAL or name.rsplit('.', 1)[0] in self.REAL[1:]:
            if os.path.isdir(base):
                return importlib.util.spec_from_loader(name, self, origin=os.path.join(base, '__init__.py'), is_package=True)
            if os.path.exists(base + '.py'):
                return importlib.util.spec_from_loader(name, self, origin=base + '.py')
        return importlib.util.spec_from_loader(name, self, origin='stub')

    def create_module(self, spec):
        return None

    def exec_module(self, module):
        origin = module.__spec__.origin
        if origin == 'stub':
            STUBS.fill(module)
            return
        module.__file__ = origin
        if module.__spec__.submodule_search_locations is not None:
            module.__path__ = [os.path.dirname(origin)]
        module.__dict__['_py2div'] = _py2div
        src = open(origin).read() if os.path.getsize(origin) else ''
        exec(to_py3(src, origin), module.__dict__)


class Stubs(object):
    """tflib.{mnist, cifar10, svhn, celebA, chairs, simple_moving_mnist}: deterministic synthetic loaders (oracle/reftrace.py);
    tflib.{plot, save_images, inception_score, visualization}: no-ops."""

    def __init__(self):
        self.fed = {}                       # id(array) -> (stream, index)
        self.specs = {}

    EPOCH = 4                               # minibatches per epoch (the scripts' logging code walks whole dev / test epochs)

    def _gen(self, stream, spec, with_labels):
        state = {'i': 0}

        def get_epoch():
            for _ in range(self.EPOCH):
                i = state['i']
                state['i'] += 1
                a = RT.det_batch(stream, i, spec)
                self.fed[id(a)] = (stream, i, a)
                self.specs[stream] = spec
                if with_labels:
                    yield a, RT.det_batch(stream + '/y', i, ('label', spec[1][0], with_labels))
                else:
                    yield a
        return get_epoch

    def fill(self, module):
        name = module.__name__.split('.', 1)[1]
        S = self
        if name == 'mnist':
            module.load = lambda bs, tbs, n_labelled=None: tuple(S._gen('mnist/' + k, ('unit', (bs, 784)), 10) for k in ('train', 'dev', 'test'))
        elif name in ('cifar10', 'svhn'):
            module.load = lambda bs, data_dir=None: tuple(S._gen(name + '/' + k, ('int', (bs, 3072)), 10) for k in ('train', 'dev'))
            module.get_reconstruction_data = lambda bs, data_dir=None: RT.det_batch(name + '/rec', 0, ('int', (bs, 3072)))
        elif name == 'celebA':
            module.load = lambda bs, data_dir=None, num_dev=5000: tuple(S._gen('celebA/' + k, ('int', (bs, 3 * 64 * 64)), 0) for k in ('train', 'dev'))
        elif name == 'chairs':
            module.load = lambda seq_length, batch_size, size, data_dir=None, num_dev=200: tuple(
                S._gen('chairs/' + k, ('int', (batch_size, seq_length, 3 * size * size)), 0) for k in ('train', 'dev'))
        elif name == 'simple_moving_mnist':
            module.load_video = lambda seq_length, batch_size, cla=None: tuple(
                S._gen('moving_mnist/' + k, ('unit', (batch_size, seq_length, 64 * 64)), 10) for k in ('train', 'dev'))
        else:                               # plot / save_images / inception_score / visualization
            module.__getattr__ = lambda attr: (lambda *a, **k: None)


STUBS = Stubs()


def install():
    sys.meta_path.insert(0, _RefTflib())
    sys.modules['tensorflow'] = shim
    for m in ('matplotlib', 'matplotlib.pyplot', 'sklearn', 'sklearn.datasets', 'sklearn.manifold', 'scipy.misc'):
        sys.modules[m] = mock.MagicMock(name=m)


def purge_tflib():
    for k in [k for k in sys.modules if k == 'tflib' or k.startswith('tflib.')]:
        del sys.modules[k]


# ---------------------------------------------------------------------------------------------------------------
# cases: (script, MODE, extra source patches for the reduced-width trace)
# ---------------------------------------------------------------------------------------------------------------
SMALL_IMG = {'DIM': 8, 'BATCH_SIZE': 6, 'ITERS': 3}
SMALL_FACE = {'DIM_G': 4, 'DIM_D': 4, 'BATCH_SIZE': 6, 'ITERS': 3}
CASES = [
    ('gan_inference_mnist', 'ali', SMALL_IMG), ('gan_inference_mnist', 'wali-gp', dict(SMALL_IMG, ITERS=2)),
    ('gan_inference_mnist', 'alice', SMALL_IMG), ('gan_inference_mnist', 'vegan', dict(SMALL_IMG, ITERS=2)),
    ('gan_inference_cifar10', 'ali', SMALL_IMG), ('gan_inference_cifar10', 'wali-gp', dict(SMALL_IMG, ITERS=2)),
    ('gan_inference_cifar10', 'alice-z', SMALL_IMG), ('gan_inference_cifar10', 'wali', dict(SMALL_IMG, ITERS=2)),
    ('gan_inference_cifar10', 'vegan-wgan-gp', dict(SMALL_IMG, ITERS=2)), ('gan_inference_cifar10', 'vegan-mmd', SMALL_IMG),
    ('gan_inference_cifar10', 'vegan-kl', dict(SMALL_IMG, Z_SAMPLES=5)), ('gan_inference_cifar10', 'vegan-jsd', dict(SMALL_IMG, Z_SAMPLES=5)),
    ('gan_inference_svhn', 'ali', SMALL_IMG),
    ('gan_inference_face', 'ali', SMALL_FACE),
    ('gmgan_inference_mnist', 'local_ep', dict(SMALL_IMG, N_COMS=5)), ('gmgan_inference_cifar10', 'local_ep', dict(SMALL_IMG, N_COMS=5)),
    ('gmgan_inference_cifar10', 'local_epce', dict(SMALL_IMG, N_COMS=5)),
    ('gmgan_inference_svhn', 'local_ep', dict(SMALL_IMG, N_COMS=5)), ('gmgan_inference_face', 'local_ep', dict(SMALL_FACE, N_COMS=5)),
    ('ssgan_inference_moving_mnist', 'local_ep', dict(DIM=4, DIM_OP=16, BATCH_SIZE=10, LEN=4, ITERS=2)),
    ('ssgan_inference_chairs', 'local_ep', dict(DIM=4, DIM_OP=16, BATCH_SIZE=2, LEN=3, ITERS=2)),
]


class StopBuild(Exception):
    pass


def patch_source(src, consts):
    """Rewrite the `NAME = <literal>` assignments of the script's hyper-parameter block."""
    for k, v in consts.items():
        pat = re.compile(r'^([ \t]*)%s[ \t]*=[ \t]*[^#\n]*' % re.escape(k), re.M)
        if not pat.search(src):
            raise KeyError('%s is never assigned' % k)
        src = pat.sub(lambda m: '%s%s = %r ' % (m.group(1), k, v), src)
    return src


def run_script(script, consts, on_session):
    """Execute the reference script with patched constants; on_session(session, globals) is called when the script opens its
    tf.Session (the graph is complete then).  -> (globals, session | None)"""
    purge_tflib()
    g = shim.reset(0)
    path = os.path.join(REF, script + '.py')
    src = patch_source(open(path).read(), consts)
    glob = {'__name__': '__main__', '__file__': path, '_py2div': _py2div}
    sess_box = []

    def hook(sess):
        sess_box.append(sess)
        on_session(sess, glob)
    shim.Session.on_create = staticmethod(hook)
    np.random.seed(0)
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix='reftrace_')
    os.chdir(tmp)                              # the scripts mkdir result/<...> and copy themselves there
    try:
        with mock.patch('shutil.copy', lambda *a, **k: None), mock.patch('sys.stdout', io.StringIO()):
            try:
                exec(to_py3(src, path), glob)
            except StopBuild:
                pass
    finally:
        os.chdir(cwd)
    return glob, (sess_box[0] if sess_box else None), g


def optimizer_records(g):
    out = []
    for o in g.optimizers:
        out.append(dict(kind=o.kind, hp=o.hp, var_list=sorted(v.name for v in (o.var_list or []))))
    return out


def manifest_case(script, mode, extra):
    """Build the graph at the script's own sizes (only BATCH_SIZE shrunk where the script allows; shapes of parameters do not
    depend on it) and stop when it opens its session."""
    consts = {'MODE': mode}
    consts['BATCH_SIZE'] = 10 if script.startswith('ssgan') else 2
    if script == 'ssgan_inference_chairs':
        consts['BATCH_SIZE'] = 2

    def stop(sess, glob):
        raise StopBuild()
    glob, _, g = run_script(script, consts, stop)
    params = {v.name: list(v.value.shape) for v in g.variables}
    trainable = sorted(v.name for v in g.variables if v.trainable)
    return dict(params=params, trainable=trainable, optimizers=optimizer_records(g), critic_iters=glob.get('CRITIC_ITERS'),
                constants={k: glob[k] for k in ('DIM', 'DIM_G', 'DIM_D', 'DIM_LATENT', 'N_COMS', 'BN_FLAG', 'LR', 'BETA1', 'LEN', 'DIM_OP',
                                                 'DIM_LATENT_G', 'DIM_LATENT_L', 'N_C') if k in glob and isinstance(glob[k], (int, float, bool))})


def trace_case(script, mode, extra):
    consts = dict(extra, MODE=mode)
    keep_names = ('disc_fake', 'disc_real', 'fake_x', 'q_z', 'p_z', 'rec_penalty', 'gradient_penalty')

    def start(sess, glob):
        g = shim.graph()
        for v in g.variables:                                  # deterministic non-trivial weights (oracle/reftrace.py)
            v.value = RT.det_weight(v.name, v.value.shape, shim.DTYPE)
        sess.noise = lambda node: RT.det_noise(len(sess.runs), node.id, node.attrs['rkind'], node.attrs['rshape'], node.attrs.get('classes'))
        keep = []
        for n in keep_names:
            t = glob.get(n)
            for i, x in enumerate(t if isinstance(t, list) else [t]):
                if isinstance(x, shim.Tensor):
                    x.keep_name = n if not isinstance(t, list) else '%s[%d]' % (n, i)
                    keep.append(x)
        sess.keep_values = keep
    t0 = time.time()
    glob, sess, g = run_script(script, consts, start)
    names = {id(o): i for i, o in enumerate(g.optimizers)}
    runs = []
    for r in sess.runs:
        if not r.train and not any(getattr(f, 'kind', None) in ('group', 'assign') for f in (r.fetches if isinstance(r.fetches, (list, tuple)) else [r.fetches])):
            continue                                           # (sample / reconstruction fetches of the logging code)
        rec = dict(run=r.index, draws=[[int(i), k, list(a.shape)] for i, k, a in r.draws], feeds=[], train=[], kept={})
        for ph, val in r.feed.items():
            key = STUBS.fed.get(id(val))
            rec['feeds'].append(dict(placeholder=ph.id, shape=list(np.shape(val)), stream=key[0] if key else None, index=key[1] if key else None,
                                     spec=list(STUBS.specs[key[0]]) if key else None,
                                     digest=RT.digest('feed%d' % ph.id, val)))
        for opt, cost, grads in r.train:
            rec['train'].append(dict(optimizer=names[id(opt)], cost=cost, t=opt.t,
                                     grads={n: (None if gr is None else RT.digest(n, gr)) for n, gr in sorted(grads.items())}))
        if not r.train:
            rec['other'] = 'assign-group'                      # wali: session.run(clip_disc_weights)
        for t in sess.keep_values:
            if t.id in r.kept:
                rec['kept'][t.keep_name] = RT.digest(t.keep_name, r.kept[t.id])
        runs.append(rec)
    sc = {k: glob[k] for k in ('DIM_LATENT', 'BN_FLAG', 'Z_SAMPLES', 'N_COMS', 'LR', 'BETA1', 'LAMBDA', 'DIM_LATENT_G', 'DIM_LATENT_L', 'N_C', 'LEN', 'DIM_OP')
          if k in glob and isinstance(glob[k], (int, float, bool))}
    return dict(constants={k: v for k, v in consts.items()}, script_constants=sc, params={v.name: list(v.value.shape) for v in g.variables},
                optimizers=optimizer_records(g), critic_iters=glob.get('CRITIC_ITERS'), runs=runs,
                final={v.name: RT.digest(v.name, v.value, RT.FINAL_SAMPLES) for v in g.variables},
                random_nodes=[[n.id, n.attrs['rkind'], list(n.attrs['rshape'])] for n in g.nodes if n.kind == 'random'],
                seconds=round(time.time() - t0, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None, help='script:MODE[,script:MODE...]')
    ap.add_argument('--manifest-only', action='store_true')
    ap.add_argument('--trace-only', action='store_true')
    args = ap.parse_args()
    if not os.path.isdir(REF):
        raise SystemExit('%s is not here: this generator runs in the build container only' % REF)
    install()
    want = set(args.only.split(',')) if args.only else None
    mpath, tpath = os.path.join(HERE, 'param_manifest.json'), os.path.join(HERE, 'reference_trace.json')
    manifest = json.load(open(mpath)) if (want and os.path.exists(mpath)) else {}
    traces = json.load(open(tpath)) if (want and os.path.exists(tpath)) else {}
    for script, mode, extra in CASES:
        key = '%s:%s' % (script, mode)
        if want and key not in want:
            continue
        if not args.trace_only:
            manifest[key] = manifest_case(script, mode, extra)
            sys.stderr.write('[manifest] %-45s %d params\n' % (key, len(manifest[key]['params'])))
        if not args.manifest_only:
            traces[key] = trace_case(script, mode, extra)
            sys.stderr.write('[trace]    %-45s %d runs, %.1f s\n' % (key, len(traces[key]['runs']), traces[key]['seconds']))
    if not args.trace_only:
        json.dump(manifest, open(mpath, 'w'), indent=1, sort_keys=True)
    if not args.manifest_only:
        json.dump(traces, open(tpath, 'w'), sort_keys=True, separators=(',', ':'))


if __name__ == '__main__':
    main()

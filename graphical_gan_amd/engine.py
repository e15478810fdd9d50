"""Step scheduler of the training loop (gmgan_inference_cifar10.py:480-494 and its nine siblings).

  iteration 0        : CRITIC_ITERS critic steps only
  iteration it > 0   : one generator(+extractor) step, then CRITIC_ITERS critic steps
  every step         : fresh minibatch + fresh noise, forward of all nets, backward, one Adam step
The reference re-enters TensorFlow (`session.run`) per step with a feed_dict host->device copy.  Here a step
is a HIP graph: the first call runs eagerly (warm-up, allocator priming), then forward + backward + gradient
packing + Adam are captured once with torch.cuda.CUDAGraph and replayed; inputs live in static device
buffers (a device-resident ring of pre-staged minibatches feeds them with one d2d copy), noise is drawn on
device inside the graph.  With data parallelism the graph is split around the gradient all-reduce:
[fwd+bwd+pack] -> RCCL all-reduce of the flat bucket -> [Adam].
"""
import contextlib
import os

import numpy as np
import torch
import torch.distributed as dist

from . import tflib as lib
from . import functional as F
from . import optim as _optim
from .models import GraphicalGAN


# Other host threads keep calling into the HIP runtime while a step is being captured (the RCCL process group's watchdog polls
# events of finished collectives): thread-local capture mode keeps those calls from invalidating the capture.
_CAPTURE_MODE = os.environ.get('GGAN_CAPTURE_MODE', 'thread_local')


_DP_GRAPH_OK = {}


def _dump_graph_dot(g, path):
    """GGAN_GRAPH_DOT=<file>: the captured iteration graph (nodes = kernel launches, edges = stream order and event waits) as Graphviz text,
    through hipGraphDebugDotPrint on the graph torch keeps (CUDAGraph(keep_graph=True)); torch's own debug_dump writes nothing on this build.
    Diagnostic only (tools/graph_edges.py reads it): which launches a replay may run side by side is decided by these edges."""
    import ctypes
    try:
        hip = ctypes.CDLL('libamdhip64.so')
        rc = hip.hipGraphDebugDotPrint(ctypes.c_void_p(g.raw_cuda_graph()), str(path).encode(), ctypes.c_uint(int(os.environ.get('GGAN_GRAPH_DOT_FLAGS', '0'))))
        if rc != 0:
            print('[engine] hipGraphDebugDotPrint returned %d' % rc)
    except (OSError, AttributeError, RuntimeError) as e:
        print('[engine] graph dump failed: %s' % e)


def dp_graph_selftest(device, comm):
    """Can an all-reduce on the directly bound RCCL communicator (rccl.py) be captured in a HIP graph and replayed here?  Two rounds,
    each closed by an ordinary (eager, torch.distributed) MIN all-reduce so that every replica takes the same path and no rank ever
    waits in a captured collective another rank never joins: (1) every rank CAPTURES a tiny sum-all-reduce; only if all of them could,
    (2) every rank replays it twice and checks the sum.  Cached per device."""
    key = str(device)
    if key in _DP_GRAPH_OK:
        return _DP_GRAPH_OK[key]

    def agree(ok):
        flag = torch.tensor([1.0 if ok else 0.0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(float(flag[0]) > 0.5)
    world = dist.get_world_size()
    t = torch.ones(256, device=device)
    s = torch.cuda.Stream(device=device)
    g, v, ok = None, None, True
    try:
        s.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(s):
            comm.all_reduce_(t.clone())                   # (communicator warm-up outside the capture)
            torch.cuda.synchronize(device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s, capture_error_mode=_CAPTURE_MODE):
                u = t * 1.0
                comm.all_reduce_(u)
                v = u + 1.0
        torch.cuda.current_stream(device).wait_stream(s)
    except Exception as e:                                # noqa: BLE001 (any capture failure selects the cut graphs)
        print('[engine] all-reduce capture self-test failed (%s): falling back to cut graphs' % (str(e).splitlines()[0] if str(e) else type(e).__name__))
        ok = False
    if agree(ok):
        try:
            with torch.cuda.stream(s):
                for _ in range(2):
                    g.replay()
            torch.cuda.synchronize(device)
            ok = abs(float(v[0]) - (world + 1.0)) <= 1e-6
        except Exception as e:                            # noqa: BLE001
            print('[engine] replay of the captured all-reduce failed (%s): falling back to cut graphs' % type(e).__name__)
            ok = False
        ok = agree(ok)
    else:
        ok = False
    _DP_GRAPH_OK[key] = ok
    return ok


class Trainer(object):
    def __init__(self, cfg, device=None, graph=True, seed=1234, inject_noise=False, model=None, sync_bn=False):
        """sync_bn: BatchNorm statistics over the global batch of all replicas (N GPUs x B/N then reproduce
        1 GPU x B).  The statistics exchange is a host-issued collective inside the forward and the backward, so this mode
        runs the steps eagerly (no HIP graphs); it is the parity mode, per-replica statistics stay the throughput default.
        model: an object with forward_nets / forward / feed_buffers / sample_noise / set_batch / single_contribution
        (default: models.GraphicalGAN(cfg); models_ssgan.StateSpaceGAN for the state-space scripts)."""
        self.cfg = cfg
        self.device = torch.device(device) if device is not None else lib.get_device()
        lib.set_device(self.device)
        self.model = model if model is not None else GraphicalGAN(cfg)
        self.graph_enabled = graph
        self.inject_noise = inject_noise
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)
        # static input buffers (what the reference feeds / samples per session.run)
        self.feed = self.model.feed_buffers(self.device)
        if hasattr(self.model, 'cfg') and isinstance(self.feed, dict) and ('p_z_noise' in self.feed or 'p_z_g' in self.feed):
            self.feed['rng_state'] = F.noise_state(self.device, seed)      # functional.noise_fill_: all noise of a step in one launch
        self._graphs = {}
        self._calls = {'gen': 0, 'disc': 0}
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # data-parallel runs split the step graph around the gradient all-reduce; the flag lets a single GPU exercise
        # exactly that code path (the collective is then a 1-rank no-op)
        # data-parallel steps: by default the gradient exchange is captured INSIDE the step graph (one graph per step again:
        # RCCL's kernels become graph nodes on the process group's stream, joined before the update), so a replica's step is the
        # single-GPU step plus the exposed part of the exchange.  GGAN_DP_GRAPH=0 restores the cut graphs
        # ([nets] [critic+backward+pack] -> host-issued all-reduce -> [update]) with the exchange overlapped across steps.
        dp = self.world > 1 or bool(os.environ.get('GGAN_FORCE_ALLREDUCE')) and dist.is_available() and dist.is_initialized()
        # the exchange itself goes through a communicator of our own (rccl.py: no process-group watchdog polls its stream, so it can be
        # captured); on other backends (gloo stages device tensors through the host) it stays a host-issued collective between cut graphs
        from . import rccl
        self.comm = rccl.get(self.device) if dp else None
        self.dp_graph = (dp and self.comm is not None and os.environ.get('GGAN_DP_GRAPH', '1') != '0'
                         and not os.environ.get('GGAN_FORCE_SPLIT_GRAPH'))
        if self.dp_graph and graph and not dp_graph_selftest(self.device, self.comm):
            self.dp_graph = False
        self.split_graph = (self.world > 1 and not self.dp_graph) or bool(os.environ.get('GGAN_FORCE_SPLIT_GRAPH'))
        self.sync_bn = bool(sync_bn) and (self.world > 1 or dp)       # (dp at world 1: the one-rank RCCL rehearsal)
        if self.sync_bn:
            lib.ops.batchnorm.set_sync_group(True)
            if self.comm is not None:
                rccl.get_stats(self.device)   # (the communicator goes serial: statistics gathers and gradient buckets in one order, functional._all_gather_rows)
            if self.comm is None:       # the statistics exchange is a host-issued collective inside the passes: eager steps.  With the
                self.graph_enabled = False   # direct communicator it is an enqueue on the step's stream and is captured like a kernel
            if hasattr(self.model, 'fork_nets'):
                self.model.fork_nets = False      # keep the statistics exchanges of the two passes in one stream order
        elif self.comm is not None:
            rccl.release_stats()                  # (a Trainer with cross-replica BatchNorm before this one left the communicator serial)
        # the pack kernel may sum the filter-gradient slabs only if every parameter receives ONE gradient contribution per
        # backward pass (true when the critic sees [fake; real] as one batch; the wali-gp penalty re-enters the critic)
        self.single_contrib = bool(self.model.single_contribution)
        self._opts = None
        self.keep_outputs, self.last_out = False, {}
        self._pending = None         # (work handle, Adam graph) of a critic-step exchange still in flight (see step())

    # ---- inputs ---------------------------------------------------------------------------------------
    def set_feed(self, feed):
        """Copy injected numpy inputs/noise (oracle.step.make_feed layout) into the static buffers."""
        c = self.cfg
        for k, v in feed.items():
            if k == 'k_idx':
                oh = np.zeros((c.B, c.K), np.float32)
                oh[np.arange(c.B), v] = 1
                self.feed['k_onehot'].copy_(torch.as_tensor(oh))
            else:
                self.feed[k].copy_(torch.as_tensor(np.asarray(v)))
        self._ahead_sync_feeds()

    def _ahead_sync_feeds(self):
        """injected inputs (inject_noise: tests) reach the per-step feeds of the ahead-of-time nets passes too"""
        st = getattr(self, '_ahead', None)
        if st is None or not self.inject_noise:
            return
        for f in st['feeds']:
            for k, v in self.feed.items():
                if torch.is_tensor(v) and k != 'rng_state' and torch.is_tensor(f.get(k)) and f[k].data_ptr() != v.data_ptr():
                    f[k].copy_(v)

    def set_batch(self, batch):
        self.model.set_batch(self.feed, batch)

    def use_host_ring(self, get_epoch, slots=None, pick=None):
        """use_ring fed from HOST memory: get_epoch() returns an iterator of host minibatches (numpy, the placeholder's shape);
        a data.RingFeeder copies them into the ring's slots on a copy stream, one iteration ahead of the replay that reads them,
        so the PCIe transfer runs under the previous iteration instead of in front of every step."""
        from .data import RingFeeder
        per_it = 1 + self.cfg.critic_iters
        R = int(slots) if slots else 4 * per_it
        assert R >= 3 * per_it, 'the ring must hold three iterations of minibatches'
        like = self.feed['real_x_int']
        ring = torch.zeros((R,) + tuple(like.shape), dtype=torch.int32, device=self.device)
        self.use_ring(list(ring))                         # (stacks a copy: take the Trainer's tensor as the feeder's target)
        ring = self.feed['ring'][0]
        self._feeder = RingFeeder(get_epoch, self.device, ring, pick=pick)
        self._ring_pos = 0                                # steps taken since use_ring == next slot to be read
        self._ring_ready = self._feeder.fill(0, per_it)   # the first iteration's minibatches

    def use_ring(self, batches):
        """batches: equal-shape int32 device minibatches (model.synthetic_ring, or a loader's staging slots).  From here on every
        step reads its minibatch from this ring inside its own launches -- slot (generator steps taken + critic steps taken) mod R,
        counted by the optimizers' device-side step counts -- instead of a staging buffer the host copies the next minibatch
        into: nothing is issued between two step graphs, and iteration() can replay a whole iteration as ONE graph.  Needs both
        step kinds to have run once (the optimizers must exist); int32 image scripts only (MNIST feeds floats: unchanged)."""
        from .optim import _optimizers
        ring = torch.stack(list(batches)).contiguous()
        if ring.dtype != torch.int32 or 'real_x_int' not in self.feed or ring[0].numel() != self.feed['real_x_int'].numel():
            raise ValueError('use_ring: int32 minibatches of the shape of the real_x_int placeholder')
        ctr = {}
        for key, o in _optimizers.items():
            if key[0] in ('gen', 'disc'):
                ctr[key[0]] = o.step
        if 'gen' not in ctr:
            raise RuntimeError('use_ring: run one eager iteration first (the optimizers own the step counters)')
        self.flush()
        torch.cuda.synchronize(self.device)
        taken = sum(int(t.item()) for t in ctr.values())
        from . import optim as _optim
        self._ring_epoch = _optim.STATE_EPOCH[0]
        self.feed['ring'] = (ring, ctr['gen'], ctr.get('disc'), -taken)
        self._graphs = {}                      # (captured steps read the staging buffer)
        self._iter_graph = None

    def _sample_noise(self):
        self.model.sample_noise(self.feed)

    # ---- one session.run ------------------------------------------------------------------------------
    def _nets(self, feed=None):
        feed = self.feed if feed is None else feed
        self._sl0 = lib.second_leaf_count()          # (see _no_second_leaves)
        if hasattr(self.model, 'begin_nets'):
            self.model.begin_nets(feed)
        if not self.inject_noise:
            self.model.sample_noise(feed)
        return self.model.forward_nets(feed)

    # ---- the critic steps' nets passes ahead of time (round 5) ---------------------------------------------------------------------
    # An iteration of the WGAN scripts is one generator step and CRITIC_ITERS critic steps (gan_inference_cifar10.py:351-366), and
    # every critic step starts with a Generator and an Extractor pass on a fresh minibatch and fresh noise that read no critic weight:
    # inside the iteration graph the passes of critic steps 2.. are issued as ONE chain on a stream of their own as soon as step 1's
    # passes are, and run beside the critic steps before theirs (~110 us each of a chain that otherwise stands in front of its critic
    # step; headline -4.4 %; the bound with those passes removed altogether is -11 %: profiles/r05_notes.md).  What that needs: a feed of its own per
    # step (noise, [fake; real] pair buffers), the noise launches in the order of the steps (they share the generator state: the chain
    # is forked behind step 1's launch and joined before the iteration ends), and a ring slot that does not depend on WHEN a pass
    # runs -- the passes read the critic's step count from a snapshot taken in front of critic step 1 plus their distance from it.
    def _ahead_ok(self, kinds):
        c = self.cfg
        kinds = list(kinds)
        if 'gen' in kinds[kinds.index('disc') if 'disc' in kinds else 0:]:
            # the chain reads the Generator's / Extractor's weights with no ordering against a generator update that follows a critic
            # step inside the same graph: only [generator step(s)] + [critic steps] is race-free
            return False
        # (replicas too, since round 6: the passes read no critic weight and no gradient bucket, so the in-graph exchange -- a branch of its
        #  own on the communicator's stream, forked behind a step's pack launch and joined in front of its update -- never meets them; the
        #  cut-graph exchange (split_graph) keeps one graph per step and has no iteration graph to put them in)
        return (not os.environ.get('GGAN_NO_NETS_AHEAD') and kinds.count('disc') >= 2 and (self.world == 1 or self.dp_graph)
                and not self.split_graph and not self.sync_bn
                and hasattr(self.model, 'fork_now') and hasattr(self.model, 'feed_buffers')
                and not getattr(c, 'K', 0) and not getattr(c, 'agg', None) and getattr(c, 'dataset', '') != 'mnist'
                and isinstance(self.feed, dict) and self.feed.get('ring') is not None and self.feed['ring'][2] is not None)

    def _ahead_prepare(self, kinds):
        """outside the capture: the per-step feeds, the snapshot of the critic's step count, the stream and its workspace"""
        n = list(kinds).count('disc') - 1
        st = getattr(self, '_ahead', None)
        if st is None or len(st['feeds']) != n or st['ring'] is not self.feed['ring']:
            ring, gen_ctr, disc_ctr, off = self.feed['ring']
            snap = torch.zeros_like(disc_ctr)
            feeds = []
            for j in range(n):
                f = self.model.feed_buffers(self.device)
                if 'rng_state' in self.feed:
                    f['rng_state'] = self.feed['rng_state']               # ONE generator state: the draws come in the order of the steps
                f['ring'] = (ring, gen_ctr, snap, off + j + 1)
                feeds.append(f)
            st = self._ahead = dict(feeds=feeds, snap=snap, ring=self.feed['ring'], stream=F.shared_stream(self.device, 'nets'))
            with torch.cuda.stream(st['stream']):
                F.workspace(self.device)
            torch.cuda.synchronize(self.device)
        self._ahead_sync_feeds()
        st.update(i=0, nets=[], events=[], pending=None)
        return st

    def _ahead_issue(self, st, k):
        """the nets pass of critic step k + 2 on the chain's stream: one chain (Generator, then Extractor), nothing forked off it"""
        m, ns = self.model, st['stream']
        saved = (m._pending_join, getattr(m, '_noise_event', None), m._early, m.fork_now)
        m.fork_now = False
        try:
            with torch.cuda.stream(ns), F.launch_hint(int(os.environ.get('GGAN_AHEAD_WGS', '128'))):
                st['nets'].append(self._nets(st['feeds'][k]))
                ev = torch.cuda.Event()
                ev.record(ns)
                st['events'].append(ev)
        finally:
            m._pending_join, m._noise_event, m._early, m.fork_now = saved

    def _ahead_step(self):
        """(nets, feed) of the next critic step of the iteration being captured.  The chain is forked behind critic step 1's own nets
        pass (its noise launch, the snapshot); the pass of step i + 2 is held back until critic step i + 1 has begun, so that one pass runs
        beside each critic step instead of all of them beside the first (3.886 -> 3.836 ms; GGAN_AHEAD_BUNCH=1: all at once).  WHERE in
        step i + 1 it is released is GGAN_AHEAD_AT: 'begin' (in front of its critic pass), 'bwd' (behind its forward: the pass then runs
        beside the backward pass and the step's tail -- the last data gradient, the thin filter gradient, the pack launch -- where a
        profiled iteration has one kernel in flight), 'pack' (behind its backward launches)."""
        st = self._ahead_run
        i, n = st['i'], len(st['feeds'])
        st['i'] += 1
        cur = torch.cuda.current_stream(self.device)
        bunch = bool(os.environ.get('GGAN_AHEAD_BUNCH'))
        at = self._ahead_at()
        if i == 0:
            st['snap'].copy_(self.feed['ring'][2])                        # the critic's step count in front of critic step 1
            nets = self._nets()
            st['stream'].wait_stream(cur)                                 # (behind step 1's noise launch and the snapshot)
            if bunch or at == 'begin':
                for k in range(n if bunch else 1):
                    self._ahead_issue(st, k)
            else:
                st['pending'] = 0
            return nets, self.feed
        if not bunch and i < n:
            if at == 'begin':
                ev = torch.cuda.Event()
                ev.record(cur)
                st['stream'].wait_event(ev)
                self._ahead_issue(st, i)
            else:
                st['pending'] = i
        cur.wait_event(st['events'][i - 1])
        self._sl0 = lib.second_leaf_count()
        return st['nets'][i - 1], st['feeds'][i - 1]

    @staticmethod
    def _ahead_at():
        at = os.environ.get('GGAN_AHEAD_AT', 'begin')
        assert at in ('begin', 'bwd', 'pack'), at
        return at

    def _ahead_release(self, point):
        """issue the nets pass a critic step holds back, if `point` is where GGAN_AHEAD_AT releases it (engine._fwd_bwd calls this behind the
        step's forward and behind its backward launches)"""
        st = getattr(self, '_ahead_run', None)
        if st is None or st.get('pending') is None or self._ahead_at() != point:
            return
        k, st['pending'] = st['pending'], None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        st['stream'].wait_event(ev)
        self._ahead_issue(st, k)

    def _forward(self, feed, which, nets):
        """model.forward for a step whose backward follows at once: the critic head may then leave its cost's gradient behind with
        its own forward and take the cost's value into its backward launch (functional.head_bce_hint, models.GraphicalGAN.head_hint)"""
        F.drop_pending_costs()
        F.drop_late_terms()
        self.model.head_hint = True
        try:
            return self.model.forward(feed, which, nets)
        finally:
            self.model.head_hint = False

    @staticmethod
    def _costs_settled():
        """after a step's backward: every cost a hinted critic head owed has been written (functional.settle_cost); a cost tensor handed
        out while its value is still owed would be unwritten memory"""
        if F.pending_costs():
            F.drop_pending_costs()
            raise RuntimeError('a hinted critic head still owes its cost value after the backward pass (functional.head_bce_hint): '
                               'the cost was not differentiated through that head')

    def _fwd_bwd(self, which, nets=None, fuse_update=False, feed=None):
        """fuse_update: the caller applies the update next with nothing but a (single-replica: empty) exchange in between"""
        out = self._forward(self.feed if feed is None else feed, which, nets if nets is not None else self._nets())
        self._ahead_release('bwd')
        if self.keep_outputs:        # (tests: the critic logits of a captured step -- static graph memory, valid after every replay)
            det = lambda v: [t.detach() for t in v] if isinstance(v, (list, tuple)) else v.detach()    # (no tape kept alive across steps)
            self.last_out[which] = {k: det(v) for k, v in out.items() if k in ('disc_fake', 'disc_real')}
        op = out[which + '_train_op']
        opt = op.optimizer
        # (the filter-gradient slabs are summed by the pack kernel: legal under the same one-contribution condition)
        with F.defer_wgrad_reduce(self.single_contrib):
            grads = opt.compute_gradients(op.cost)
            late = F.add_late_terms()        # (the penalty's value into the critic cost, on the penalty's stream: the backward pass did not wait for it)
            self._ahead_release('pack')
            keep = opt.pack(grads, fuse_update=fuse_update)
            for ev in late:
                torch.cuda.current_stream(self.device).wait_event(ev)
        self._costs_settled()
        return out[which + '_cost'].detach(), opt, keep

    def _two_bucket_plan(self, opt, nets):
        """(k, flat offset, cut tensors) when this generator step can exchange the Generator's gradients first, else None"""
        cut = self.model.cut_tensors(nets) if hasattr(self.model, 'cut_tensors') else None
        if not cut or any(not t.requires_grad for t in cut) or os.environ.get('GGAN_ONE_BUCKET'):
            return None
        sp = opt.split_at(lambda p: 'Extractor' not in getattr(p, 'param_name', ''))
        return None if sp is None else (sp[0], sp[1], cut)

    def _bwd_phase1(self, nets):
        """generator step up to the cut: critic pass, backward to the Generator's parameters and to the Extractor's outputs,
        Generator gradients packed.  Returns everything phase 2 needs."""
        out = self._forward(self.feed, 'gen', nets)
        op = out['gen_train_op']
        opt = op.optimizer
        plan = self._two_bucket_plan(opt, nets)
        if plan is None:
            return None
        k, off, cut = plan
        self._no_second_leaves(opt)
        if opt._one is None or opt._one.shape != op.cost.shape:
            opt._one = F.unit_seed(op.cost)
        with F.defer_wgrad_reduce(self.single_contrib), F.serial_backward():      # (Generator half, then Extractor half: one chain at a time)
            g = torch.autograd.grad(op.cost, list(opt.params[:k]) + cut, grad_outputs=opt._one, allow_unused=True)
            keep = opt.pack_subset(g[:k], 0, k, bump=True)
        self._costs_settled()
        return dict(cost=out['gen_cost'].detach(), opt=opt, k=k, off=off, cut=cut, g_cut=g[k:], keep=keep)

    def _bwd_phase2(self, st):
        """the Extractor's backward from the gradients at the cut; its gradients packed behind the Generator's"""
        opt, k = st['opt'], st['k']
        pairs = [(t, g) for t, g in zip(st['cut'], st['g_cut']) if g is not None]
        with F.defer_wgrad_reduce(self.single_contrib), F.serial_backward():
            g = torch.autograd.grad([t for t, _ in pairs], opt.params[k:], grad_outputs=[gg for _, gg in pairs], allow_unused=True)
            return opt.pack_subset(g, k, len(opt.params), bump=False)

    def _no_second_leaves(self, opt):
        """the two-bucket paths ask the tape for opt.params directly (not through AdamOptimizer.compute_gradients): a contribution that
        arrives through a second leaf (tflib.second_leaf: a weight two passes of the step reach) would be dropped without a word"""
        if lib.second_leaf_count() != getattr(self, '_sl0', lib.second_leaf_count()):
            raise NotImplementedError('two-bucket gradient exchange in a step that handed out second leaves (a weight reached by two '
                                      'passes): its second contribution would be lost -- use one bucket (GGAN_ONE_BUCKET=1)')

    def _eager(self, which):
        self.flush()
        with self._launch_hint(which):       # (the warm-up steps of a capture launch what the capture will launch)
            cost, opt, _ = self._fwd_bwd(which, fuse_update=True)
        opt.all_reduce()
        opt.update()
        return cost

    def _step_as_captured(self, which):
        """eager_as_captured for the one-graph-per-step workloads: the step as its graph launches it (site plan of a [which] graph)"""
        self.flush()
        F.set_site_plan(self._site_plan([which], False))
        try:
            cost, _, _ = self._step_body(which)
        finally:
            F.set_site_plan(None)
        return cost

    def _optimizers(self):
        from .optim import _optimizers
        return list(_optimizers.values())

    def _capture(self, which):
        # two-stream nets pass inside the step graph (models.GraphicalGAN.forward_nets): single-graph steps only -- with the
        # graph cut for the gradient exchange it measured slower (40.1 k vs 41.7 k img/s on the forced-split path)
        forkable = hasattr(self.model, 'fork_now') and not self.split_graph and not self.sync_bn
        if forkable:
            self.model.fork_now = True
        try:
            return self._capture_impl(which)
        finally:
            if forkable:
                self.model.fork_now = False

    @contextlib.contextmanager
    def eager_as_captured(self):
        """eager steps that launch what the captured graphs launch: the two-stream nets pass (and with it the launch plan of side-by-side
        conv chains, models.launch_hint / functional.target_workgroups) is otherwise switched on for captures only.  bench.py brackets
        the kernels of such steps with HIP events, so that its per-kernel table is the timed graph's kernel mix."""
        forkable = hasattr(self.model, 'fork_now') and not self.split_graph and not self.sync_bn
        prev_graph, self.graph_enabled = self.graph_enabled, False
        self._as_captured = True
        if forkable:
            self.model.fork_now = True
        try:
            yield
        finally:
            self.graph_enabled = prev_graph
            self._as_captured = False
            if forkable:
                self.model.fork_now = False

    def _step_body(self, which, ordinal=0):
        """one whole step as it is captured into a single graph: forward, backward, pack, (gradient exchange), update.
        ordinal: which of the iteration's steps of this kind it is -- the scope of its launch sites (functional.site_scope)"""
        with self._launch_hint(which), F.site_scope('%s%d' % (which, ordinal)):
            return self._step_body_impl(which)

    # ---- launch plans per launch site (round 6) ---------------------------------------------------------------------------------
    def site_plan_key(self, kinds, ahead):
        """which entry of graphical_gan_amd/site_plans.json describes the step graph about to be captured: everything the ORDER of its
        conv launches depends on"""
        c = self.cfg
        return '%s/%s/B%d/K%d/%s%s%s' % (getattr(c, 'dataset', '?'), getattr(c, 'mode', '?'), getattr(c, 'B', 0), getattr(c, 'K', 0) or 0,
                                        '+'.join(kinds), ('/ahead' + ('' if self._ahead_at() == 'begin' else '-' + self._ahead_at())) if ahead else '',
                                        '/dp' if self.dp_graph else '')

    def _site_plan(self, kinds, ahead):
        """the site plan of the graph about to be captured: GGAN_SITE_PLAN=0 none, =<file> that table, else the committed one"""
        env = os.environ.get('GGAN_SITE_PLAN', '')
        if env == '0':
            return None
        import json
        key = self.site_plan_key(kinds, ahead)
        self.last_site_plan_key = key
        if getattr(self, 'site_plan_override', None) is not None:       # (tools/site_sweep.py: a table per trial)
            return self.site_plan_override
        path = env or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'site_plans.json')
        try:
            tab = json.load(open(path))
        except (OSError, ValueError):
            if env:
                raise
            return None
        return (tab.get(key) or {}).get('sites')

    @contextlib.contextmanager
    def _launch_hint(self, which):
        """a step whose conv launches run as TWO chains side by side (models.launch_hint: the wali-gp critic step with the penalty pass on
        the second stream) plans each launch for fewer workgroups, so that the chains share the chip by CUs instead of time-slicing it"""
        hint = self.model.launch_hint(which) if hasattr(self.model, 'launch_hint') else 0
        if not hint:
            yield
            return
        with F.launch_hint(int(hint)):
            yield

    def _step_body_impl(self, which):
        st, nets = None, None
        if self.dp_graph and which == 'gen':
            # two gradient buckets inside the one graph: the Generator's bucket is on the wire (the process group's
            # stream: a parallel branch of the graph) while the Extractor's backward pass still runs
            nets = nets if nets is not None else self._nets()
            cut = self.model.cut_tensors(nets) if hasattr(self.model, 'cut_tensors') else None
            # (two buckets in the generator step mean two autograd passes, Generator half then Extractor half -- and those two backward
            #  chains otherwise run SIDE BY SIDE on two streams, planned for half the chip each: one rank, forced exchange, 1.20 ms with
            #  the split against 1.09 without.  The overlap the split buys is at most the Generator bucket's time on the wire during the
            #  Extractor's ~100 us of backward, so the default is one bucket here; GGAN_GEN_TWO_BUCKETS=1 restores the split.  The
            #  critic step keeps its two buckets: one chain, and the large tail bucket travels during the conv stack's backward.)
            st = self._bwd_phase1(nets) if (cut and os.environ.get('GGAN_GEN_TWO_BUCKETS') and not os.environ.get('GGAN_ONE_BUCKET')) else None
            if st is not None:
                opt, cost = st['opt'], st['cost']
                w1 = opt.all_reduce(async_op=True, lo=0, hi=st['off'])
                _optim._xlog('backward', 'Extractor')
                keep = (st['keep'], self._bwd_phase2(st), st, nets)
                w2 = opt.all_reduce(async_op=True, lo=st['off'], hi=None)
                for w in (w1, w2):
                    if w is not None:
                        _optim._xlog('wait')
                        w.wait()
            else:
                cost, opt, keep = self._fwd_bwd(which, nets)
                opt.all_reduce()
        else:
            feed = None
            if which == 'disc' and getattr(self, '_ahead_run', None) is not None:
                nets, feed = self._ahead_step()
            if self.dp_graph and which == 'disc' and hasattr(self.model, 'critic_cut') and not os.environ.get('GGAN_ONE_BUCKET'):
                cost, opt, keep = self._disc_two_buckets(nets, feed)
            else:
                cost, opt, keep = self._fwd_bwd(which, nets, fuse_update=not self.dp_graph, feed=feed)
                if self.dp_graph:
                    opt.all_reduce()
            if feed is not None:
                keep = (keep, nets)
        opt.update()
        return cost, opt, keep

    def _disc_two_buckets(self, nets=None, feed=None):
        """critic step with two gradient buckets: autograd reaches the critic's tail first, and that is where most of the bytes
        are (Discriminator.zx1: 2.6 M of the 4.1 M parameters) -- its bucket is exchanged while the conv stack's backward pass
        (two thirds of the critic's backward time) still runs; the conv stack's bucket follows.  Same sums as one bucket."""
        out = self._forward(self.feed if feed is None else feed, 'disc', nets if nets is not None else self._nets())
        self._ahead_release('bwd')
        op = out['disc_train_op']
        opt = op.optimizer
        cutinfo = self.model.critic_cut()
        sp = None
        if cutinfo is not None:
            conv = tuple('Discriminator.%d.' % (i + 1) for i in range(cutinfo[1]))
            sp = opt.split_at(lambda p: getattr(p, 'param_name', '').startswith(conv))
        if opt._one is None or opt._one.shape != op.cost.shape:
            opt._one = F.unit_seed(op.cost)
        if sp is None:
            with F.defer_wgrad_reduce(self.single_contrib):
                grads = opt.compute_gradients(op.cost)
                late = F.add_late_terms()
                self._ahead_release('pack')
                keep = opt.pack(grads)
                for ev in late:
                    torch.cuda.current_stream(self.device).wait_event(ev)
            opt.all_reduce()
            self._costs_settled()
            return out['disc_cost'].detach(), opt, (keep, out)
        k, off = sp
        cut = cutinfo[0]
        self._no_second_leaves(opt)
        with F.defer_wgrad_reduce(self.single_contrib):
            g = torch.autograd.grad(op.cost, list(opt.params[k:]) + [cut], grad_outputs=opt._one, allow_unused=True)
            keep_a = opt.pack_subset(g[:-1], k, len(opt.params), bump=True)
            w1 = opt.all_reduce(async_op=True, lo=off, hi=None)
            _optim._xlog('backward', 'critic conv stack')
            g2 = torch.autograd.grad([cut], opt.params[:k], grad_outputs=[g[-1]], allow_unused=True)
            self._ahead_release('pack')
            keep_b = opt.pack_subset(g2, 0, k, bump=False)
            w2 = opt.all_reduce(async_op=True, lo=0, hi=off)
        for w in (w1, w2):
            if w is not None:
                _optim._xlog('wait')
                w.wait()
        self._costs_settled()
        return out['disc_cost'].detach(), opt, (keep_a, keep_b, g, g2, out)

    def _capture_impl(self, which):
        # warm-up eagerly on a side stream (allocator + lazy init), restoring optimizer state afterwards
        if getattr(self, '_cap_stream', None) is None:
            self._cap_stream = F.shared_stream(self.device, 'capture')  # warm-up AND capture run on this stream, so the
        s = self._cap_stream                                            # per-stream scratch buffers exist before capture
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            # (the optimizers exist: the first call of each kind ran eagerly)
            snap = [(o, o.theta.clone(), o.m.clone(), o.v.clone(), o.step.clone()) for o in self._optimizers()]
            # (the device noise state too: a graph=True and a graph=False run with the same seed draw the same noise sequence)
            rng = self.feed.get('rng_state') if isinstance(self.feed, dict) else None
            rng_snap = rng.clone() if torch.is_tensor(rng) else None
            self._eager(which)
            if self.split_graph:
                self._eager(which)
            else:
                # dress rehearsal: the step exactly as it will be captured (site plans included), once, eagerly -- every plan-time cache
                # and lazily set function attribute then exists before the capture (see _capture_iteration)
                F.set_site_plan(self._site_plan([which], False))
                try:
                    self._step_body(which)
                finally:
                    F.set_site_plan(None)
            for o, th, m, v, st in snap:
                o.theta.copy_(th); o.m.copy_(m); o.v.copy_(v); o.step.copy_(st)
            if rng_snap is not None:
                rng.copy_(rng_snap)
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        g1 = torch.cuda.CUDAGraph()
        if not self.split_graph:
            F.set_site_plan(self._site_plan([which], False))
            lib.drop_taps([which])
            try:
                with torch.cuda.graph(g1, stream=s, capture_error_mode=_CAPTURE_MODE):
                    cost, opt, keep = self._step_body(which)
            finally:
                F.set_site_plan(None)
            return dict(g0=None, g1=g1, g1b=None, split=None, g2=None, cost=cost, opt=opt, keep=keep)
        # data parallel: [forward + backward + pack] -> all-reduce -> [Adam].  Every step is cut once more, after the
        # Extractor/Generator passes (g0): they read no critic variable, so they run while the previous critic step's gradient
        # exchange is still on the wire (step()) -- under the next generator step, or under the next critic step when
        # CRITIC_ITERS > 1.
        g0 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g0, stream=s, capture_error_mode=_CAPTURE_MODE):
            nets = self._nets()
        g1b, st = None, None
        with torch.cuda.graph(g1, stream=s, pool=g0.pool(), capture_error_mode=_CAPTURE_MODE):
            if which == 'gen':
                # two gradient buckets: autograd reaches the Generator's parameters before the Extractor's, so the Generator
                # bucket goes on the wire while the Extractor's backward pass (g1b) still runs
                st = self._bwd_phase1(nets)
            if st is None:
                cost, opt, keep = self._fwd_bwd(which, nets)
        if st is not None:
            g1b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1b, stream=s, pool=g0.pool(), capture_error_mode=_CAPTURE_MODE):
                keep = (st['keep'], self._bwd_phase2(st), st)
            cost, opt = st['cost'], st['opt']
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=s, capture_error_mode=_CAPTURE_MODE):
            opt.update()
        return dict(g0=g0, g1=g1, g1b=g1b, split=(st['off'] if st is not None else None), g2=g2, cost=cost, opt=opt,
                    keep=(keep, nets))

    def flush(self):
        """Finish a critic step whose gradient exchange was left in flight: wait for it, apply its Adam update."""
        if self._pending is not None:
            work, g2 = self._pending
            self._pending = None
            if work is not None:
                work.wait()              # the current stream waits; the host does not block
            g2.replay()

    def step(self, which):
        """One gen or disc session.run on the current contents of the static buffers -> 0-dim cost tensor."""
        self._calls[which] += 1
        if self._calls['gen'] >= 2 and (self._calls['disc'] >= 2 or not getattr(self.cfg, 'critic_iters', 1)):
            lib.end_build_phase()        # both step kinds have been built once (critic-free modes: the generator step): from here on
                                         # the layer calls draw no initial values
        if not self.graph_enabled:
            if getattr(self, '_as_captured', False) and self._calls[which] > 1 and not self.split_graph:
                return self._step_as_captured(which)
            return self._eager(which)
        rec = self._graphs.get(which)
        if rec is None:
            if self._calls[which] == 1:
                # the very first call of each kind runs eagerly AND counts as a real step
                return self._eager(which)
            self.flush()
            rec = self._capture(which)
            self._graphs[which] = rec
        if rec['g2'] is None:            # single GPU: the whole step is one graph
            rec['g1'].replay()
            return rec['cost']
        if rec['g0'] is not None:
            rec['g0'].replay()           # overlaps the pending critic-gradient all-reduce
        self.flush()
        rec['g1'].replay()
        if rec['g1b'] is not None:       # generator step in two buckets
            w1 = rec['opt'].all_reduce(async_op=True, lo=0, hi=rec['split'])
            rec['g1b'].replay()          # the Extractor's backward overlaps the Generator bucket's exchange
            w2 = rec['opt'].all_reduce(async_op=True, lo=rec['split'], hi=None)
            for w in (w1, w2):
                if w is not None:
                    w.wait()
            rec['g2'].replay()
            return rec['cost']
        work = rec['opt'].all_reduce(async_op=True)
        if which == 'disc':
            self._pending = (work, rec['g2'])    # finished by the next step (or flush()): nothing before that reads it
        else:
            if work is not None:
                work.wait()
            rec['g2'].replay()
        return rec['cost']

    def gen_step(self, feed=None):
        if feed is not None:
            self.set_feed(feed)
        return self.step('gen')

    def disc_step(self, feed=None):
        if feed is not None:
            self.set_feed(feed)
        return self.step('disc')

    def _iteration_body(self, kinds, ahead):
        """the steps of one iteration as they are captured into ONE graph (and rehearsed eagerly in front of the capture): each step under
        its site scope, the critic steps' nets passes ahead of time when `ahead` is the state _ahead_prepare made"""
        costs, keeps, seen = {}, [], {}
        self._ahead_run = ahead
        try:
            for k in kinds:
                cost, opt, keep = self._step_body(k, seen.get(k, 0))    # (with the in-graph gradient exchange when there are replicas)
                seen[k] = seen.get(k, 0) + 1
                costs[k + '_cost'] = cost
                keeps.append((opt, keep))
        finally:
            self._ahead_run = None
        if ahead is not None:
            torch.cuda.current_stream(self.device).wait_stream(ahead['stream'])
            keeps.append((None, (ahead['nets'], ahead['events'])))
        return costs, keeps

    def _iteration_as_captured(self, kinds):
        """eager_as_captured: one iteration launched eagerly exactly as the iteration graph launches it -- site plans, ahead-of-time passes --
        on the capture stream (whose per-stream workspaces exist); the steps DO update the weights"""
        forkable = hasattr(self.model, 'fork_now') and not self.sync_bn
        s = self._cap_stream if getattr(self, '_cap_stream', None) is not None else F.shared_stream(self.device, 'capture')
        self._cap_stream = s
        use_ahead = self._ahead_ok(kinds)
        F.set_site_plan(self._site_plan(kinds, use_ahead))
        if forkable:
            self.model.fork_now = True
        try:
            s.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(s):
                costs, _ = self._iteration_body(kinds, self._ahead_prepare(kinds) if use_ahead else None)
            torch.cuda.current_stream(self.device).wait_stream(s)
            return costs
        finally:
            F.set_site_plan(None)
            if forkable:
                self.model.fork_now = False

    def _capture_iteration(self, kinds):
        """one HIP graph for a whole iteration (ring mode, one GPU): [generator step, critic step x CRITIC_ITERS], each
        forward + backward + pack + Adam -- no graph-launch gap between the steps"""
        forkable = hasattr(self.model, 'fork_now') and not self.sync_bn
        if forkable:
            self.model.fork_now = True
        try:
            if getattr(self, '_cap_stream', None) is None:
                self._cap_stream = F.shared_stream(self.device, 'capture')
            s = self._cap_stream
            s.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(s):
                snap = [(o, o.theta.clone(), o.m.clone(), o.v.clone(), o.step.clone()) for o in self._optimizers()]
                rng = self.feed.get('rng_state') if isinstance(self.feed, dict) else None
                rng_snap = rng.clone() if torch.is_tensor(rng) else None
                for k in kinds:
                    self._eager(k)
                for o, th, m, v, st in snap:
                    o.theta.copy_(th); o.m.copy_(m); o.v.copy_(v); o.step.copy_(st)
                if rng_snap is not None:
                    rng.copy_(rng_snap)
            torch.cuda.current_stream(self.device).wait_stream(s)
            torch.cuda.synchronize(self.device)
            if getattr(self, 'record_site_log', False):
                F.record_sites(True)
            use_ahead = self._ahead_ok(kinds)
            F.set_site_plan(self._site_plan(kinds, use_ahead))

            try:
                # dress rehearsal: the body exactly as it will be captured -- ahead-of-time passes, site plans -- run once eagerly, so that
                # every plan-time cache (slab tables: a launch that finds none under capture bakes the slower in-kernel descriptors into
                # the graph), every per-stream workspace and every lazily set function attribute exists before the capture
                with torch.cuda.stream(s):
                    snap = [(o, o.theta.clone(), o.m.clone(), o.v.clone(), o.step.clone()) for o in self._optimizers()]
                    rng_snap = rng.clone() if torch.is_tensor(rng) else None
                    self._iteration_body(kinds, self._ahead_prepare(kinds) if use_ahead else None)
                    for o, th, m, v, st in snap:
                        o.theta.copy_(th); o.m.copy_(m); o.v.copy_(v); o.step.copy_(st)
                    if rng_snap is not None:
                        rng.copy_(rng_snap)
                torch.cuda.current_stream(self.device).wait_stream(s)
                torch.cuda.synchronize(self.device)
                self.site_log = F.site_log()
                F.record_sites(False)
                ahead = self._ahead_prepare(kinds) if use_ahead else None
                lib.drop_taps(set(kinds))                  # (tests: only the captured graph's activations are of interest)
                dot = os.environ.get('GGAN_GRAPH_DOT')
                g = torch.cuda.CUDAGraph(keep_graph=True) if dot else torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s, capture_error_mode=_CAPTURE_MODE):
                    costs, keeps = self._iteration_body(kinds, ahead)
                if dot:
                    _dump_graph_dot(g, dot)
                self.site_mismatches = F.site_mismatches()
            finally:
                F.set_site_plan(None)
                F.record_sites(False)
            return dict(g=g, costs=costs, keep=keeps, kinds=tuple(kinds))
        finally:
            if forkable:
                self.model.fork_now = False

    def iteration(self, it, batches):
        """batches: iterator of device minibatches (or feed dicts when inject_noise); ignored in ring mode (use_ring)."""
        feed = getattr(self, 'feed', None)
        if isinstance(feed, dict) and feed.get('ring') is not None:
            from . import optim as _optim
            if _optim.STATE_EPOCH[0] != getattr(self, '_ring_epoch', _optim.STATE_EPOCH[0]):
                # the optimizers' step counters were overwritten (checkpoint.restore / load_adam_state) since the ring was anchored:
                # the device-side slot index (counters + offset) would jump; anchor it again at the slot the host-side feeder is at
                if getattr(self, '_feeder', None) is not None:
                    raise RuntimeError('optimizer state was restored while a host-fed ring is active: call use_host_ring() again')
                self.use_ring(list(feed['ring'][0]))
            kinds = (['gen'] if it > 0 else []) + ['disc'] * self.cfg.critic_iters
            feeder = getattr(self, '_feeder', None)
            if feeder is None:
                return self._iteration_ring(it, kinds)
            # host-fed ring: this iteration's slots were filled while the previous one ran; the next iteration's are issued now
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(self._ring_ready)
            res = self._iteration_ring(it, kinds)
            done = torch.cuda.Event()
            done.record(cur)
            feeder.mark_read(self._ring_pos, len(kinds), done)
            self._ring_pos += len(kinds)
            self._ring_ready = feeder.fill(self._ring_pos, 1 + self.cfg.critic_iters)
            return res
        res = {}

        def load(b):
            if isinstance(b, dict):
                self.set_feed(b)
            else:
                self.set_batch(b)
        if it > 0:
            load(next(batches))
            res['gen_cost'] = self.step('gen')
        for _ in range(self.cfg.critic_iters):
            load(next(batches))
            res['disc_cost'] = self.step('disc')
        return res

    def _iteration_ring(self, it, kinds):
        """ring mode: the steps of an iteration with nothing issued between them -- as ONE graph replay where that is possible"""
        one_graph = (self.graph_enabled and it > 0 and (self.world == 1 or self.dp_graph) and not self.split_graph
                     and all(self._calls[k] >= 1 for k in set(kinds)) and not os.environ.get('GGAN_NO_ITER_GRAPH'))
        if (not one_graph and getattr(self, '_as_captured', False) and it > 0 and (self.world == 1 or self.dp_graph) and not self.split_graph
                and all(self._calls[k] >= 1 for k in set(kinds)) and not os.environ.get('GGAN_NO_ITER_GRAPH')):
            for k in kinds:
                self._calls[k] += 1
            return self._iteration_as_captured(kinds)       # (the iteration graph's launches, issued eagerly: site plans, ahead-of-time passes)
        if not one_graph:
            return {k + '_cost': self.step(k) for k in kinds}
        for k in kinds:
            self._calls[k] += 1
        if all(self._calls[k] >= 2 for k in set(kinds)):
            lib.end_build_phase()        # (as step(): every kind has been built once)
        rec = getattr(self, '_iter_graph', None)
        if rec is None or rec['kinds'] != tuple(kinds):
            self.flush()
            rec = self._iter_graph = self._capture_iteration(kinds)
        rec['g'].replay()
        return dict(rec['costs'])

    # ---- parameters -------------------------------------------------------------------------------------
    def load_params(self, params):
        """name -> numpy array (e.g. oracle.nets.init_params) into the registry, creating entries as needed."""
        self.flush()
        with torch.no_grad():
            for name, v in params.items():
                trainable = not (name.endswith('.moving_mean') or name.endswith('.moving_variance'))
                p = lib.param(name, np.asarray(v, dtype=np.float32), trainable=trainable)
                p.data.copy_(torch.as_tensor(np.asarray(v, dtype=np.float32)).reshape(p.shape))

    def get_params(self):
        self.flush()
        return {n: p.detach().cpu().numpy().copy() for n, p in lib.named_params().items()}


def broadcast_params(src=0):
    """Replicas start from rank-`src`'s weights (the reference has a single replica; DP adds this)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for _, p in sorted(lib.named_params().items()):
            dist.broadcast(p.data, src=src)


def synthetic_ring(cfg, device, n=8, seed=1234):
    """Device-resident ring of pre-staged synthetic minibatches for cfg's default model."""
    return GraphicalGAN(cfg).synthetic_ring(device, n, seed)

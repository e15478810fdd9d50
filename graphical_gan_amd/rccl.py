"""RCCL bound directly (ctypes on librccl.so): the gradient exchange on a HIP stream WE choose.

torch.distributed's NCCL process group issues a collective on an internal stream and leaves a watchdog thread polling the
completion event of every eagerly issued one.  As soon as a communicator's stream is part of a HIP graph capture, such a poll
raises in the watchdog thread and terminates the process (round-2 finding) -- the captured exchange then only worked
behind sleeps that let the watchdog drain.  A communicator of our own has no watchdog: `ncclAllReduce(..., stream)` is an ordinary
enqueue on the given stream, captured into a step graph like any kernel launch, and nothing else in the process ever touches it.
torch.distributed (any backend) stays for what happens once, outside the graphs: rendezvous (the 128-byte unique id travels
through it), the initial parameter broadcast, the benchmark's barriers.

    comm = Communicator(rank, world, device)        # collective: every rank calls it at the same point
    comm.all_reduce_(flat_grads, stream)            # in place, sum, float32; stream: torch.cuda.Stream (e.g. the capturing one)
    comm.all_gather(out[world, n], row[n], stream)
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

NCCL_UNIQUE_ID_BYTES = 128
ncclFloat32, ncclSum = 7, 0


class ncclUniqueId(C.Structure):
    _fields_ = [('internal', C.c_char * NCCL_UNIQUE_ID_BYTES)]


class RcclError(RuntimeError):
    pass


_LIB = [None]


def _lib():
    if _LIB[0] is None:
        cands = [os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so'), '/opt/rocm/lib/librccl.so', 'librccl.so']
        err = None
        for p in cands:             # (torch's copy first: the same shared object the process already has mapped)
            try:
                L = C.CDLL(p)
                break
            except OSError as e:    # noqa: PERF203
                err = e
        else:
            raise RcclError('librccl.so not found (%s)' % err)
        L.ncclGetErrorString.restype = C.c_char_p
        L.ncclGetErrorString.argtypes = [C.c_int]
        L.ncclGetUniqueId.argtypes = [C.POINTER(ncclUniqueId)]
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, ncclUniqueId, C.c_int]
        L.ncclCommDestroy.argtypes = [C.c_void_p]
        L.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        _LIB[0] = L
    return _LIB[0]


def _check(rc, what):
    if rc != 0:
        raise RcclError('%s failed: %s' % (what, _lib().ncclGetErrorString(rc).decode()))


class Work(object):
    """what GradBucket's callers need of a work handle: wait() = the CURRENT stream waits for the exchange (the host does not)"""

    def __init__(self, stream, device):
        self.stream, self.device = stream, device

    def wait(self):
        torch.cuda.current_stream(self.device).wait_stream(self.stream)


class Communicator(object):
    def __init__(self, rank, world, device, group=None):
        """Collective over the ranks of `group` (default: the world).  The unique id is created on rank 0 and broadcast through
        torch.distributed as a byte tensor (CPU tensor on gloo, device tensor on nccl)."""
        self.rank, self.world, self.device = int(rank), int(world), torch.device(device)
        L = _lib()
        uid = ncclUniqueId()
        rc0 = 0
        if self.rank == 0:
            rc0 = L.ncclGetUniqueId(C.byref(uid))
        if self.world > 1:
            # the id travels with a success flag: a rank 0 that could not create it must not leave the others sitting in the broadcast
            # (and then walk into the next collective alone) -- every rank abandons the creation together
            on_dev = dist.get_backend(group) == 'nccl'
            payload = (list(bytes(uid.internal)) + [1 if rc0 == 0 else 0]) if self.rank == 0 else [0] * (NCCL_UNIQUE_ID_BYTES + 1)
            t = torch.tensor(payload, dtype=torch.uint8, device=self.device if on_dev else 'cpu')
            dist.broadcast(t, src=0, group=group)
            raw = bytes(t.cpu().tolist())
            if raw[NCCL_UNIQUE_ID_BYTES] != 1:
                raise RcclError('ncclGetUniqueId failed on rank 0' + (': %s' % L.ncclGetErrorString(rc0).decode() if self.rank == 0 else ''))
            C.memmove(C.byref(uid), raw[:NCCL_UNIQUE_ID_BYTES], NCCL_UNIQUE_ID_BYTES)
        else:
            _check(rc0, 'ncclGetUniqueId')
        self._comm = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(L.ncclCommInitRank(C.byref(self._comm), self.world, uid, self.rank), 'ncclCommInitRank')
        # the exchanges run on a stream of their own: forked from the issuing stream, joined by Work.wait() -- inside a capture this is
        # a parallel branch of the step graph, so the backward pass that follows the pack keeps running while the bucket is on the wire
        self.stream = torch.cuda.Stream(device=self.device)
        self.serial = False          # get_stats(): every operation goes through self.stream, an in-order queue

    def _enqueue(self, fn, stream):
        with torch.cuda.device(self.device):
            _check(fn(C.c_void_p(stream.cuda_stream)), 'rccl enqueue')

    def all_reduce_(self, t, stream=None, async_op=False):
        """in-place float32 sum over the replicas.  stream=None: on the communicator's own stream behind everything the current
        stream has issued; returns a Work when async_op, else joins the current stream before returning."""
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        L = _lib()
        cur = torch.cuda.current_stream(self.device)
        s = stream if stream is not None else self.stream
        same = s.cuda_stream == cur.cuda_stream           # (current_stream() returns a fresh object each call: compare the handles)
        if not same:
            s.wait_stream(cur)
        self._enqueue(lambda st: L.ncclAllReduce(C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr()), t.numel(), ncclFloat32, ncclSum,
                                                 self._comm, st), s)
        if same:
            return None
        w = Work(s, self.device)
        if async_op:
            return w
        w.wait()
        return None

    def all_gather(self, out, row, stream=None):
        """out[world, *row.shape] <- every replica's row, in rank order; on the current stream by default"""
        assert out.is_cuda and row.is_cuda and out.dtype == row.dtype == torch.float32 and out.is_contiguous() and row.is_contiguous()
        assert out.numel() == self.world * row.numel()
        L = _lib()
        cur = torch.cuda.current_stream(self.device)
        # serial mode (cross-replica BatchNorm): through the communicator's own stream, forked from and joined to the current one -- behind
        # whatever bucket is still on the wire there, and the next bucket behind it: ONE in-order queue for the communicator
        s = stream if stream is not None else (self.stream if self.serial else cur)
        via = s.cuda_stream != cur.cuda_stream
        if via:
            s.wait_stream(cur)
        self._enqueue(lambda st: L.ncclAllGather(C.c_void_p(row.data_ptr()), C.c_void_p(out.data_ptr()), row.numel(), ncclFloat32,
                                                 self._comm, st), s)
        if via:
            cur.wait_stream(s)
        return out

    def count(self):
        """the number of ranks as RCCL itself reports it (ncclCommCount)"""
        n = C.c_int(0)
        _check(_lib().ncclCommCount(self._comm, C.byref(n)), 'ncclCommCount')
        return int(n.value)

    def destroy(self):
        if self._comm:
            _lib().ncclCommDestroy(self._comm)
            self._comm = C.c_void_p()


_COMM = [None]
_STATS = [None]
_FAILED = [False]      # creation was tried and refused: do not retry (every retry is a collective)


def get_stats(device=None):
    """The communicator for the cross-replica BatchNorm statistics (all-gathers on the step's own stream): the SAME communicator as the
    gradient buckets', switched to serial mode -- from here on its all-gathers go through the communicator's own stream too (forked from and
    joined to the issuing stream: Communicator.all_gather), so all its operations form one in-order queue.  The buckets travel on the communicator's stream while the backward pass
    (whose BatchNorm layers gather statistics) runs on the step's: operations of one communicator must not be in flight from two
    streams, and two communicators whose kernels the ranks may start in different orders can deadlock (rounds 3-4 used a second
    communicator).  Cross-replica BatchNorm is the parity mode, not the throughput mode: the lost overlap is its price.  None where
    get() is None (the statistics then go through torch.distributed.all_gather between eager launches)."""
    base = get(device)
    if base is not None:
        base.serial = True
    _STATS[0] = base
    return base


def release_stats():
    """Leave serial mode: the communicator's all-gathers go back to the issuing stream and nothing joins the gradient buckets' stream any
    more (a Trainer without cross-replica BatchNorm, tflib.ops.batchnorm.set_sync_group(None)).  ADVICE r5: serial mode used to stick to
    the shared communicator for the rest of the process, and every later Trainer lost the bucket / backward overlap."""
    base = _STATS[0]
    if base is not None:
        base.serial = False
    _STATS[0] = None


def get(device=None, create=True):
    """The process's communicator for captured (and eager) exchanges: created on first use when torch.distributed runs on the nccl
    backend (a collective call: every rank reaches it at Trainer construction); None on other backends -- gloo stages device
    tensors through the host and the exchange stays a host-issued torch.distributed collective between cut graphs."""
    if _COMM[0] is None and create and not _FAILED[0] and dist.is_available() and dist.is_initialized() \
            and dist.get_backend() == 'nccl' and not os.environ.get('GGAN_NO_DIRECT_RCCL'):
        dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        comm, err = None, None
        try:
            comm = Communicator(dist.get_rank(), dist.get_world_size(), dev)
        except (RcclError, OSError, AttributeError) as e:       # library missing / symbol missing / init refused on this rank
            err = e
        # every replica takes the same path: one rank without a communicator sends all of them to the process-group exchange
        # (host-issued between cut graphs, engine.Trainer.split_graph) instead of leaving the others in a collective it never joins
        flag = torch.tensor([0.0 if comm is None else 1.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag[0]) < 0.5:
            if comm is not None:
                comm.destroy()
            if dist.get_rank() == 0 or err is not None:
                import sys
                sys.stderr.write('[rccl] rank %d: no direct communicator (%s): the gradient exchange goes through torch.distributed '
                                 'between cut graphs\n' % (dist.get_rank(), err if err is not None else 'another rank failed'))
            _FAILED[0] = True
            return None
        _COMM[0] = comm
    return _COMM[0]


def reset():
    _STATS[0] = None
    if _COMM[0] is not None:
        _COMM[0].destroy()
        _COMM[0] = None
    _FAILED[0] = False

"""Host -> device minibatch staging for real-data runs: a ring of pinned host buffers read by a COPY KERNEL.

The reference feeds every session.run through `feed_dict` (a synchronous host->device copy of the minibatch,
gmgan_inference_cifar10.py:480-494).  `DevicePrefetcher` wraps a tflib-style `get_epoch` generator (tflib/mnist.py,
cifar10.py, celebA.py, simple_moving_mnist.py) and yields device tensors that `engine.Trainer.iteration` consumes.

Measured on this stack (MI355X, ROCm 7.2, tools/host_feed_probe.py): `hipMemcpyAsync` from pinned memory blocks the HOST until the
stream's earlier work has drained (0.35 ms per 786 KB minibatch instead of 30 us, and the host can no longer run ahead of the
step graphs), with or without a separate copy stream.  So the transfer is done by a kernel instead: pinned host memory is
device-addressable, and `ggan_pack` (a float4 copy kernel) reads it over PCIe straight into a device slot -- an ordinary
asynchronous launch on the consumer's stream.  A pinned slot is rewritten by the host only after the kernel that read it has
finished (event query), which also bounds how far the host runs ahead."""
import time

import numpy as np
import torch


class DevicePrefetcher(object):
    def __init__(self, get_epoch, device, depth=4, pick=None, dtypes=None):
        """get_epoch: callable returning an iterator of minibatches (array or tuple of arrays); restarted forever.
        pick: indices of the tuple elements to stage (default: all).  dtypes: numpy dtype per staged element (default: keep;
        images of the uint8 loaders are fed as int32, the reference's placeholder type; elements must be 4 bytes wide).
        depth: ring size.  A yielded tensor stays valid for `depth - 1` further calls."""
        self.get_epoch, self.device, self.depth = get_epoch, torch.device(device), depth
        self.pick, self.dtypes = pick, dtypes
        self.slots = []                       # ring of {pinned, dev, read}
        self.head = 0
        self._it = None

    def _next_host(self):
        while True:
            if self._it is None:
                self._it = iter(self.get_epoch())
            try:
                b = next(self._it)
            except StopIteration:
                self._it = None
                continue
            if not isinstance(b, (tuple, list)):
                b = (b,)
            if self.pick is not None:
                b = tuple(b[i] for i in self.pick)
            if self.dtypes is not None:
                b = tuple(np.asarray(a).astype(dt, copy=False) if dt is not None else np.asarray(a) for a, dt in zip(b, self.dtypes))
            return tuple(np.ascontiguousarray(a) for a in b)

    def __iter__(self):
        return self

    def __next__(self):
        from . import functional as F
        host = self._next_host()
        if len(self.slots) <= self.head:
            tt = [torch.from_numpy(a) for a in host]
            for t in tt:
                if t.element_size() != 4:
                    raise TypeError('DevicePrefetcher stages 4-byte elements (got %s): pass dtypes=' % t.dtype)
            pinned = [torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in tt]
            self.slots.append(dict(pinned=pinned, host=[p.numpy() for p in pinned],
                                   dev=[torch.empty(t.shape, dtype=t.dtype, device=self.device) for t in tt], read=None))
        slot = self.slots[self.head]
        self.head = (self.head + 1) % self.depth
        if slot['read'] is not None:
            while not slot['read'].query():           # the copy kernel that read this pinned slot `depth` calls ago
                time.sleep(2e-5)
        for a, hb in zip(host, slot['host']):
            np.copyto(hb, a)                           # plain memcpy (torch's threaded CPU copy_ is erratic for these sizes)
        for pb, db in zip(slot['pinned'], slot['dev']):
            n = pb.numel()
            F.pack_([pb.view(torch.float32).reshape(-1)], [(0, n)], db.view(torch.float32).reshape(-1))   # bit-preserving copy
        slot['read'] = torch.cuda.Event()
        slot['read'].record(torch.cuda.current_stream(self.device))
        dev = slot['dev']
        return dev[0] if len(dev) == 1 else tuple(dev)


class RingFeeder(object):
    """Host minibatches -> the device-resident ring that Trainer.use_ring reads in place (no staging buffer, one graph per
    iteration).  fill(first, count) copies the next `count` host minibatches into ring slots first .. first+count-1 (mod R) on a
    COPY stream -- pinned staging buffer -> slot by a copy kernel reading host memory over PCIe, as DevicePrefetcher -- and
    returns the event the consuming stream waits for; before a slot is overwritten the copy stream waits for the replay that
    last read it (`done` events handed in by the Trainer).  With R >= 3 iterations' worth of slots the transfer of iteration
    i+1 runs under the replay of iteration i."""

    def __init__(self, get_epoch, device, ring, dtype=np.int32, pick=None):
        """get_epoch: callable returning an iterator of host minibatches, or a DevicePrefetcher whose host iterator is continued"""
        self.src = get_epoch if isinstance(get_epoch, DevicePrefetcher) else DevicePrefetcher(get_epoch, device, pick=pick, dtypes=[dtype])
        self.device, self.ring = torch.device(device), ring
        self.R = ring.shape[0]
        self.stream = torch.cuda.Stream(device=self.device)
        self.pinned = [torch.empty(ring[0].shape, dtype=ring.dtype).pin_memory() for _ in range(self.R)]
        self.read_done = [None] * self.R                 # copy kernel that last read pinned[i]
        self.slot_free = [None] * self.R                 # replay that last read ring[i] (set by the Trainer)

    def fill(self, first, count):
        from . import functional as F
        assert count <= self.R
        with torch.cuda.stream(self.stream):
            for j in range(count):
                i = (first + j) % self.R
                host = self.src._next_host()[0]
                if self.read_done[i] is not None:
                    self.read_done[i].synchronize()      # (the host buffer is about to be rewritten)
                np.copyto(self.pinned[i].numpy(), host.reshape(self.pinned[i].shape))
                if self.slot_free[i] is not None:
                    self.stream.wait_event(self.slot_free[i])
                n = self.pinned[i].numel()
                F.pack_([self.pinned[i].view(torch.float32).reshape(-1)], [(0, n)], self.ring[i].view(torch.float32).reshape(-1))
                self.read_done[i] = torch.cuda.Event()
                self.read_done[i].record(self.stream)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return ev

    def mark_read(self, first, count, event):
        for j in range(count):
            self.slot_free[(first + j) % self.R] = event

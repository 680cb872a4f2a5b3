"""Host <-> device transfers of whole images for the numpy-facing API (`transformImage*`, `stitchPanorama` from arrays).

The reference's callers hand over numpy arrays and expect numpy arrays back (app.py:356, 364); on this path the warp is
PCIe-bound (a 4K frame: 25 MB up, 23 MB down; config 4's stitch: 2 x 134 MB up, 250 MB down).  `Tensor.to(device)` of a
pageable array and `.cpu()` into a fresh one move ~10-13 GB/s (one thread copies through the runtime's bounce buffers
and takes the page faults of the fresh array); the link does ~55 GB/s.  Here: a small ring of page-locked staging
buffers, allocated once; several host threads copy array <-> staging chunk by chunk (numpy releases the interpreter lock
for plain copies) while the DMA engine moves the chunks already staged, on a stream of its own.

Plumbing only: nothing is computed here.  Arrays below `MIN_BYTES` take the plain torch calls."""
import threading
from collections import deque
from concurrent.futures import ThreadPoolExecutor

import numpy as np

CHUNK = 8 << 20          # bytes per staging slot
SLOTS = 6
THREADS = 6
MIN_BYTES = 4 << 20

_pipes = {}
_lock = threading.Lock()


class _Pipe(object):
    def __init__(self, dev):
        import torch
        self.slots = [torch.empty(CHUNK, dtype=torch.uint8, pin_memory=True) for _ in range(SLOTS)]
        self.views = [s.numpy() for s in self.slots]
        self.stream = torch.cuda.Stream(dev)
        self.pool = ThreadPoolExecutor(THREADS)
        self.lock = threading.Lock()


_side = {}


def side_stream(dev, name):
    """A long-lived side stream per (device, name): the pipelined stitch composes on one and downloads on another."""
    import torch
    key = (dev.type, dev.index, name)
    with _lock:
        if key not in _side:
            _side[key] = torch.cuda.Stream(dev)
        return _side[key]


def _pipe(dev):
    key = (dev.type, dev.index)
    with _lock:
        if key not in _pipes:
            _pipes[key] = _Pipe(dev)
        return _pipes[key]


def upload_tasks(tasks, dev, on_chunk=None, join=True):
    """Staged host -> device copies in the caller's order.  tasks: iterable of (src, dst, off, m, tag) -- `m` bytes at byte offset `off`
    of the flat uint8 numpy array `src` to the same offset of the flat uint8 device tensor `dst` (m <= CHUNK).  Host threads fill the
    page-locked ring, the DMA engine drains it on the pipe's own stream.  on_chunk(tag, off + m, event) runs on the calling thread
    right after a chunk's DMA has been enqueued (`event` fires when it has landed).  join: make torch's current stream wait for
    the whole upload."""
    import torch
    p = _pipe(dev)
    with p.lock:
        cur = torch.cuda.current_stream(dev)
        p.stream.wait_stream(cur)                                  # the destinations belong to the current stream's allocator history
        events = [None] * SLOTS
        staged = deque()
        with torch.cuda.stream(p.stream):
            def issue():
                i, dst, off, m, tag, fut = staged.popleft()
                fut.result()
                dst[off:off + m].copy_(p.slots[i][:m], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(p.stream)
                events[i] = ev
                if on_chunk is not None:
                    on_chunk(tag, off + m, ev)
            for c, (src, dst, off, m, tag) in enumerate(tasks):
                i = c % SLOTS
                while any(s_[0] == i for s_ in staged):            # the slot still waits for its own DMA to be issued
                    issue()
                if events[i] is not None:
                    events[i].synchronize()                        # its previous DMA has read the slot
                staged.append((i, dst, off, m, tag, p.pool.submit(np.copyto, p.views[i][:m], src[off:off + m])))
                while len(staged) > SLOTS - 2:
                    issue()
            while staged:
                issue()
        if join:
            cur.wait_stream(p.stream)
        for ev in events:                                          # the slots are free again when this call returns
            if ev is not None:
                ev.synchronize()


def to_device(arr, dev):
    """contiguous numpy array -> tensor of the same shape / dtype on `dev`, usable on torch's current stream."""
    import torch
    a = np.ascontiguousarray(arr)
    if a.nbytes < MIN_BYTES:
        return torch.from_numpy(a).to(dev)
    flat = a.reshape(-1).view(np.uint8)
    n = flat.size
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    upload_tasks(((flat, dst, off, min(CHUNK, n - off), None) for off in range(0, n, CHUNK)), dev)
    return dst.view(torch.from_numpy(a[:0].reshape(-1)).dtype).reshape(a.shape)


PINNED_RESULT_MAX = 1 << 30     # results up to this size come back in page-locked memory of torch's caching host allocator


def to_host(t):
    """device tensor -> numpy array of the same shape / dtype (the producer kernels are on torch's current stream).

    The array lives in a page-locked block of torch's caching host allocator and the DMA engine writes straight into it:
    no staging copy, no page faults of a fresh pageable array (together 3x the DMA time for a 250 MB canvas).  When the
    caller drops the array the block goes back to that cache, so a loop that replaces its previous result -- what app.py
    does with g_ctx.oimg -- re-uses two blocks and pins nothing after its second call.  If page-locked memory cannot be had
    the result is an ordinary array filled through the staging ring (`_to_host_staged`)."""
    import torch
    t = t.contiguous()
    nbytes = t.numel() * t.element_size()
    if nbytes < MIN_BYTES or not t.is_cuda:
        return t.cpu().numpy()
    if nbytes <= PINNED_RESULT_MAX:
        try:
            host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            host.copy_(t, non_blocking=True)
            torch.cuda.current_stream(t.device).synchronize()
            return host.numpy()                                    # shares the block; the array keeps the tensor alive
        except RuntimeError:
            pass
    return _to_host_staged(t)


def _to_host_staged(t):
    """device tensor -> new pageable numpy array, through the staging ring (DMA and host copies overlapped)."""
    import torch
    nbytes = t.numel() * t.element_size()
    dev = t.device
    flat = t.reshape(-1).view(torch.uint8)
    out = np.empty(nbytes, dtype=np.uint8)
    p = _pipe(dev)
    with p.lock:
        p.stream.wait_stream(torch.cuda.current_stream(dev))
        busy = [None] * SLOTS                                      # per slot: the host copy that still reads it
        pending = []

        def drain(ev, i, off, m):
            ev.synchronize()
            np.copyto(out[off:off + m], p.views[i][:m])

        with torch.cuda.stream(p.stream):
            for c, off in enumerate(range(0, nbytes, CHUNK)):
                i = c % SLOTS
                if busy[i] is not None:
                    busy[i].result()
                m = min(CHUNK, nbytes - off)
                p.slots[i][:m].copy_(flat[off:off + m], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(p.stream)
                busy[i] = p.pool.submit(drain, ev, i, off, m)
                pending.append(busy[i])
        for f in pending:
            f.result()
        torch.cuda.current_stream(dev).wait_stream(p.stream)       # `t` may be freed / reused by the caller from here on
    return out.view(torch.empty(0, dtype=t.dtype).numpy().dtype).reshape(tuple(t.shape))

"""Device-level operators: torch CUDA tensors in, torch CUDA tensors out, one C-ABI call each.

torch is used for device memory and streams only; all arithmetic happens in libic_hip.so.  The
numpy-in / numpy-out mirrors of the reference's call sites (vq.py, features.py, classifiers.py,
robust_matching.py, scene_functions.py) are thin layers over these functions.
"""
import collections
import ctypes
import os
import threading

import numpy as np
import torch

from . import _lib

TOP2_REC_DTYPE = np.dtype([("d1", "<f4"), ("i1", "<i4"), ("d2", "<f4"), ("i2", "<i4")])
# The matrix-core search contracts over slabs of 128 columns: rows of up to 128 columns (SURF-128 /
# SIFT, narrower ones zero-padded) are the tuned case; wider rows, up to this limit, carry the
# accumulators from slab to slab (the 1937-column face signatures of
# vision/facerec/face_indexing.py:25 are quantised that way).
MAX_DESCRIPTOR_DIM = 2048


def check_descriptor_dim(d, who):
    """Rows wider than MAX_DESCRIPTOR_DIM are refused here, before anything is launched."""
    if int(d) > MAX_DESCRIPTOR_DIM:
        raise ValueError("%s: rows have %d columns, the descriptor search handles at most %d "
                         "(use ops.knn_rows / ops.pairwise for wider rows)"
                         % (who, int(d), MAX_DESCRIPTOR_DIM))


def _dev_index(t):
    if not t.is_cuda:
        raise ValueError("expected a CUDA tensor (there is no CPU path)")
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


def _stream_ptr(device_index):
    return ctypes.c_void_p(torch.cuda.current_stream(device_index).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _check_2d_f32(t, name):
    if t.dim() != 2:
        raise ValueError("%s must be 2-D, got shape %s" % (name, tuple(t.shape)))
    if t.dtype != torch.float32:
        raise ValueError("%s must be float32 on the device, got %s" % (name, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be C-contiguous" % name)


def as_device_f32(a, device):
    """numpy (any float dtype) or torch tensor -> contiguous float32 CUDA tensor on `device`."""
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float32).contiguous()
    a = np.ascontiguousarray(a, dtype=np.float32)
    return _from_numpy(a).to(device)


def _from_numpy(a):
    """torch view of a host array that is only read (read-only memory maps included)."""
    if a.flags.writeable:
        return torch.from_numpy(a)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)
        return torch.from_numpy(a)


def row_sqnorm(x):
    _check_2d_f32(x, "x")
    dev = _dev_index(x)
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    rc = _lib.load().ic_row_sqnorm(_lib.handle(dev), _ptr(x), x.shape[0], x.shape[1], _ptr(out),
                                   _stream_ptr(dev))
    _lib.check(rc, "ic_row_sqnorm")
    return out


def codebook_aux(c):
    """The auxiliary block of a codebook (float32 device tensor): its mean vector, the squared
    norms about that mean and the two maxima of the search kernel's rounding model.  Computed once
    per codebook and passed to the assign_* functions as `c_aux`."""
    _check_2d_f32(c, "codebook")
    check_descriptor_dim(c.shape[1], "codebook_aux")
    dev = _dev_index(c)
    lib = _lib.load()
    aux = torch.empty(int(lib.ic_codebook_aux_floats_d(c.shape[0], c.shape[1])),
                      dtype=torch.float32, device=c.device)
    rc = lib.ic_codebook_aux(_lib.handle(dev), _ptr(c), c.shape[0], c.shape[1], _ptr(aux),
                             _stream_ptr(dev))
    _lib.check(rc, "ic_codebook_aux")
    return aux


def debug_flagged(device_index, counter):
    """Test hook: `counter` (uint32/int32 device tensor with one element, or None) receives the
    number of rows every following search on this device hands to the direct second look."""
    rc = _lib.load().ic_debug_flagged(_lib.handle(device_index), _ptr(counter))
    _lib.check(rc, "ic_debug_flagged")


def _check_pair(x, c, descriptor_search=True):
    _check_2d_f32(x, "x")
    _check_2d_f32(c, "codebook")
    if x.shape[1] != c.shape[1]:
        raise ValueError("dimension mismatch: x has %d columns, codebook %d" % (x.shape[1], c.shape[1]))
    if descriptor_search:
        check_descriptor_dim(x.shape[1], "descriptor search")
    if c.shape[0] < 1:
        raise ValueError("empty codebook")
    if x.device != c.device:
        raise ValueError("x and codebook live on different devices")


def _check_aux(c, c_aux):
    """The library reads the auxiliary block by the layout that belongs to the codebook's shape
    (a copy of the codebook in scan order included): a block of any other size is refused here
    instead of being read out of bounds on the device."""
    if c_aux is None:
        return
    want = int(_lib.load().ic_codebook_aux_floats_d(c.shape[0], c.shape[1]))
    if (c_aux.dtype != torch.float32 or c_aux.dim() != 1 or c_aux.numel() != want
            or c_aux.device != c.device or not c_aux.is_contiguous()):
        raise ValueError("c_aux must be the float32 block of %d elements that codebook_aux() "
                         "returns for this codebook, on its device" % want)


def assign_top1(x, c, c_aux=None, want_dist=True, out=None, hint=None):
    """idx[i] = argmin_j |x_i - c_j| (lowest j on ties), dist[i] = that Euclidean distance.
    `out` = (idx, dist) preallocated contiguous device tensors (views are fine).
    `hint` = int32 [n] guess of idx (e.g. the labels of the previous k-means iteration): same
    result, less ranking work when the guess is good."""
    _check_pair(x, c)
    _check_aux(c, c_aux)
    dev = _dev_index(x)
    n = x.shape[0]
    if out is not None:
        idx, dist = out
    else:
        idx = torch.empty(n, dtype=torch.int32, device=x.device)
        dist = torch.empty(n, dtype=torch.float32, device=x.device) if want_dist else None
    if hint is not None:
        if hint.dtype != torch.int32 or hint.shape[0] != n or hint.device != x.device:
            raise ValueError("hint must be an int32 device tensor with one entry per row of x")
        rc = _lib.load().ic_assign_top1_hint(
            _lib.handle(dev), _ptr(x), n, _ptr(c), c.shape[0], x.shape[1], _ptr(c_aux),
            _ptr(hint.contiguous()), _ptr(idx), _ptr(dist), _stream_ptr(dev))
        _lib.check(rc, "ic_assign_top1_hint")
        return idx, dist
    rc = _lib.load().ic_assign_top1(_lib.handle(dev), _ptr(x), n, _ptr(c), c.shape[0], x.shape[1],
                                    _ptr(c_aux), _ptr(idx), _ptr(dist), _stream_ptr(dev))
    _lib.check(rc, "ic_assign_top1")
    return idx, dist


def assign_top1_static(x, c, c_aux, out):
    """assign_top1 through the entry point that allocates nothing (static device scratch, launches
    only).  `out` = (idx int32 [n], dist float32 [n] or None), preallocated by the caller.  Raises
    when the input needs more scratch than the static buffer holds."""
    _check_pair(x, c)
    _check_aux(c, c_aux)
    dev = _dev_index(x)
    idx, dist = out
    rc = _lib.load().ic_assign_top1_static(_lib.handle(dev), _ptr(x), x.shape[0], _ptr(c),
                                           c.shape[0], x.shape[1], _ptr(c_aux), _ptr(idx),
                                           _ptr(dist), _stream_ptr(dev))
    _lib.check(rc, "ic_assign_top1_static")
    return idx, dist


def assign_top2(x, c, ratio=0.8, c_aux=None, want_mask=True, out=None):
    """Two nearest rows of c per row of x, ascending (distance, index), and the ratio-test mask.
    `out` = (i1, d1, i2, d2, mask) preallocated contiguous device tensors."""
    _check_pair(x, c)
    _check_aux(c, c_aux)
    dev = _dev_index(x)
    n = x.shape[0]
    if out is not None:
        i1, d1, i2, d2, mask = out
    else:
        i1 = torch.empty(n, dtype=torch.int32, device=x.device)
        i2 = torch.empty(n, dtype=torch.int32, device=x.device)
        d1 = torch.empty(n, dtype=torch.float32, device=x.device)
        d2 = torch.empty(n, dtype=torch.float32, device=x.device)
        mask = torch.empty(n, dtype=torch.uint8, device=x.device) if want_mask else None
    rc = _lib.load().ic_assign_top2(_lib.handle(dev), _ptr(x), n, _ptr(c), c.shape[0], x.shape[1],
                                    _ptr(c_aux), _ptr(i1), _ptr(d1), _ptr(i2), _ptr(d2),
                                    float(ratio), _ptr(mask), _stream_ptr(dev))
    _lib.check(rc, "ic_assign_top2")
    return i1, d1, i2, d2, mask


def assign_top2_exact(x, c, ratio=0.8):
    """assign_top2 by direct float32 differences (no norm expansion): the slow on-device checker."""
    _check_pair(x, c)
    dev = _dev_index(x)
    n = x.shape[0]
    i1 = torch.empty(n, dtype=torch.int32, device=x.device)
    i2 = torch.empty(n, dtype=torch.int32, device=x.device)
    d1 = torch.empty(n, dtype=torch.float32, device=x.device)
    d2 = torch.empty(n, dtype=torch.float32, device=x.device)
    mask = torch.empty(n, dtype=torch.uint8, device=x.device)
    rc = _lib.load().ic_assign_top2_exact(_lib.handle(dev), _ptr(x), n, _ptr(c), c.shape[0],
                                          x.shape[1], _ptr(i1), _ptr(d1), _ptr(i2), _ptr(d2),
                                          float(ratio), _ptr(mask), _stream_ptr(dev))
    _lib.check(rc, "ic_assign_top2_exact")
    return i1, d1, i2, d2, mask


_copy_streams = {}
HOST_CHUNK_ROWS = 1 << 17   # rows per upload chunk of the streamed host path (64 MiB of float32)


def _copy_stream(device):
    key = (os.getpid(), str(device))
    s = _copy_streams.get(key)
    if s is None:
        s = _copy_streams[key] = torch.cuda.Stream(device=device)
    return s


def assign_from_host(x_np, c, c_aux=None, top2=False, ratio=0.8, want_dist=True,
                     chunk_rows=None):
    """Search rows that live in HOST memory (numpy [n, d]): the one-array case of
    assign_from_host_parts, through the same page-locked upload pipeline (an upload straight from
    pageable memory runs at less than half the link's rate).
    Returns what assign_top1 / assign_top2 return (device tensors covering all rows).
    """
    return assign_from_host_parts([x_np], c, c_aux, top2, ratio, want_dist, chunk_rows)


_pinned = {}     # (pid, rows, d) -> page-locked upload buffers; (pid, "hist_down") -> download buffers
_staging_lock = threading.Lock()   # the staging buffers are shared by all threads of a process
_gather_pool = None
GATHER_THREADS = 8
UPLOAD_CHUNK_BYTES = 32 << 20    # rows travel to the device in pieces of this size
SEGMENT_CHUNKS = 4                # uploads that share one device buffer, the most one search takes


def _pinned_set(rows, d):
    """Three page-locked upload buffers of [rows, d] floats: one being gathered into, one whose
    gather is queued behind it, one on its way to the device."""
    key = (os.getpid(), rows, d)
    bufs = _pinned.get(key)
    if bufs is None:
        for k in [k for k in _pinned if k[0] == key[0] and len(k) == 3]:   # one upload size per process
            del _pinned[k]
        bufs = _pinned[key] = [torch.empty((rows, d), dtype=torch.float32).pin_memory()
                               for _ in range(3)]
    return bufs


def _pool():
    global _gather_pool
    if _gather_pool is None or getattr(_gather_pool, "_pid", None) != os.getpid():
        from concurrent.futures import ThreadPoolExecutor
        _gather_pool = ThreadPoolExecutor(max_workers=GATHER_THREADS)
        _gather_pool._pid = os.getpid()
    return _gather_pool


_gatherer = None
_F64 = np.dtype(np.float64)


def _gather_thread():
    """The one Python thread that runs the native gathers, so that the calling thread can go on
    enqueueing uploads and searches meanwhile (the copying itself happens on GATHER_THREADS native
    threads that never need the interpreter lock)."""
    global _gatherer
    if _gatherer is None or getattr(_gatherer, "_pid", None) != os.getpid():
        from concurrent.futures import ThreadPoolExecutor
        _gatherer = ThreadPoolExecutor(max_workers=1)
        _gatherer._pid = os.getpid()
    return _gatherer


def _gather_rows(dst, jobs, d, threads=None):
    """dst (page-locked float32 tensor) <- the row ranges (array, lo, hi) of `jobs`, one after the
    other; the arrays are C-contiguous float32 or float64."""
    n = len(jobs)
    src = np.empty(n, dtype=np.uintp)
    elems = np.empty(n, dtype=np.int64)
    kind = np.empty(n, dtype=np.int32)
    for j, (p, lo, hi) in enumerate(jobs):
        src[j] = p.__array_interface__["data"][0] + lo * p.strides[0]
        elems[j] = (hi - lo) * d
        kind[j] = p.dtype == _F64
    rc = _lib.load().ic_host_gather_f32(dst.data_ptr(), src.ctypes.data, elems.ctypes.data,
                                        kind.ctypes.data, n, threads or GATHER_THREADS)
    _lib.check(rc, "ic_host_gather_f32")


_small_stage = {}                 # pid -> [page-locked buffer, event of its last upload]
SMALL_STAGE_BYTES = 16 << 20


def upload_parts(parts, device, extra_i64=None):
    """float32 device tensor [sum n_i, d] of a short list of host arrays (the candidates of one
    RobustMatcher.match call, the images of one small extract call): the rows are laid into a
    page-locked buffer by the native gather and leave it in one asynchronous copy on the current
    stream.  np.concatenate followed by a copy from pageable memory costs the caller about twice
    the link time of the rows (0.17 ms for 5 MB), during which it cannot enqueue anything.

    extra_i64 (the images' row offsets): a short int64 array that rides behind the rows in the
    same copy -- a separate upload of a few bytes from pageable memory costs a per-image call
    20 to 30 us; the result is then (rows, int64 device tensor)."""
    d = parts[0].shape[1]
    n = 0
    own = False
    for i, p in enumerate(parts):
        if p.ndim != 2 or p.shape[1] != d:
            raise ValueError("dimension mismatch among the parts: %s vs %d columns" % (p.shape, d))
        n += p.shape[0]
        if not p.flags.c_contiguous or p.dtype.char not in "fd":
            if not own:
                parts, own = list(parts), True
            parts[i] = np.ascontiguousarray(p, dtype=np.float32)
    if extra_i64 is not None:
        extra_i64 = np.ascontiguousarray(extra_i64, dtype=np.int64).reshape(-1)
    n_extra = 0 if extra_i64 is None else 2 * extra_i64.size      # in float32 slots
    if (n * d + 1 + n_extra) * 4 > SMALL_STAGE_BYTES or n == 0:
        x_np = parts[0] if len(parts) == 1 else np.concatenate(parts)
        xd = as_device_f32(x_np, device)
        if extra_i64 is None:
            return xd
        return xd, torch.from_numpy(extra_i64).to(device)
    with _staging_lock:
        st = _small_stage.get(os.getpid())
        if st is None:
            # (a buffer inherited from a parent process is left alone: it is not this process's
            # to free)
            st = _small_stage[os.getpid()] = [
                torch.empty(SMALL_STAGE_BYTES // 4, dtype=torch.float32).pin_memory(), None]
        if st[1] is not None:
            st[1].synchronize()           # the previous upload has left the buffer
        _gather_rows(st[0], [(p, 0, p.shape[0]) for p in parts if p.shape[0]], d, threads=2)
        at = (n * d + 1) // 2 * 2                  # the int64 payload starts on 8 bytes
        if n_extra:
            st[0][at:at + n_extra].view(torch.int64).numpy()[:] = extra_i64
        buf = torch.empty(at + n_extra, dtype=torch.float32, device=device)
        buf.copy_(st[0][:at + n_extra], non_blocking=True)
        st[1] = torch.cuda.Event()
        st[1].record(torch.cuda.current_stream(device))
    xd = buf[:n * d].view(n, d)
    if extra_i64 is None:
        return xd
    return xd, buf[at:at + n_extra].view(torch.int64)


def assign_from_host_parts(parts, c, c_aux=None, top2=False, ratio=0.8, want_dist=True,
                           chunk_rows=None):
    with _staging_lock:
        return _assign_from_host_parts(parts, c, c_aux, top2, ratio, want_dist, chunk_rows)


def _assign_from_host_parts(parts, c, c_aux, top2, ratio, want_dist, chunk_rows,
                            after_chunk=None):
    """Search rows held as a LIST of host arrays (one per image, any float dtype, [n_i, d]).

    The reference concatenates such lists on the host before every call; here worker threads
    gather the pieces straight into one of two page-locked staging buffers (converting to float32
    on the way) and the staging buffer goes to the device on a second stream, in upload chunks of
    UPLOAD_CHUNK_BYTES (`chunk_rows` rows when given).  Searches are cut independently of the
    uploads: rows that have arrived are searched at once while fewer than two searches are
    outstanding on the device, otherwise they wait and the next search takes them together with
    what has arrived since (up to SEGMENT_CHUNKS uploads, which share one device buffer).  The
    device therefore starts on the first small chunk, never idles while rows are available, and
    once it falls behind the host it is handed few, large searches instead of many small ones.
    Returns what assign_top1 / assign_top2 return (device tensors covering all rows in order).
    `after_chunk(outs, row_hi)`, when given, is called once the search of rows [0, row_hi) has
    been enqueued on the current stream (what consumes the codes block by block hooks in here).
    """
    d = c.shape[1]
    device = c.device
    n, own = 0, False
    for i, p in enumerate(parts):
        if p.ndim != 2 or p.shape[1] != d:
            raise ValueError("dimension mismatch: a part has shape %s, codebook %d columns"
                             % (p.shape, d))
        n += p.shape[0]
        if not p.flags.c_contiguous or p.dtype.char not in "fd":
            # the native gather reads packed float32 / float64 rows; anything else is converted
            if not own:
                parts, own = list(parts), True      # the caller's list stays as it is
            parts[i] = np.ascontiguousarray(p, dtype=np.float32)
    if top2:
        outs = (torch.empty(n, dtype=torch.int32, device=device),
                torch.empty(n, dtype=torch.float32, device=device),
                torch.empty(n, dtype=torch.int32, device=device),
                torch.empty(n, dtype=torch.float32, device=device),
                torch.empty(n, dtype=torch.uint8, device=device))
    else:
        outs = (torch.empty(n, dtype=torch.int32, device=device),
                torch.empty(n, dtype=torch.float32, device=device) if want_dist else None)
    if n == 0:
        return outs
    up_rows = int(chunk_rows) if chunk_rows else max(1024, UPLOAD_CHUNK_BYTES // (4 * d))
    seg_rows = up_rows * SEGMENT_CHUNKS

    def upload_chunks():
        # (copy jobs, rows) per upload; the first two are a quarter and a half of the full size,
        # so that the device has rows to search as early as possible
        want = [max(1, up_rows // 4), max(1, up_rows // 2)]
        jobs, rows = [], 0
        for p in parts:
            lo = 0
            while lo < p.shape[0]:
                size = want[0] if want else up_rows
                take = min(p.shape[0] - lo, size - rows)
                jobs.append((p, lo, lo + take))
                rows += take
                lo += take
                if rows == size:
                    yield jobs, rows
                    jobs, rows = [], 0
                    if want:
                        want.pop(0)
        if rows:
            yield jobs, rows

    stage = _pinned_set(up_rows, d)
    stage_np = [t.numpy() for t in stage]
    free = [None] * len(stage)     # event after which a staging buffer may be overwritten
    gatherer = _gather_thread()
    cur = torch.cuda.current_stream(device)
    side = _copy_stream(device)
    side.wait_stream(cur)
    outstanding = collections.deque()      # events of the searches enqueued and not yet seen done
    seg = None                             # device buffer of the current segment
    seg_row0 = seg_fill = 0                # its first global row, rows uploaded into it so far
    searched = 0                           # global rows whose search has been enqueued
    ready = None

    def search(upto):
        # rows [searched, upto) of the current segment
        nonlocal searched
        cur.wait_event(ready)
        xd = seg[searched - seg_row0:upto - seg_row0]
        view = tuple(o[searched:upto] if o is not None else None for o in outs)
        if top2:
            assign_top2(xd, c, ratio, c_aux, out=view)
        else:
            assign_top1(xd, c, c_aux, want_dist, out=view)
        done = torch.cuda.Event()
        done.record(cur)
        outstanding.append(done)
        searched = upto
        if after_chunk is not None:
            after_chunk(outs, searched)

    gen = upload_chunks()
    gathering = collections.deque()        # (future, rows, staging buffer) of gathers under way
    issued = 0

    def gather_next():
        # hands the next upload chunk to the gather thread, which is thereby never without work
        # while this thread enqueues the upload and the search of the chunk before
        nonlocal issued
        nxt = next(gen, None)
        if nxt is None:
            return
        jobs, rows = nxt
        b = issued % len(stage)
        issued += 1
        if free[b] is not None:
            free[b].synchronize()
        gathering.append((gatherer.submit(_gather_rows, stage[b], jobs, d), rows, b))

    gather_next()
    gather_next()
    while gathering:
        future, rows, b = gathering.popleft()
        future.result()
        gather_next()
        with torch.cuda.stream(side):
            if seg is None:
                seg_row0, seg_fill = seg_row0 + seg_fill, 0
                seg = torch.empty((min(seg_rows, n - seg_row0), d), dtype=torch.float32,
                                  device=device)
                seg.record_stream(cur)
            seg[seg_fill:seg_fill + rows].copy_(stage[b][:rows], non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(side)
        free[b] = ready
        seg_fill += rows
        while outstanding and outstanding[0].query():
            outstanding.popleft()
        arrived = seg_row0 + seg_fill
        if arrived == n:
            # the end: what consumes the last search's output waits for all of it, so the rows
            # still unsearched go in halves
            while arrived - searched > max(up_rows // 2, 8192):
                search(searched + (arrived - searched + 1) // 2)
            search(arrived)
        elif seg_fill + up_rows > seg.shape[0] or len(outstanding) < 2:
            search(arrived)
        if seg_fill + up_rows > seg.shape[0]:
            seg = None                     # the next upload opens a new segment
    return outs


TO_HOST_CHUNK_BYTES = 16 << 20


def _touch_pages(buf):
    buf[::4096] = 0


def quantize_parts_to_host(parts, offsets_np, c, c_aux, k, norm="l1", power=0.5, chunk_rows=None):
    """Codes and per-image histograms of images whose descriptors are a list of host arrays.

    `parts` are the non-empty descriptor arrays in image order, `offsets_np` int64 [n_img + 1]
    the row range of every image (empty images repeat an offset).  Returns (codes int32 device
    tensor [total], histograms numpy float64 [n_img, k]).

    One pipeline instead of three phases: while chunk i + 1 of the rows is gathered into
    page-locked memory, uploaded and searched, the histograms of the images chunk i completed are
    built (ic_hist into a row block of the device result) and that block travels back on a third
    stream through its own page-locked buffers; worker threads move it into the final array,
    whose pages other worker threads have touched while the first chunk was on its way (a fresh
    76 MB array otherwise costs about 15 ms of page faults inside those copies).  The link
    carries rows up and histograms down at the same time, and the download no longer follows the
    search but hides behind it.
    """
    offsets_np = np.ascontiguousarray(offsets_np, dtype=np.int64)
    n_img = offsets_np.shape[0] - 1
    device = c.device
    k = int(k)
    out = np.empty((max(n_img, 0), k), dtype=np.float64)
    if n_img <= 0:
        return torch.empty(0, dtype=torch.int32, device=device), out
    out_bytes = out.reshape(-1).view(np.uint8)
    row_bytes = k * 8
    with _staging_lock:
        pool = _pool()
        slab = -(-out_bytes.shape[0] // GATHER_THREADS)
        touching = [pool.submit(_touch_pages, out_bytes[o:o + slab])
                    for o in range(0, out_bytes.shape[0], max(slab, 1))]
        offsets_dev = torch.from_numpy(offsets_np).to(device)
        hist_dev = torch.empty((n_img, k), dtype=torch.float64, device=device)
        hist_flat = hist_dev.view(-1).view(torch.uint8)
        key = (os.getpid(), "hist_down")
        stage = _pinned.get(key)
        if stage is None:
            stage = _pinned[key] = [torch.empty(TO_HOST_CHUNK_BYTES, dtype=torch.uint8).pin_memory()
                                    for _ in range(2)]
        stage_np = [s_.numpy() for s_ in stage]
        down = _down_stream(device)
        from concurrent.futures import ThreadPoolExecutor
        courier = ThreadPoolExecutor(max_workers=1)   # takes the blocks down, one after the other
        state = {"img_done": 0, "buf": 0, "pending": [None, None], "jobs": []}
        rows_per_piece = max(1, TO_HOST_CHUNK_BYTES // row_bytes)

        def take_down(ev_hist, img_lo, img_hi):
            # courier thread: device block -> page-locked buffer -> (worker threads) final array
            for f in touching:
                f.result()
            with torch.cuda.device(device):
                for r0 in range(img_lo, img_hi, rows_per_piece):
                    r1 = min(img_hi, r0 + rows_per_piece)
                    lo, hi = r0 * row_bytes, r1 * row_bytes
                    b = state["buf"]
                    state["buf"] ^= 1
                    if state["pending"][b] is not None:
                        for f in state["pending"][b]:
                            f.result()              # the copies out of this buffer are done
                    with torch.cuda.stream(down):
                        down.wait_event(ev_hist)
                        stage[b][:hi - lo].copy_(hist_flat[lo:hi], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(down)
                    ev.synchronize()
                    step = max(1 << 20, -(-(hi - lo) // 4))
                    state["pending"][b] = [
                        pool.submit(np.copyto, out_bytes[lo + o:min(hi, lo + o + step)],
                                    stage_np[b][o:min(hi - lo, o + step)])
                        for o in range(0, hi - lo, step)]

        def after_chunk(outs, row_hi):
            # images whose last row has now been searched
            img_hi = int(np.searchsorted(offsets_np, row_hi, side="right")) - 1
            img_lo = state["img_done"]
            if img_hi <= img_lo:
                return
            hist(outs[0], offsets_dev[img_lo:img_hi + 1], k, norm, power,
                 out=hist_dev[img_lo:img_hi])
            ev_hist = torch.cuda.Event()
            ev_hist.record(torch.cuda.current_stream(device))
            state["jobs"].append(courier.submit(take_down, ev_hist, img_lo, img_hi))
            state["img_done"] = img_hi

        try:
            outs = _assign_from_host_parts(parts, c, c_aux, False, 0.8, False, chunk_rows,
                                           after_chunk=after_chunk)
            after_chunk(outs, int(offsets_np[-1]))          # trailing images without descriptors
            if state["img_done"] < n_img:
                raise ValueError("offsets do not cover the descriptors: %d rows, last offset %d"
                                 % (outs[0].shape[0], int(offsets_np[-1])))
            for j in state["jobs"]:
                j.result()
            for fs in state["pending"]:
                for f in fs or ():
                    f.result()
        finally:
            courier.shutdown(wait=True)
        hist_dev.record_stream(down)
    return outs[0], out


_down_streams = {}


def _down_stream(device):
    key = (os.getpid(), str(device))
    s = _down_streams.get(key)
    if s is None:
        s = _down_streams[key] = torch.cuda.Stream(device=device)
    return s


def to_host(t):
    """Device tensor -> numpy array.  Large results (histogram blocks of tens of MB) come back
    through two page-locked staging buffers with the copy into the final array done by worker
    threads while the next piece is in flight; a plain pageable download runs at a fraction of
    the link rate."""
    nbytes = t.numel() * t.element_size()
    if nbytes < 2 * TO_HOST_CHUNK_BYTES or not t.is_cuda:
        return t.cpu().numpy()
    with _staging_lock:
        return _to_host_staged(t, nbytes)


def _to_host_staged(t, nbytes):
    t = t.contiguous()
    flat = t.view(-1).view(torch.uint8)
    out = np.empty(tuple(t.shape), dtype=torch.empty(0, dtype=t.dtype).numpy().dtype)
    out_bytes = out.reshape(-1).view(np.uint8)
    key = (os.getpid(), "to_host")
    stage = _pinned.get(key)
    if stage is None:
        stage = _pinned[key] = [torch.empty(TO_HOST_CHUNK_BYTES, dtype=torch.uint8).pin_memory()
                                for _ in range(2)]
    stage_np = [s_.numpy() for s_ in stage]
    pool = _pool()
    side = _copy_stream(t.device)
    side.wait_stream(torch.cuda.current_stream(t.device))
    pending = [None, None]     # (event, lo, hi, future of the previous host copy)
    copies = []
    chunks = [(lo, min(nbytes, lo + TO_HOST_CHUNK_BYTES)) for lo in range(0, nbytes, TO_HOST_CHUNK_BYTES)]
    for ci, (lo, hi) in enumerate(chunks):
        b = ci & 1
        if pending[b] is not None:
            pending[b].result()                      # the host copy out of this buffer is done
        with torch.cuda.stream(side):
            stage[b][:hi - lo].copy_(flat[lo:hi], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)

        def drain(ev=ev, b=b, lo=lo, hi=hi):
            ev.synchronize()
            np.copyto(out_bytes[lo:hi], stage_np[b][:hi - lo])

        pending[b] = pool.submit(drain)
        copies.append(pending[b])
    for f in copies:
        f.result()
    return out


def assign_top2_rec(x, c, c_aux=None):
    """Same search, packed as [n, 4] int32 words = ic_top2_rec {d1, i1, d2, i2} per row."""
    _check_pair(x, c)
    _check_aux(c, c_aux)
    dev = _dev_index(x)
    n = x.shape[0]
    rec = torch.empty((n, 4), dtype=torch.int32, device=x.device)
    rc = _lib.load().ic_assign_top2_rec(_lib.handle(dev), _ptr(x), n, _ptr(c), c.shape[0],
                                        x.shape[1], _ptr(c_aux), _ptr(rec), _stream_ptr(dev))
    _lib.check(rc, "ic_assign_top2_rec")
    return rec


def top2_merge(gathered, shard_base, ratio=0.8, want_mask=True):
    """gathered: [n_shards, n, 4] int32 (ic_top2_rec words); shard_base: sequence of ints -- or
    None when gathered is [n_shards, n + 1, 4] and the last record of every shard carries its
    base as an int64 (see `pack_records`)."""
    if gathered.dim() != 3 or gathered.shape[2] != 4 or gathered.dtype != torch.int32:
        raise ValueError("gathered must be int32 [n_shards, n, 4]")
    if not gathered.is_contiguous():
        raise ValueError("gathered must be contiguous")
    dev = _dev_index(gathered)
    n_shards, n = gathered.shape[0], gathered.shape[1]
    if shard_base is None:
        n -= 1
        base = None
    else:
        if len(shard_base) != n_shards:
            raise ValueError("need one shard base per shard")
        base = (ctypes.c_int64 * n_shards)(*[int(b) for b in shard_base])
    i1 = torch.empty(n, dtype=torch.int64, device=gathered.device)
    i2 = torch.empty(n, dtype=torch.int64, device=gathered.device)
    d1 = torch.empty(n, dtype=torch.float32, device=gathered.device)
    d2 = torch.empty(n, dtype=torch.float32, device=gathered.device)
    mask = torch.empty(n, dtype=torch.uint8, device=gathered.device) if want_mask else None
    rc = _lib.load().ic_top2_merge(_lib.handle(dev), _ptr(gathered), n_shards, n, base, _ptr(i1),
                                   _ptr(d1), _ptr(i2), _ptr(d2), float(ratio), _ptr(mask),
                                   _stream_ptr(dev))
    _lib.check(rc, "ic_top2_merge")
    return i1, d1, i2, d2, mask


def hist(codes, offsets, k, norm="l1", power=0.5, out=None):
    """codes int32 [total], offsets int64 [n_img+1] -> float64 [n_img, k] normalised histograms
    (`out`: a preallocated contiguous float64 [n_img, k] device tensor, a row block of a larger one
    for instance)."""
    if codes.dtype != torch.int32 or offsets.dtype != torch.int64:
        raise ValueError("codes must be int32 and offsets int64")
    if norm not in _lib.NORM_MODES:
        raise ValueError("norm must be one of %s" % sorted(_lib.NORM_MODES))
    dev = _dev_index(offsets)
    n_img = offsets.shape[0] - 1
    if out is None:
        out = torch.empty((n_img, int(k)), dtype=torch.float64, device=offsets.device)
    elif (out.dtype != torch.float64 or tuple(out.shape) != (n_img, int(k))
          or not out.is_contiguous() or out.device != offsets.device):
        raise ValueError("out must be a contiguous float64 [%d, %d] tensor on %s"
                         % (n_img, int(k), offsets.device))
    rc = _lib.load().ic_hist(_lib.handle(dev), _ptr(codes.contiguous()), _ptr(offsets.contiguous()),
                             n_img, int(k), _lib.NORM_MODES[norm], float(power), _ptr(out),
                             _stream_ptr(dev))
    _lib.check(rc, "ic_hist")
    return out


def spatial_hist(codes, kp_cells, cell_weights, k):
    """codes int32 [n_img, n_kp], kp_cells int32 [n_levels, n_kp], cell_weights float64 [n_cells]
    -> float64 [n_img, n_cells * k] weighted spatial-pyramid counts."""
    if codes.dtype != torch.int32 or kp_cells.dtype != torch.int32 or codes.dim() != 2:
        raise ValueError("codes must be int32 [n_img, n_kp] and kp_cells int32 [n_levels, n_kp]")
    if kp_cells.dim() != 2 or kp_cells.shape[1] != codes.shape[1]:
        raise ValueError("kp_cells must have one column per keypoint")
    if cell_weights.dtype != torch.float64:
        raise ValueError("cell_weights must be float64")
    dev = _dev_index(codes)
    n_img, n_kp = codes.shape
    n_levels = kp_cells.shape[0]
    n_cells = sum(4 ** l for l in range(n_levels))
    if cell_weights.shape[0] != n_cells:
        raise ValueError("need %d cell weights for %d levels" % (n_cells, n_levels))
    out = torch.empty((n_img, n_cells * int(k)), dtype=torch.float64, device=codes.device)
    rc = _lib.load().ic_spatial_hist(_lib.handle(dev), _ptr(codes.contiguous()),
                                     _ptr(kp_cells.contiguous()), n_img, n_kp, n_levels, int(k),
                                     _ptr(cell_weights.contiguous()), _ptr(out), _stream_ptr(dev))
    _lib.check(rc, "ic_spatial_hist")
    return out


def centroid_partials(x, codes, k, out=None):
    """sums float64 [k, d] and counts int64 [k] of the rows of x grouped by code.
    `out` = (sums, counts) preallocated contiguous device tensors (views are fine)."""
    _check_2d_f32(x, "x")
    if codes.dtype != torch.int32 or codes.shape[0] != x.shape[0]:
        raise ValueError("codes must be int32 with one entry per row of x")
    dev = _dev_index(x)
    if out is not None:
        sums, counts = out
        if (sums.dtype != torch.float64 or tuple(sums.shape) != (int(k), x.shape[1])
                or counts.dtype != torch.int64 or tuple(counts.shape) != (int(k),)
                or not sums.is_contiguous() or not counts.is_contiguous()
                or sums.device != x.device or counts.device != x.device):
            raise ValueError("out must be (float64 [k, d], int64 [k]) contiguous on x's device")
    else:
        sums = torch.empty((int(k), x.shape[1]), dtype=torch.float64, device=x.device)
        counts = torch.empty(int(k), dtype=torch.int64, device=x.device)
    rc = _lib.load().ic_centroid_partials(_lib.handle(dev), _ptr(x), _ptr(codes.contiguous()),
                                          x.shape[0], int(k), x.shape[1], _ptr(sums), _ptr(counts),
                                          _stream_ptr(dev))
    _lib.check(rc, "ic_centroid_partials")
    return sums, counts


def centroid_finalize(sums, counts, prev_c):
    """new_c = sums / counts, or prev_c where a cluster is empty; also returns has_members."""
    _check_2d_f32(prev_c, "prev_c")
    if sums.dtype != torch.float64 or counts.dtype != torch.int64:
        raise ValueError("sums must be float64 and counts int64")
    dev = _dev_index(prev_c)
    k, d = prev_c.shape
    new_c = torch.empty_like(prev_c)
    has = torch.empty(k, dtype=torch.uint8, device=prev_c.device)
    rc = _lib.load().ic_centroid_finalize(_lib.handle(dev), _ptr(sums.contiguous()),
                                          _ptr(counts.contiguous()), _ptr(prev_c), k, d,
                                          _ptr(new_c), _ptr(has), _stream_ptr(dev))
    _lib.check(rc, "ic_centroid_finalize")
    return new_c, has


def _metric_number(metric):
    if metric not in _lib.METRICS:
        raise ValueError("metric must be one of %s, got %r" % (sorted(_lib.METRICS), metric))
    return _lib.METRICS[metric]


def pairwise(x, c, metric="euclidean"):
    """float64 [m, n] matrix of `metric` between the rows of x [m, d] and c [n, d] (float32)."""
    _check_2d_f32(x, "x")
    _check_2d_f32(c, "c")
    if x.shape[1] != c.shape[1] or x.shape[1] < 1:
        raise ValueError("dimension mismatch: x has %d columns, c %d" % (x.shape[1], c.shape[1]))
    dev = _dev_index(x)
    out = torch.empty((x.shape[0], c.shape[0]), dtype=torch.float64, device=x.device)
    rc = _lib.load().ic_pairwise(_lib.handle(dev), _ptr(x), x.shape[0], _ptr(c), c.shape[0],
                                 x.shape[1], _metric_number(metric), _ptr(out), _stream_ptr(dev))
    _lib.check(rc, "ic_pairwise")
    return out


def knn_rows(x, c, k, metric="euclidean"):
    """(dist float64 [m, k], idx int32 [m, k]): the k rows of c nearest to every row of x,
    ascending by (distance, index)."""
    _check_pair(x, c, descriptor_search=False)
    dev = _dev_index(x)
    k = int(k)
    dist = torch.empty((x.shape[0], k), dtype=torch.float64, device=x.device)
    idx = torch.empty((x.shape[0], k), dtype=torch.int32, device=x.device)
    rc = _lib.load().ic_knn_rows(_lib.handle(dev), _ptr(x), x.shape[0], _ptr(c), c.shape[0],
                                 x.shape[1], _metric_number(metric), k, _ptr(dist), _ptr(idx),
                                 _stream_ptr(dev))
    _lib.check(rc, "ic_knn_rows")
    return dist, idx


def compact_matches(mask, i1, row_offsets, kp_query=None, kp_rows=None):
    """Per-candidate match lists from the ratio-test mask of one batched search.

    mask uint8 [rows], i1 int32 [rows], row_offsets int64 [n_cand + 1] (device).  Returns
    (out_offsets int64 [n_cand + 1], match_q int32 [rows], match_im int32 [rows], pts): candidate
    c's matches are entries out_offsets[c]:out_offsets[c+1]; pts float64 [rows, 4] holds
    (x_q, y_q, x_im, y_im) per match when both keypoint arrays (float64 [n, 2]) are given."""
    if mask.dtype != torch.uint8 or i1.dtype != torch.int32 or row_offsets.dtype != torch.int64:
        raise ValueError("mask must be uint8, i1 int32 and row_offsets int64")
    rows = mask.shape[0]
    if i1.shape[0] != rows:
        raise ValueError("mask and i1 must have one entry per search row")
    dev = _dev_index(mask)
    n_cand = row_offsets.shape[0] - 1
    out_off = torch.empty(n_cand + 1, dtype=torch.int64, device=mask.device)
    match_q = torch.empty(rows, dtype=torch.int32, device=mask.device)
    match_im = torch.empty(rows, dtype=torch.int32, device=mask.device)
    pts = None
    n_q = 0
    if kp_query is not None or kp_rows is not None:
        if kp_query is None or kp_rows is None:
            raise ValueError("give both kp_query and kp_rows, or neither")
        for t, name in ((kp_query, "kp_query"), (kp_rows, "kp_rows")):
            if t.dtype != torch.float64 or t.dim() != 2 or t.shape[1] != 2 or not t.is_contiguous():
                raise ValueError("%s must be contiguous float64 [n, 2]" % name)
        if kp_rows.shape[0] != rows or kp_query.shape[0] < 1:
            raise ValueError("kp_rows needs one keypoint per search row and kp_query at least one")
        n_q = kp_query.shape[0]
        pts = torch.empty((rows, 4), dtype=torch.float64, device=mask.device)
    rc = _lib.load().ic_compact_matches(
        _lib.handle(dev), _ptr(mask.contiguous()), _ptr(i1.contiguous()),
        _ptr(row_offsets.contiguous()), n_cand, _ptr(kp_query), n_q, _ptr(kp_rows), _ptr(out_off),
        _ptr(match_q), _ptr(match_im), _ptr(pts), _stream_ptr(dev))
    _lib.check(rc, "ic_compact_matches")
    return out_off, match_q, match_im, pts


def ransac_homography(pts, offsets, n_iter, th, samples=None, seed=0, min_inliers=10,
                      accept_mean_err=True, cv_rules=False):
    """Homography RANSAC for every candidate's matched pairs at once.

    cv_rules=False: the reference's own ransac() loop; cv_rules=True: the loop of
    cv2.findHomography(..., cv2.RANSAC, th) (inliers at squared error <= th^2, adaptive iteration
    limit at confidence 0.995 starting from n_iter, refit over the inliers + LM polish).

    pts float64 [total, 4], offsets int64 [n_cand + 1] (device), samples int32
    [n_cand, n_iter, 4] or None.  Returns (H float64 [n_cand, 3, 3], n_inliers int32 [n_cand],
    best_err float64 [n_cand], best_iter int32 [n_cand], inlier_mask uint8 [total])."""
    if pts.dtype != torch.float64 or pts.dim() != 2 or pts.shape[1] != 4 or not pts.is_contiguous():
        raise ValueError("pts must be contiguous float64 [total, 4]")
    if offsets.dtype != torch.int64:
        raise ValueError("offsets must be int64")
    dev = _dev_index(pts)
    n_cand = offsets.shape[0] - 1
    n_iter = int(n_iter)
    if samples is not None:
        if samples.dtype != torch.int32 or tuple(samples.shape) != (n_cand, n_iter, 4):
            raise ValueError("samples must be int32 [n_cand, n_iter, 4]")
        samples = samples.contiguous()
    H = torch.empty((n_cand, 3, 3), dtype=torch.float64, device=pts.device)
    n_inl = torch.empty(n_cand, dtype=torch.int32, device=pts.device)
    err = torch.empty(n_cand, dtype=torch.float64, device=pts.device)
    best_it = torch.empty(n_cand, dtype=torch.int32, device=pts.device)
    mask = torch.zeros(pts.shape[0], dtype=torch.uint8, device=pts.device)
    rc = _lib.load().ic_ransac_homography(
        _lib.handle(dev), _ptr(pts), _ptr(offsets.contiguous()), n_cand, n_iter, _ptr(samples),
        int(seed) & 0xFFFFFFFFFFFFFFFF, float(th), int(min_inliers),
        (1 if accept_mean_err else 0) | (2 if cv_rules else 0),
        _ptr(H), _ptr(n_inl), _ptr(err), _ptr(best_it), _ptr(mask), _stream_ptr(dev))
    _lib.check(rc, "ic_ransac_homography")
    return H, n_inl, err, best_it, mask


def reserve_workspace(device_index, nbytes):
    """Grow the handle's device workspace to at least `nbytes` now, so that later calls of known
    size do not allocate."""
    _lib.check(_lib.load().ic_reserve_workspace(_lib.handle(device_index), int(nbytes)),
               "ic_reserve_workspace")


def workspace_bytes(device_index):
    """Current size of the handle's device workspace."""
    out = ctypes.c_size_t(0)
    _lib.check(_lib.load().ic_workspace_bytes(_lib.handle(device_index), ctypes.byref(out)),
               "ic_workspace_bytes")
    return int(out.value)


def profile_begin(device_index):
    """Start timing the distance kernel of every following assign call on this device."""
    _lib.check(_lib.load().ic_profile_begin(_lib.handle(device_index)), "ic_profile_begin")


def profile_end(device_index):
    """(summed kernel milliseconds, launches) since profile_begin; waits for the kernels."""
    ms = ctypes.c_double(0.0)
    n = ctypes.c_int64(0)
    rc = _lib.load().ic_profile_end(_lib.handle(device_index), ctypes.byref(ms), ctypes.byref(n))
    _lib.check(rc, "ic_profile_end")
    return ms.value, n.value

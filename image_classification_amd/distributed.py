"""Multi-GPU decomposition of the path: one process per GPU, torch.distributed (backend "nccl" is
RCCL on ROCm) for the two steps that really exchange data.

  * assignment / BoW quantisation / video batches: rows are independent, so each rank takes a
    contiguous block of images and there is NO collective (`quantize_sharded`; the optional gather
    of the histograms to one rank is for the caller's convenience only).
  * k-means training: rows sharded, one all-reduce(sum) per Lloyd iteration over a single packed
    float64 buffer holding the centroid sums and the member counts (`kmeans_fit_sharded`).
  * 2-NN with the INDEX side sharded (the logo database of model/robust_matching.py split over
    GPUs): every rank searches the replicated queries in its shard, the 16-byte per-row records
    are all-gathered and merged by (distance, global index) (`knn2_index_sharded`).
  * 2-NN with the SEARCH rows sharded (the reference's own direction, model/robust_matching.py:80:
    the index is fitted on the query image): rows are independent, results concatenate
    (`knn2_rows_sharded`).

The reference has no distributed code; these functions have no counterpart there.  Every function
takes the compute backend explicitly (HipBackend by default), so the collectives and the sharding
arithmetic are testable with gloo on CPU-only machines.
"""
import numpy as np
import torch
import torch.distributed as dist

from .backend import default_backend
from .clustering import lloyd


def world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_bounds(n_items, n_shards):
    """Contiguous blocks, sizes differing by at most one: int64 [n_shards + 1]."""
    base, rem = divmod(int(n_items), int(n_shards))
    sizes = [base + (1 if r < rem else 0) for r in range(n_shards)]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def shard_images(offsets, n_shards):
    """Split images (not rows) into contiguous blocks of nearly equal row counts.

    offsets: int64 [n_img + 1].  Returns image bounds int64 [n_shards + 1]; shard r owns images
    [b[r], b[r+1]) and therefore rows [offsets[b[r]], offsets[b[r+1]]).
    """
    offsets = np.asarray(offsets, dtype=np.int64)
    total = offsets[-1]
    targets = [total * r // n_shards for r in range(1, n_shards)]
    cuts = [int(np.searchsorted(offsets, t, side="left")) for t in targets]
    b = np.asarray([0] + cuts + [len(offsets) - 1], dtype=np.int64)
    return np.maximum.accumulate(b)


def quantize_sharded(x_local, offsets_local, codebook, norm="l1", power=0.5, backend=None,
                     gather_to=None, group=None):
    """BoW quantisation of this rank's images.  Returns (codes, hists) for the local images; with
    `gather_to=r` rank r also receives the list of every rank's histograms (others get None)."""
    b = backend or default_backend()
    cb = b.codebook(codebook)
    xd = b.f32(x_local)
    off = b.i64(offsets_local)
    if xd.shape[0]:
        codes, _ = b.assign_top1(xd, cb, want_dist=False)
    else:
        codes = b.i32(np.empty(0, dtype=np.int32))
    hists = b.hist(codes, off, cb.shape[0], norm, power)
    gathered = None
    rank, n = world(group)
    if gather_to is not None and n > 1:
        gathered = gather_rows(hists, group=group)
        if rank != gather_to:
            gathered = None
    elif gather_to is not None:
        gathered = [hists]
    return codes, hists, gathered


def _staged(t, group):
    """Over gloo (several ranks rehearsing the multi-GPU path on one device, or CPU tests) device
    tensors travel through host copies; RCCL works on them in place."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_reduce(t, group, op=None):
    op = op if op is not None else dist.ReduceOp.SUM
    if _staged(t, group):
        host = t.cpu()
        dist.all_reduce(host, op=op, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=op, group=group)


def _broadcast(t, src, group):
    if _staged(t, group):
        host = t.cpu()
        dist.broadcast(host, src=src, group=group)
        t.copy_(host)
    else:
        dist.broadcast(t, src=src, group=group)


def _all_gather_into(flat, t, group):
    if _staged(t, group):
        host = torch.empty(flat.shape, dtype=flat.dtype)
        dist.all_gather_into_tensor(host, t.cpu().contiguous(), group=group)
        flat.copy_(host)
    else:
        dist.all_gather_into_tensor(flat, t.contiguous(), group=group)


def gather_rows(t, group=None):
    """All-gather tensors whose first dimension differs per rank; returns the list of them."""
    rank, n = world(group)
    if n == 1:
        return [t]
    sizes = torch.zeros(n, dtype=torch.int64, device=t.device)
    sizes[rank] = t.shape[0]
    _all_reduce(sizes, group)
    sizes = [int(s) for s in sizes.cpu()]
    m = max(sizes)
    padded = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    padded[:t.shape[0]] = t
    flat = torch.empty((n * m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    _all_gather_into(flat, padded, group)
    return [flat[r * m:r * m + s] for r, s in enumerate(sizes)]


class PackedAllReduce(object):
    """In-place sum over ranks of (sums float64 [k,d], counts int64 [k]) as ONE float64 buffer.

    The buffer lives as long as this object: `outputs(k, d, device)` hands out the views the
    centroid-sum kernel writes into, so an iteration moves no copy of the sums (67 MB at
    k = 65536) -- only the k counts are converted (exact below 2^53 members) into the buffer's
    tail before the collective and back after it."""

    def __init__(self, group=None):
        self.group = group
        self._buf = None
        self._key = None

    def outputs(self, k, d, device):
        key = (int(k), int(d), str(device))
        if self._key != key:
            self._buf = torch.empty(k * d + k, dtype=torch.float64, device=device)
            self._counts = torch.empty(k, dtype=torch.int64, device=device)
            self._key = key
        return self._buf[:k * d].view(k, d), self._counts

    def __call__(self, sums, counts):
        k, d = sums.shape
        mine = self._buf is not None and self._key[:2] == (k, d) \
            and sums.data_ptr() == self._buf.data_ptr()
        if not mine:                                   # sums produced elsewhere: copy in and out
            own_sums, _ = self.outputs(k, d, sums.device)
            own_sums.copy_(sums)
        tail = self._buf[k * d:]
        tail.copy_(counts)                             # int64 -> float64
        _all_reduce(self._buf, self.group)
        counts.copy_(tail.round().to(torch.int64))
        if not mine:
            sums.copy_(self._buf[:k * d].view(k, d))


def make_allreduce(group=None):
    return PackedAllReduce(group)


def kmeans_fit_sharded(x_local, init, iters, backend=None, group=None, on_empty=None):
    """Lloyd iterations with the rows sharded over ranks.

    x_local: this rank's rows; init: [k, d] initial centroids, identical on every rank (broadcast
    from rank 0 here to make sure).  Returns (codebook device tensor, identical on all ranks;
    labels of the local rows; empty-cluster count per iteration).
    """
    b = backend or default_backend()
    rank, n = world(group)
    xd = b.f32(x_local)
    c0 = b.f32(init)
    if n > 1:
        _broadcast(c0, dist.get_global_rank(group, 0) if group is not None else 0, group)
    return lloyd(xd, c0, iters, b, allreduce=make_allreduce(group) if n > 1 else None,
                 on_empty=on_empty)


_trailers = {}


def pack_records(rec, index_base):
    """[n, 4] records -> [n + 1, 4] whose trailing record carries the shard's base (an int64 in
    its first two words): the block a rank contributes to the all-gather, so the bases need no
    collective of their own and never come back to the host."""
    key = (str(rec.device), int(index_base))
    trailer = _trailers.get(key)
    if trailer is None:
        words = np.zeros(4, dtype=np.int32)
        words[:2] = np.array([int(index_base)], dtype=np.int64).view(np.int32)
        if len(_trailers) > 64:
            _trailers.clear()
        trailer = _trailers[key] = torch.from_numpy(words).to(rec.device)
    out = torch.empty((rec.shape[0] + 1, 4), dtype=torch.int32, device=rec.device)
    out[:-1] = rec
    out[-1] = trailer
    return out


def gather_records(rec, index_base, group=None):
    """All-gather every rank's packed records: int32 [n_ranks, n + 1, 4] (see pack_records)."""
    rank, n = world(group)
    send = pack_records(rec, index_base)
    if n == 1:
        return send.unsqueeze(0)
    flat = torch.empty((n * send.shape[0], 4), dtype=send.dtype, device=send.device)
    _all_gather_into(flat, send, group)               # rank r -> rows [r*(n+1), (r+1)*(n+1))
    return flat.view(n, send.shape[0], 4)


def knn2_index_sharded(queries, index_local, index_base, ratio=0.8, backend=None, group=None):
    """Two nearest index rows of every query when the index is split over ranks.

    queries: replicated [n, d]; index_local: this rank's index rows; index_base: global row number
    of its first row.  Returns (i1 int64, d1, i2 int64, d2, mask) identical on every rank.  One
    collective (the all-gather of 16 bytes per query and rank), no host round trip.
    """
    b = backend or default_backend()
    rec = b.assign_top2_rec(b.f32(queries), b.codebook(index_local))
    return b.top2_merge(gather_records(rec, index_base, group), None, ratio)


def knn2_rows_sharded(index, rows_local, ratio=0.8, backend=None, gather=False, group=None):
    """The reference's direction: the (small) index is replicated, the search rows are sharded.
    Rows are independent; `gather=True` concatenates every rank's results on every rank."""
    b = backend or default_backend()
    res = b.assign_top2(b.f32(rows_local), b.codebook(index), ratio)
    if not gather:
        return res
    return tuple(torch.cat(gather_rows(t, group=group)) for t in res)

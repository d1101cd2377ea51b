"""The compute backend the host-side mirrors call.

`HipBackend` is the only backend this package ships: every method is one C-ABI call into
libic_hip.so (see ops.py).  The host logic (batching, sharding, the Lloyd loop, the collectives)
takes the backend as an argument so that the multi-process tests can drive that logic on CPU-only
machines with a checker backend of their own; nothing in the package ever substitutes one.
"""
import numpy as np
import torch

from . import device as _device
from . import ops


def _as_f32_rows(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class HipBackend(object):
    name = "hip-gfx950"

    def __init__(self, device=None):
        self.device = device or _device.default_device()

    # -- data movement -------------------------------------------------------------------------
    def f32(self, a):
        return ops.as_device_f32(a, self.device)

    def i32(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=torch.int32).contiguous()
        return torch.as_tensor(a, dtype=torch.int32).to(self.device)

    def i64(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=torch.int64).contiguous()
        return torch.as_tensor(a, dtype=torch.int64).to(self.device)

    def codebook(self, c):
        return _device.resident(c, self.device)

    def to_host(self, t):
        return ops.to_host(t)

    # -- searches whose rows start in host memory (numpy): streamed upload -----------------------
    def assign_top1_np(self, x_np, cb, want_dist=True):
        x_np = _as_f32_rows(x_np)
        if x_np.shape[0] <= ops.HOST_CHUNK_ROWS:
            return self.assign_top1(self.f32(x_np), cb, want_dist)
        return ops.assign_from_host(x_np, cb.c, cb.aux, top2=False, want_dist=want_dist)

    def assign_top1_host(self, x_np, cb):
        """(idx int32, dist float32) as numpy arrays.  Small inputs come back in ONE download:
        indices and distances are written into the two halves of one device buffer (a per-image
        call is dominated by such fixed costs, not by the search)."""
        x_np = _as_f32_rows(x_np)
        n = x_np.shape[0]
        if n > ops.HOST_CHUNK_ROWS:
            idx, dist = ops.assign_from_host(x_np, cb.c, cb.aux, top2=False, want_dist=True)
            return idx.cpu().numpy(), dist.cpu().numpy()
        both = torch.empty(2 * n, dtype=torch.int32, device=self.device)
        ops.assign_top1(self.f32(x_np), cb.c, cb.aux, True,
                        out=(both[:n], both[n:].view(torch.float32)))
        host = both.cpu().numpy()
        return host[:n], host[n:].view(np.float32)

    def assign_top2_np(self, x_np, cb, ratio=0.8):
        x_np = _as_f32_rows(x_np)
        if x_np.shape[0] <= ops.HOST_CHUNK_ROWS:
            return self.assign_top2(self.f32(x_np), cb, ratio)
        return ops.assign_from_host(x_np, cb.c, cb.aux, top2=True, ratio=ratio)

    # -- searches whose rows are a list of host arrays (one per image): threaded gather ---------
    def assign_top1_parts(self, parts, cb, want_dist=True):
        total = sum(p.shape[0] for p in parts)
        if total <= ops.HOST_CHUNK_ROWS // 4:
            return self.assign_top1(ops.upload_parts(parts, self.device), cb, want_dist)
        return ops.assign_from_host_parts(parts, cb.c, cb.aux, top2=False, want_dist=want_dist)

    def quantize_parts(self, parts, offsets, cb, k, norm="l1", power=0.5):
        """(codes device tensor, histograms numpy float64 [n_img, k]) for descriptors held as a
        list of host arrays: small inputs in three plain steps, large ones through the pipeline
        that takes histogram blocks down while later rows are still being searched."""
        total = sum(p.shape[0] for p in parts)
        if 0 < total <= ops.HOST_CHUNK_ROWS // 4:
            # the images' offsets ride behind the rows in one upload
            x, off = ops.upload_parts(parts, self.device, extra_i64=offsets)
            codes, _ = self.assign_top1(x, cb, want_dist=False)
            return codes, self.to_host(self.hist(codes, off, k, norm, power))
        if total <= ops.HOST_CHUNK_ROWS * 2:
            # (the offsets go up first: a copy from pageable memory blocks the caller until
            # everything enqueued before it has run, which would otherwise include the search)
            off = self.i64(offsets)
            if parts:
                codes, _ = self.assign_top1_parts(parts, cb, want_dist=False)
            else:
                codes = self.i32(np.empty(0, dtype=np.int32))
            hists = self.hist(codes, off, k, norm, power)
            return codes, self.to_host(hists)
        return ops.quantize_parts_to_host(parts, offsets, cb.c, cb.aux, k, norm, power)

    def assign_top2_parts(self, parts, cb, ratio=0.8):
        total = sum(p.shape[0] for p in parts)
        if total <= ops.HOST_CHUNK_ROWS // 4:
            return self.assign_top2(ops.upload_parts(parts, self.device), cb, ratio)
        return ops.assign_from_host_parts(parts, cb.c, cb.aux, top2=True, ratio=ratio)

    # -- operators -----------------------------------------------------------------------------
    def assign_top1(self, x, cb, want_dist=True, hint=None):
        return ops.assign_top1(x, cb.c, cb.aux, want_dist, hint=hint)

    def assign_top2(self, x, cb, ratio=0.8):
        return ops.assign_top2(x, cb.c, ratio, cb.aux)

    def assign_top2_rec(self, x, cb):
        return ops.assign_top2_rec(x, cb.c, cb.aux)

    def top2_merge(self, gathered, shard_base, ratio=0.8):
        return ops.top2_merge(gathered, shard_base, ratio)

    def hist(self, codes, offsets, k, norm="l1", power=0.5):
        return ops.hist(codes, offsets, k, norm, power)

    def f64(self, a):
        return torch.as_tensor(a, dtype=torch.float64).to(self.device)

    def spatial_hist(self, codes, kp_cells, cell_weights, k):
        return ops.spatial_hist(codes, kp_cells, cell_weights, k)

    def centroid_partials(self, x, codes, k, out=None):
        return ops.centroid_partials(x, codes, k, out=out)

    def centroid_finalize(self, sums, counts, prev_c):
        return ops.centroid_finalize(sums, counts, prev_c)

    def pairwise(self, x, c, metric="euclidean"):
        return ops.pairwise(x, c, metric)

    def knn_rows(self, x, c, k, metric="euclidean"):
        return ops.knn_rows(x, c, k, metric)

    def u8(self, a):
        return torch.as_tensor(a, dtype=torch.uint8).to(self.device)

    def compact_matches(self, mask, i1, row_offsets, kp_query=None, kp_rows=None):
        return ops.compact_matches(mask, i1, row_offsets, kp_query, kp_rows)

    def ransac_homography(self, pts, offsets, n_iter, th, samples=None, seed=0, min_inliers=10,
                          accept_mean_err=True, cv_rules=False):
        return ops.ransac_homography(pts, offsets, n_iter, th, samples, seed, min_inliers,
                                     accept_mean_err, cv_rules)

    def make_codebook(self, c_dev):
        """Wrap a device tensor produced by centroid_finalize as a codebook."""
        return _device.ResidentCodebook(c_dev, self.device)


_default = {}


def default_backend():
    dev = _device.default_device()
    key = str(dev)
    b = _default.get(key)
    if b is None:
        b = _default[key] = HipBackend(dev)
    return b

"""Comparison rules for device results against the float64 oracle (test infrastructure).

Indices must be equal wherever the oracle's gap between the neighbour in question and the next
one exceeds 1e-5 relative; distances, centroids and normalised histograms within 1e-4 relative.
"""
import numpy as np

__all__ = ["check_indices", "check_close", "agreement", "mismatches_outside_f32_band"]

GAP_REL = 1e-5
VAL_REL = 1e-4


def _gap_ok(d_this, d_next):
    """True where the neighbour is separated from the next candidate by more than GAP_REL."""
    with np.errstate(invalid="ignore", divide="ignore"):
        rel = (d_next - d_this) / np.where(d_this > 0, d_this, 1.0)
    rel = np.where(np.isfinite(d_next), rel, np.inf)
    return rel > GAP_REL


def check_indices(got, ref_ind, ref_dist, rank=0, got_dist=None):
    """`got` [n] vs column `rank` of the oracle's top-k (ref_ind/ref_dist [n, >= rank+2]).

    A row must match when the oracle's neighbour `rank` is separated by more than GAP_REL both
    from the one after it and (for rank > 0) from the one before it.  Rows inside the gap may
    pick either neighbour.  Returns (n_must_match, n_mismatch_where_required, overall agreement).
    """
    got = np.asarray(got).astype(np.int64)
    must = _gap_ok(ref_dist[:, rank], ref_dist[:, rank + 1])
    if rank > 0:
        must &= _gap_ok(ref_dist[:, rank - 1], ref_dist[:, rank])
    eq = got == ref_ind[:, rank]
    bad = int(np.count_nonzero(must & ~eq))
    return int(np.count_nonzero(must)), bad, float(np.mean(eq)) if len(eq) else 1.0


def mismatches_outside_f32_band(got, x, c, ref_ind, ref_dist, rank=0, ulps=32.0):
    """Rows that check_indices would count as mismatches AND whose returned neighbour is not
    explained by float32 resolution.

    The device ranks candidates by |c|^2 - 2 x.c in float32; that quantity carries an absolute
    rounding error of a few ulps of |x|^2 + |c|^2, whatever the distance itself is.  When rows lie
    much closer to their neighbours than their own norm (one-dimensional data, near duplicates),
    two candidates whose squared distances differ by less than that error cannot be told apart
    in float32 even though their relative gap exceeds GAP_REL.  A returned index is accepted when
    its TRUE squared distance (float64) is within `ulps` float32 ulps of |x|^2 + max|c|^2 of the
    oracle's squared distance at that rank."""
    got = np.asarray(got).astype(np.int64)
    x64 = np.asarray(x, dtype=np.float64)
    c64 = np.asarray(c, dtype=np.float64)
    must = _gap_ok(ref_dist[:, rank], ref_dist[:, rank + 1])
    if rank > 0:
        must &= _gap_ok(ref_dist[:, rank - 1], ref_dist[:, rank])
    rows = np.nonzero(must & (got != ref_ind[:, rank]))[0]
    if rows.size == 0:
        return 0
    cmax = float((c64 ** 2).sum(1).max())
    bad = 0
    for r in rows:
        if got[r] < 0:
            bad += 1
            continue
        true_sq = float(((x64[r] - c64[got[r]]) ** 2).sum())
        band = ulps * np.finfo(np.float32).eps * (float((x64[r] ** 2).sum()) + cmax)
        if abs(true_sq - float(ref_dist[r, rank]) ** 2) > band:
            bad += 1
    return bad


def agreement(got, ref):
    got = np.asarray(got).astype(np.int64)
    ref = np.asarray(ref).astype(np.int64)
    return float(np.mean(got == ref)) if len(ref) else 1.0


def check_close(got, ref, rel=VAL_REL, abs_floor=1e-6):
    """max over elements of |got - ref| / max(|ref|, abs_floor) <= rel; returns that maximum."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    both_inf = np.isinf(got) & np.isinf(ref) & (np.sign(got) == np.sign(ref))
    err = np.abs(got - ref) / np.maximum(np.abs(ref), abs_floor)
    err = np.where(both_inf, 0.0, err)
    return float(err.max()) if err.size else 0.0

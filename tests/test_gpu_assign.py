"""GPU parity of the distance / top-2 kernels against the float64 oracle, through the C ABI."""
import numpy as np
import pytest
import torch

import oracle
from image_classification_amd import ops
from image_classification_amd.synthetic import sift_like, surf_like

pytestmark = pytest.mark.gpu

# Tolerances (BASELINE north_star): indices equal wherever the oracle's gap to the next candidate
# exceeds 1e-5 relative; distances within 1e-4 relative.
DIST_REL = 1e-4


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _run_top2(x, c, ratio=0.8):
    i1, d1, i2, d2, mask = ops.assign_top2(_dev(x), _dev(c), ratio)
    torch.cuda.synchronize()
    return (i1.cpu().numpy(), d1.cpu().numpy(), i2.cpu().numpy(), d2.cpu().numpy(),
            mask.cpu().numpy().astype(bool))


def _check_top2(x, c, ratio=0.8, min_agree=0.9999):
    i1, d1, i2, d2, mask = _run_top2(x, c, ratio)
    rd, ri = oracle.topk(x.astype(np.float32), c.astype(np.float32), 3)
    must1, bad1, agree1 = oracle.check_indices(i1, ri, rd, 0)
    assert bad1 == 0, "%d of %d well-separated best neighbours differ" % (bad1, must1)
    assert agree1 >= min_agree
    if c.shape[0] >= 2:
        must2, bad2, agree2 = oracle.check_indices(i2, ri, rd, 1)
        assert bad2 == 0, "%d of %d well-separated second neighbours differ" % (bad2, must2)
        assert agree2 >= min_agree - 1e-3
        ok1 = i1 == ri[:, 0]
        ok2 = i2 == ri[:, 1]
        assert oracle.check_close(d1[ok1], rd[ok1, 0]) <= DIST_REL
        assert oracle.check_close(d2[ok2], rd[ok2, 1]) <= DIST_REL
        # ratio mask: must agree except within 1e-5 relative of the threshold
        ref_mask = rd[:, 0] < rd[:, 1] * ratio
        margin = np.abs(rd[:, 0] - rd[:, 1] * ratio) / np.maximum(rd[:, 1], 1e-30)
        clear = (margin > 1e-5) & ok1 & ok2
        assert np.array_equal(mask[clear], ref_mask[clear])
    return agree1


@pytest.mark.parametrize("n,k", [(2048, 64), (1000, 1000), (513, 65), (31, 2), (7, 1), (4096, 130)])
def test_top2_vs_oracle_sift(n, k):
    x = sift_like(n, seed=11)
    c = sift_like(k, seed=12)
    _check_top2(x, c)


def test_top2_surf_like_signed_unit_norm():
    x = surf_like(3000, seed=3)
    c = surf_like(500, seed=4)
    _check_top2(x, c)


def test_config1_100k_by_1k():
    # BASELINE config 1 at full size: 100k descriptors vs a 1k-word codebook
    x = sift_like(100000, seed=1)
    c = sift_like(1000, seed=2)
    agree = _check_top2(x, c)
    assert agree >= 0.9999


def test_rows_beyond_whole_machine_rounds_are_k_split():
    # 256 row tiles of 512 rows fill one round of the machine and are searched unsplit; the 700
    # rows after them form a partly filled round whose codebook scan is split over the CUs
    x = sift_like(256 * 512 + 700, seed=13)
    c = sift_like(700, seed=14)
    _check_top2(x, c)


@pytest.mark.parametrize("n,k", [(70_000, 1000), (90_000, 500), (140_000, 300), (3000, 5000),
                                 (100, 5000), (131_072 + 40_000, 2000)])
def test_runs_that_cross_row_tiles_equal_the_direct_search(n, k):
    # the row tiles of the last, partly filled round are walked in runs of equal length that end
    # inside one row tile and go on in the next (70 000 x 1000: 274 row tiles x 16 stages in runs
    # of 9; 140 000 x 300: 35 tail row tiles x 5 stages in runs of 2), or cut one row tile into
    # many pieces (100 x 5000): every row must still get the direct search's answer
    _same_as_direct(sift_like(n, n_modes=512, seed=n % 97), sift_like(k, n_modes=512, seed=k % 89))


def test_ties_lowest_index_wins():
    # duplicated codewords and rounded (integer) data: exact ties; scipy/numpy keep the lowest index
    x = sift_like(1500, seed=5, rounded=True)
    c = sift_like(96, seed=6, rounded=True)
    c2 = np.concatenate([c, c[:40], c[10:20]])  # 40 + 10 duplicated codewords
    idx, dist = ops.assign_top1(_dev(x), _dev(c2))
    ref_code, ref_dist = oracle.vq(x, c2)
    got = idx.cpu().numpy()
    assert np.array_equal(got, ref_code)
    assert got.max() < 96  # a duplicate (higher index) never wins
    assert oracle.check_close(dist.cpu().numpy(), ref_dist) <= DIST_REL
    # descriptors equal to codewords: distance exactly 0
    idx2, dist2 = ops.assign_top1(_dev(c[:50]), _dev(c2))
    assert np.array_equal(idx2.cpu().numpy(), np.arange(50))
    assert np.all(dist2.cpu().numpy() == 0.0)
    # top-2 of an exact duplicate pair: (j, j + 96), both at distance 0
    i1, d1, i2, d2, _ = _run_top2(c[:40], c2)
    assert np.array_equal(i1, np.arange(40)) and np.array_equal(i2, np.arange(40) + 96)
    assert np.all(d1 == 0) and np.all(d2 == 0)


@pytest.mark.parametrize("copies,stride", [(4, 1024), (3, 700), (2, 2052)])
def test_exact_ties_across_many_stages(copies, stride):
    # small-integer data (every product and sum exact in float32) and a codebook repeated several
    # times: every candidate ties exactly with copies of itself that sit many 128-codeword stages
    # later and, for strides that are not multiples of 8, in the other half of a tile's rows.
    # The two best must be the two lowest-numbered copies of the nearest codeword.
    rng = np.random.default_rng(copies * 10 + stride)
    base = rng.integers(0, 4, size=(stride, 128)).astype(np.float32)
    c = np.concatenate([base] * copies)
    x = rng.integers(0, 4, size=(3000, 128)).astype(np.float32)
    x[:stride][:500] = base[:500]   # exact hits too
    i1, d1, i2, d2, _ = _run_top2(x, c)
    # exact integer reference; a stable sort keeps the lowest index among equal distances
    xi, ci = x.astype(np.int64), c.astype(np.int64)
    sq = (xi * xi).sum(1)[:, None] - 2 * (xi @ ci.T) + (ci * ci).sum(1)[None, :]
    ri = np.argsort(sq, axis=1, kind="stable")[:, :2]
    rd = np.sqrt(np.take_along_axis(sq, ri, 1).astype(np.float32))
    assert np.array_equal(i1, ri[:, 0])
    assert np.array_equal(i2, ri[:, 1])
    assert np.array_equal(d1, rd[:, 0])
    assert np.array_equal(d2, rd[:, 1])
    idx, _ = ops.assign_top1(_dev(x), _dev(c))
    assert np.array_equal(idx.cpu().numpy(), ri[:, 0])


@pytest.mark.parametrize("d", [64, 36, 100, 1])
def test_other_dimensions_are_zero_padded(d):
    rng = np.random.default_rng(d)
    x = rng.standard_normal((700, d)).astype(np.float32) * 10
    c = rng.standard_normal((90, d)).astype(np.float32) * 10
    _check_top2(x, c, min_agree=0.999)


def test_k_split_large_index_small_query():
    # few search rows, large index: the grid splits the index and merges the partial top-2
    x = sift_like(200, seed=21)
    c = sift_like(60000, n_modes=4096, seed=22)
    _check_top2(x, c, min_agree=0.9999)


def test_planted_matches_ratio_mask():
    rng = np.random.default_rng(7)
    index = sift_like(800, seed=31)
    x = sift_like(5000, seed=32)
    planted = rng.choice(5000, size=250, replace=False)
    src = rng.integers(0, 800, size=250)
    x[planted] = index[src] + rng.normal(0, 2.0, size=(250, 128)).astype(np.float32)
    i1, d1, i2, d2, mask = _run_top2(x, index, 0.8)
    assert np.array_equal(i1[planted], src)
    assert mask[planted].all()
    _check_top2(x, index)


def test_top1_matches_top2_and_codebook_aux():
    x = sift_like(3000, seed=41)
    c = sift_like(300, seed=42)
    xd, cd = _dev(x), _dev(c)
    idx, dist = ops.assign_top1(xd, cd)
    i1, d1, _, _, _ = ops.assign_top2(xd, cd)
    assert torch.equal(idx, i1) and torch.equal(dist, d1)
    # a caller-provided auxiliary block gives the same result
    aux = ops.codebook_aux(cd)
    idx2, dist2 = ops.assign_top1(xd, cd, aux)
    assert torch.equal(idx, idx2) and torch.equal(dist, dist2)
    # its contents: mean vector, the two maxima, the mean's norm (rounded up), centred squared
    # norms, the padding to 128 (a huge finite norm: padding must never be ranked), and behind
    # them the doubled mean the search kernel centres with.  A codebook of this size is scanned
    # in ascending order of the norms: they are stored in that order, followed by the
    # permutation, its inverse and the codewords in scan order.
    a = aux.cpu().numpy()
    c64 = c.astype(np.float64)
    mean = c64.mean(0)
    assert a.shape[0] == 256 + 384 + 128 + 2 * 384 + 300 * 128
    assert oracle.check_close(a[:128], mean, rel=1e-6) <= 1e-6
    cn = ((c64 - a[:128].astype(np.float64)) ** 2).sum(1)
    perm = a[768:768 + 384].view(np.int32)
    rank = a[768 + 384:768 + 768].view(np.int32)
    assert sorted(perm[:300]) == list(range(300)) and np.array_equal(rank[perm[:300]], np.arange(300))
    norms = a[256:256 + 300]
    assert (np.diff(norms) >= 0).all()
    ties = np.diff(norms) == 0
    assert (np.diff(perm[:300])[ties] > 0).all()       # equal norms: by codeword index
    assert oracle.check_close(norms, cn[perm[:300]], rel=1e-5) <= 1e-5
    assert np.array_equal(a[768 + 768:].reshape(300, 128), c[perm[:300]])
    assert (a[256 + 300:256 + 384] > 3.1e38).all() and np.isfinite(a[256 + 300:256 + 384]).all()
    # a codebook too small to be worth ordering keeps the plain layout
    small = ops.codebook_aux(_dev(c[:200])).cpu().numpy()
    assert small.shape[0] == 256 + 256 + 128
    assert oracle.check_close(small[256:256 + 200], ((c64[:200] - small[:128].astype(np.float64)) ** 2).sum(1),
                              rel=1e-5) <= 1e-5
    assert abs(a[128] - cn.max()) <= 1e-5 * cn.max()
    assert abs(a[129] - (c64 ** 2).sum(1).max()) <= 1e-5 * (c64 ** 2).sum(1).max()
    m_norm = np.sqrt((a[:128].astype(np.float64) ** 2).sum())
    assert m_norm <= a[130] <= m_norm * (1 + 1e-6)
    assert np.array_equal(a[256 + 384:768], 2.0 * a[:128])
    # a narrow codebook is padded before its block is taken
    c36 = _dev(c[:, :36])
    i36, d36 = ops.assign_top1(_dev(x[:, :36]), c36, ops.codebook_aux(c36))
    j36, e36 = ops.assign_top1(_dev(x[:, :36]), c36)
    assert torch.equal(i36, j36) and torch.equal(d36, e36)


def _same_as_direct(x, c, ratio=0.8):
    """The fast path must return what the all-direct search returns, bit for bit, on every row."""
    xd, cd = _dev(x), _dev(c)
    counter = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.debug_flagged(0, counter)
    try:
        fast = ops.assign_top2(xd, cd, ratio)
        torch.cuda.synchronize()
        flagged2 = int(counter.item())
        idx, dist = ops.assign_top1(xd, cd)
        torch.cuda.synchronize()
        flagged1 = int(counter.item())
        # index only: most rows are settled from the search records alone
        idx_only, none = ops.assign_top1(xd, cd, want_dist=False)
        torch.cuda.synchronize()
        flagged0 = int(counter.item())
    finally:
        ops.debug_flagged(0, None)
    direct = ops.assign_top2_exact(xd, cd, ratio)
    torch.cuda.synchronize()
    for name, a, b in zip(("i1", "d1", "i2", "d2", "mask"), fast, direct):
        bad = int((a != b).sum()) if a.dtype != torch.float32 else int((a.view(torch.int32) != b.view(torch.int32)).sum())
        assert bad == 0, "%s differs from the direct search on %d of %d rows" % (name, bad, len(a))
    assert torch.equal(idx, direct[0])
    assert torch.equal(dist.view(torch.int32), direct[1].view(torch.int32))
    assert none is None and torch.equal(idx_only, direct[0])
    assert flagged0 <= flagged1      # the shortcut sends no additional row to the second look
    return flagged1, flagged2


def test_fast_path_equals_direct_search_on_descriptor_like_data():
    f1, f2 = _same_as_direct(sift_like(60_000, n_modes=512, seed=81), sift_like(3000, n_modes=512, seed=82))
    # only a small share of the rows needs the second look
    assert f1 < 600 and f2 < 1200, (f1, f2)
    _same_as_direct(surf_like(20_000, seed=83), surf_like(700, seed=84))


def test_fast_path_equals_direct_search_on_offset_data():
    # the same descriptors far from the origin: norms dwarf distances, which is where a search by
    # |c|^2 - 2 x.c alone goes wrong
    x = sift_like(20_000, seed=85) + np.float32(1000.0)
    c = sift_like(900, seed=86) + np.float32(1000.0)
    f1, f2 = _same_as_direct(x, c)
    assert f1 < 2000, f1            # centring keeps the second look rare here too
    rd, ri = oracle.topk(x, c, 3)
    idx, _ = ops.assign_top1(_dev(x), _dev(c))
    must, bad, agree = oracle.check_indices(idx.cpu().numpy(), ri, rd, 0)
    assert bad == 0 and agree > 0.9999


def test_fast_path_equals_direct_search_on_near_ties_and_duplicates():
    rng = np.random.default_rng(9)
    # one-dimensional ramps
    _same_as_direct((rng.standard_normal((5000, 1)) * 300).astype(np.float32),
                    (rng.standard_normal((430, 1)) * 300).astype(np.float32))
    # triples of near-duplicate codewords, rows next to them
    base = sift_like(300, seed=87)
    c = np.concatenate([base, base + rng.normal(0, 1e-3, base.shape).astype(np.float32),
                        base + rng.normal(0, 1e-3, base.shape).astype(np.float32)])
    x = np.concatenate([base[:200] + rng.normal(0, 1e-3, (200, 128)).astype(np.float32),
                        sift_like(3000, seed=88)])
    _same_as_direct(x, c)
    # exact duplicates and integer data (exact ties everywhere): everything is flagged
    xi = rng.integers(0, 3, size=(4000, 128)).astype(np.float32)
    ci = rng.integers(0, 3, size=(500, 128)).astype(np.float32)
    _same_as_direct(xi, np.concatenate([ci, ci, ci[:100]]))
    # so many flagged rows that the second look does not cut the codebook into pieces (more
    # 16-row groups than its grid has workgroups): every best is a seven-way tie (equal norms
    # put the copies next to each other in the scan order, so at least four of them meet in one
    # lane's list, which names three), and the first 131 072 rows are searched unsplit, so
    # k_rescore cannot settle them from other splits' records
    xb = rng.integers(0, 2, size=(140_000, 128)).astype(np.float32)
    cb = rng.integers(0, 2, size=(1200, 128)).astype(np.float32)
    f1, _ = _same_as_direct(xb, np.concatenate([cb] * 7))
    assert f1 > 65536
    # single codeword, two codewords
    _same_as_direct(sift_like(100, seed=89), sift_like(1, seed=90))
    _same_as_direct(sift_like(100, seed=89), sift_like(2, seed=90))


def test_merge_of_index_shards_equals_unsharded():
    x = sift_like(1234, seed=51)
    c = sift_like(5000, n_modes=512, seed=52)
    xd = _dev(x)
    bounds = [0, 1300, 1301, 3333, 5000]  # ragged shards, one of a single row
    recs = [ops.assign_top2_rec(xd, _dev(c[lo:hi])) for lo, hi in zip(bounds[:-1], bounds[1:])]
    gathered = torch.stack(recs).contiguous()
    mi1, md1, mi2, md2, mmask = ops.top2_merge(gathered, bounds[:-1], 0.8)
    # the same with every shard's base packed behind its records (what the ranks all-gather)
    from image_classification_amd.distributed import pack_records
    packed = torch.stack([pack_records(r, b) for r, b in zip(recs, bounds[:-1])]).contiguous()
    for a, b in zip(ops.top2_merge(packed, None, 0.8), (mi1, md1, mi2, md2, mmask)):
        assert torch.equal(a, b)
    big = pack_records(recs[0][:3], (1 << 40) + 5)           # bases beyond 32 bits survive
    assert int(ops.top2_merge(big.unsqueeze(0).contiguous(), None, 0.8)[0][0]) >= (1 << 40)
    i1, d1, i2, d2, mask = ops.assign_top2(xd, _dev(c), 0.8)
    rd, ri = oracle.topk(x, c, 3)
    for got, rank in ((mi1, 0), (mi2, 1)):
        must, bad, _ = oracle.check_indices(got.cpu().numpy(), ri, rd, rank)
        assert bad == 0
    same = (mi1.cpu() == i1.cpu().long()) & (mi2.cpu() == i2.cpu().long())
    assert same.float().mean() > 0.999
    assert torch.equal(md1.cpu()[same], d1.cpu()[same])


def test_streamed_host_path_equals_device_path():
    # numpy rows are uploaded chunk by chunk on a side stream while the previous chunk is searched
    x = sift_like(300_001, seed=61)
    c = sift_like(700, seed=62)
    cd = _dev(c)
    cn = ops.codebook_aux(cd)
    idx_d, dist_d = ops.assign_top1(_dev(x), cd, cn)
    idx_h, dist_h = ops.assign_from_host(x, cd, cn, top2=False, chunk_rows=65536)
    torch.cuda.synchronize()
    assert torch.equal(idx_d, idx_h) and torch.equal(dist_d, dist_h)
    t_d = ops.assign_top2(_dev(x), cd, 0.8, cn)
    t_h = ops.assign_from_host(x, cd, cn, top2=True, ratio=0.8, chunk_rows=100_000)
    torch.cuda.synchronize()
    for a, b in zip(t_d, t_h):
        assert torch.equal(a, b)
    from image_classification_amd import vq
    code, dist = vq(x, c.astype(np.float64))
    assert np.array_equal(code, idx_d.cpu().numpy())
    # large inputs are checked for infs / NaNs through the result instead of a host scan
    bad = x.copy()
    bad[250_000, 5] = np.nan
    with pytest.raises(ValueError):
        vq(bad, c.astype(np.float64))
    bad[250_000, 5] = np.inf
    with pytest.raises(ValueError):
        vq(bad, c.astype(np.float64))


@pytest.mark.parametrize("n,k", [(98_304, 256), (131_089, 255), (200_000, 100), (150_001, 33),
                                 (120_000, 1), (300_000, 129)])
def test_codebooks_resident_in_lds_equal_the_direct_search(n, k):
    """From about 100 000 rows on a codebook of at most 256 words stays in LDS whole and every
    wave walks 32-row tiles on its own (k_top2_resident): last tiles that are partly filled, waves
    with one tile more than others, a single stage, one codeword, duplicates (ties go to the
    lowest index), top-1, top-2 and hinted calls."""
    rng = np.random.default_rng(n + k)
    c = sift_like(k, seed=400 + k % 13)
    if k > 8:
        dup = rng.integers(0, k, 8)
        c[dup[:4]] = c[dup[4:]]
    x = sift_like(n, seed=k % 17)
    x[:k] = c                                        # exact hits, some of them on duplicates
    _same_as_direct(x, c)
    xd, cd = _dev(x), _dev(c)
    idx, dist = ops.assign_top1(xd, cd)
    stale = torch.from_numpy(rng.integers(-1, k + 1, size=n).astype(np.int32)).cuda()
    for hint in (idx, stale):
        hi, hd = ops.assign_top1(xd, cd, hint=hint)
        assert torch.equal(hi, idx) and torch.equal(hd, dist)


@pytest.mark.parametrize("k", [256, 257, 16384, 16385])
def test_scan_order_boundaries_equal_the_direct_search(k):
    """Codebooks of 257 ... 16384 words are scanned in ascending order of their distance to the
    mean, smaller and larger ones as stored: on both sides of both limits the answer is the direct
    search's, with a few codewords duplicated so that ties must go to the lowest codeword index."""
    rng = np.random.default_rng(k)
    c = sift_like(k, seed=300 + k % 7)
    dup = rng.integers(0, k, 12)
    c[dup[:6]] = c[dup[6:]]
    x = np.concatenate([sift_like(2500, seed=k % 11), c[dup] + 0.0])      # exact hits on duplicates
    _same_as_direct(x, c)


def test_non_finite_codebook_entries_are_memory_safe():
    """vq() refuses non-finite input on the host; the C ABI takes what it is given.  A codebook
    with NaN / infinite entries (any bit pattern, the all-ones NaN included) must still be laid out
    in a scan order that is a permutation -- the norms are ranked as bit patterns -- and every
    index that comes back must be -1 or a codeword index."""
    x = sift_like(5000, seed=71)
    c = sift_like(600, seed=72)
    c[5, 3] = np.nan
    c[17] = np.inf
    c[40:42].view(np.uint32)[:] = 0xFFFFFFFF
    xd, cd = _dev(x), _dev(c)
    aux = ops.codebook_aux(cd)
    a = aux.cpu().numpy()
    kpad = 640
    perm = a[256 + kpad + 128:256 + 2 * kpad + 128].view(np.int32)
    rank = a[256 + 2 * kpad + 128:256 + 3 * kpad + 128].view(np.int32)
    assert sorted(perm[:600]) == list(range(600)) and np.array_equal(rank[perm[:600]], np.arange(600))
    for out in (ops.assign_top2(xd, cd, 0.8, aux), ops.assign_top2(xd, cd, 0.8)):
        torch.cuda.synchronize()
        for idx in (out[0], out[2]):
            assert bool(((idx >= -1) & (idx < 600)).all())
    # with the damaged codewords removed the search is the ordinary one again
    keep = np.ones(600, bool)
    keep[[5, 17, 40, 41]] = False
    _same_as_direct(x, c[keep])


def test_gathered_parts_path_equals_device_path():
    # a list of per-image arrays (ragged, mixed float dtypes, one empty) is gathered by worker
    # threads into page-locked staging buffers and uploaded chunk by chunk
    rng = np.random.default_rng(5)
    sizes = [int(v) for v in rng.integers(1, 9000, size=40)] + [0, 70000]
    parts = [sift_like(s, seed=100 + i) if s else np.empty((0, 128), np.float32)
             for i, s in enumerate(sizes)]
    parts[3] = parts[3].astype(np.float64)
    # what the native gather does not read as it is (strided views, Fortran order, half floats,
    # integers) is converted first, and the caller's list is left alone
    parts[5] = np.ascontiguousarray(np.repeat(parts[5], 2, axis=0))[::2]
    parts[7] = np.asfortranarray(parts[7])
    parts[9] = np.round(parts[9]).astype(np.uint8)
    parts[11] = parts[11].astype(np.float16)
    kept = list(parts)
    c = sift_like(600, seed=63)
    cd = _dev(c)
    cn = ops.codebook_aux(cd)
    allx = np.concatenate([p.astype(np.float32) for p in parts])
    ref = ops.assign_top2(_dev(allx), cd, 0.8, cn)
    for chunk in (1000, 50_000, 131_072):
        got = ops.assign_from_host_parts(parts, cd, cn, top2=True, ratio=0.8, chunk_rows=chunk)
        torch.cuda.synchronize()
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
    idx, dist = ops.assign_from_host_parts(parts, cd, cn, top2=False, want_dist=False)
    assert dist is None and torch.equal(idx, ref[0])
    assert all(a is b for a, b in zip(parts, kept))
    with pytest.raises(ValueError):
        ops.assign_from_host_parts([np.zeros((3, 64), np.float32)], cd, cn)


@pytest.mark.parametrize("k,norm", [(300, "l1"), (70_000, "l2"), (1000, "power_l2")])
def test_pipelined_quantise_equals_the_three_plain_steps(k, norm):
    # search -> histogram -> download as one pipeline (blocks of histograms travel down while
    # later rows are searched) must give the bytes of the three separate steps; small chunks force
    # many blocks, images straddle chunk boundaries, some are empty, and 70 000 words make one
    # image's histogram (560 KB) cut the 16 MB download pieces at odd places
    rng = np.random.default_rng(k)
    sizes = [int(v) for v in rng.integers(0, 900, size=60)]
    sizes[0] = 0
    sizes[17] = 0
    sizes[-1] = 0
    sizes[30] = 9000                       # one image spanning several chunks
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    x = sift_like(int(offsets[-1]), n_modes=128, seed=3)
    c = sift_like(k, n_modes=128, seed=4)
    parts = [x[offsets[i]:offsets[i + 1]].astype(np.float64 if i % 3 == 0 else np.float32)
             for i in range(len(sizes)) if sizes[i]]
    cd = _dev(c)
    aux = ops.codebook_aux(cd)
    codes, hists = ops.quantize_parts_to_host(parts, offsets, cd, aux, k, norm, 0.5, chunk_rows=4096)
    ref_codes, _ = ops.assign_top1(_dev(x), cd, aux, want_dist=False)
    ref_h = ops.hist(ref_codes, _dev_i64(offsets), k, norm, 0.5).cpu().numpy()
    assert torch.equal(codes, ref_codes)
    assert hists.dtype == np.float64 and hists.shape == (len(sizes), k)
    assert np.array_equal(hists, ref_h)
    # no descriptors at all: every histogram is an all-zero row
    codes0, h0 = ops.quantize_parts_to_host([], np.zeros(4, np.int64), cd, aux, k, norm)
    assert codes0.numel() == 0 and h0.shape == (3, k) and not h0.any()


def _dev_i64(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to("cuda")


def test_staged_download_equals_plain_download():
    # results of tens of MB come back through page-locked staging buffers
    t = torch.arange(0, 9_000_000, dtype=torch.float64, device="cuda").reshape(3, -1) * 0.5
    got = ops.to_host(t)
    assert got.dtype == np.float64 and got.shape == (3, 3_000_000)
    assert np.array_equal(got, t.cpu().numpy())
    odd = torch.arange(0, 40_000_003, dtype=torch.uint8, device="cuda")   # not a whole number of chunks
    assert np.array_equal(ops.to_host(odd), odd.cpu().numpy())
    small = torch.arange(10, dtype=torch.int32, device="cuda")
    assert np.array_equal(ops.to_host(small), np.arange(10, dtype=np.int32))


def test_hinted_search_gives_the_same_answer():
    # guesses of every quality: exact, stale (previous codebook), random, out of range, worst
    x = sift_like(200_000, seed=71)
    c = sift_like(3000, n_modes=1024, seed=72)
    xd, cd = _dev(x), _dev(c)
    cn = ops.codebook_aux(cd)
    idx, dist = ops.assign_top1(xd, cd, cn)
    rng = np.random.default_rng(1)
    stale_cb = _dev(c + rng.normal(0, 8.0, size=c.shape).astype(np.float32))
    stale, _ = ops.assign_top1(xd, stale_cb)
    rand = torch.from_numpy(rng.integers(0, 3000, size=len(x)).astype(np.int32)).cuda()
    bad = torch.full((len(x),), -7, dtype=torch.int32, device="cuda")
    bad[::2] = 3000
    far = torch.cdist(xd[:5000], cd).argmax(1).int()
    for hint in (idx, stale, rand, bad):
        hi, hd = ops.assign_top1(xd, cd, cn, hint=hint)
        assert torch.equal(hi, idx) and torch.equal(hd, dist)
    hi, hd = ops.assign_top1(xd[:5000], cd, cn, hint=far)
    assert torch.equal(hi, idx[:5000]) and torch.equal(hd, dist[:5000])
    # ties and exact hits keep the lowest index
    c2 = np.concatenate([c[:100], c[:100]])
    i_ref, _ = ops.assign_top1(_dev(c[:100]), _dev(c2))
    i_hint, d_hint = ops.assign_top1(_dev(c[:100]), _dev(c2),
                                     hint=(torch.arange(100, device="cuda") + 100).int())
    assert torch.equal(i_hint, i_ref) and torch.all(d_hint == 0)
    with pytest.raises(ValueError):
        ops.assign_top1(xd, cd, cn, hint=idx[:10])


@pytest.mark.parametrize("n,k,d", [(1000, 300, 128), (37, 5, 64), (513, 1, 128), (2000, 700, 33)])
def test_direct_search_vs_oracle(n, k, d):
    # the slow kernel without the norm expansion: the strict rule holds even for near ties
    rng = np.random.default_rng(n + k)
    x = (rng.standard_normal((n, d)) * 40).astype(np.float32)
    c = (rng.standard_normal((k, d)) * 40).astype(np.float32)
    if k > 4:
        c[k - 1] = c[0]
        x[:3] = c[:3]
    i1, d1, i2, d2, mask = (t.cpu().numpy() for t in ops.assign_top2_exact(_dev(x), _dev(c), 0.8))
    rd, ri = oracle.topk(x, c, min(3, k))
    if k == 1:
        assert (i1 == 0).all() and (i2 == -1).all() and np.isinf(d2).all()
        assert oracle.check_close(d1, rd[:, 0], abs_floor=1e-4) <= DIST_REL
        return
    must, bad, agree = oracle.check_indices(i1, ri, rd, 0)
    assert bad == 0 and agree > 0.999
    assert oracle.check_close(d1[i1 == ri[:, 0]], rd[i1 == ri[:, 0], 0], abs_floor=1e-4) <= DIST_REL
    if k >= 3:
        must, bad, agree = oracle.check_indices(i2, ri, rd, 1)
        assert bad == 0


@pytest.mark.parametrize("n,k,d", [(3000, 600, 400), (2500, 600, 1937), (700, 70, 129),
                                   (40_000, 300, 256), (300, 5000, 640), (1000, 40, 2048),
                                   (70_000, 2100, 512)])
def test_rows_wider_than_128_columns(n, k, d):
    """Rows of more than 128 columns are searched in 128-column slabs with the accumulators
    carried from slab to slab (the 1937-column face signatures of
    vision/facerec/face_indexing.py:25 at :141 / :158; 600 clusters is face_index_builder.py:44's
    default): the result must equal the direct float32 search bit for bit on every row, and the
    float64 oracle under the gap rule."""
    rng = np.random.default_rng(d)
    centres = rng.standard_normal((64, d)).astype(np.float32) * 3
    c = (centres[rng.integers(0, 64, k)] + rng.standard_normal((k, d))).astype(np.float32)
    x = (centres[rng.integers(0, 64, n)] + rng.standard_normal((n, d))).astype(np.float32)
    x[:20] = c[:20] + rng.normal(0, 1e-3, (20, d)).astype(np.float32)   # near-duplicates
    f1, f2 = _same_as_direct(x, c)
    assert f1 <= n // 10, (f1, n)      # the second look stays the exception at any width
    ns = min(n, 1500)
    rd, ri = oracle.topk(x[:ns], c, 3)
    i1, d1, i2, d2, _ = _run_top2(x[:ns], c)
    for got, rank in ((i1, 0), (i2, 1)):
        must, bad, agree = oracle.check_indices(got, ri, rd, rank)
        assert bad == 0 and agree >= 0.999
    assert oracle.check_close(d1, rd[:, 0]) <= DIST_REL
    # the hinted search (k-means iterations) gives the same answer
    xd, cd = _dev(x), _dev(c)
    plain, pd = ops.assign_top1(xd, cd)
    hinted, hd = ops.assign_top1(xd, cd, hint=plain)
    assert torch.equal(plain, hinted) and torch.equal(pd, hd)
    # centroid sums of wide rows
    sums, counts = ops.centroid_partials(xd, plain, k)
    lab = plain.cpu().numpy()
    ref = np.zeros((k, d))
    np.add.at(ref, lab, x.astype(np.float64))
    assert np.array_equal(counts.cpu().numpy(), np.bincount(lab, minlength=k))
    assert np.allclose(sums.cpu().numpy(), ref, rtol=1e-12, atol=1e-9)


def test_one_dimensional_near_ties_are_strict_on_both_paths():
    # rows far closer to their neighbours than float32 resolves from the norms
    rng = np.random.default_rng(38)
    x = (rng.standard_normal((2059, 1)) * 300).astype(np.float32)
    c = (rng.standard_normal((430, 1)) * 300).astype(np.float32)
    rd, ri = oracle.topk(x, c, 3)
    e = [t.cpu().numpy() for t in ops.assign_top2_exact(_dev(x), _dev(c), 0.8)]
    for col, rank in ((0, 0), (2, 1)):
        must, bad, _ = oracle.check_indices(e[col], ri, rd, rank)
        assert bad == 0
    # the matrix-core path hands such rows to the direct search: the strict rule holds for it too
    f = [t.cpu().numpy() for t in ops.assign_top2(_dev(x), _dev(c), 0.8)]
    for col, rank in ((0, 0), (2, 1)):
        must, bad, _ = oracle.check_indices(f[col], ri, rd, rank)
        assert bad == 0
        assert np.array_equal(f[col], e[col])


def test_errors_are_loud():
    x = _dev(np.zeros((4, 128)))
    with pytest.raises(ValueError):
        ops.assign_top1(x, _dev(np.zeros((3, 64))))
    with pytest.raises(ValueError, match="at most 2048"):     # refused before any launch
        ops.assign_top1(_dev(np.zeros((4, 2100))), _dev(np.zeros((3, 2100))))
    with pytest.raises(ValueError):
        ops.assign_top1(x.cpu(), x)
    # an auxiliary block that belongs to a codebook of another shape is refused before any launch
    c300, c400 = _dev(sift_like(300, seed=1)), _dev(sift_like(400, seed=2))
    with pytest.raises(ValueError, match="c_aux"):
        ops.assign_top1(x, c400, ops.codebook_aux(c300))
    with pytest.raises(ValueError, match="c_aux"):
        ops.assign_top2(x, c400, 0.8, ops.codebook_aux(c300)[:-1])


def test_reserved_workspace_is_not_reallocated():
    # a workspace sized up front: calls that fit it leave its size (and the device's free memory)
    # alone; a larger call grows it, and the results are the same either way
    x = _dev(sift_like(50_000, seed=61))
    c = _dev(sift_like(700, seed=62))
    before = ops.assign_top2(x, c, 0.8)
    torch.cuda.synchronize()
    need = ops.workspace_bytes(0)
    assert need > 0
    ops.reserve_workspace(0, 4 * need)
    size = ops.workspace_bytes(0)
    assert size >= 4 * need
    free0 = torch.cuda.mem_get_info()[0]
    for n in (50_000, 20_000, 100_000):
        got = ops.assign_top2(x[:n] if n <= 50_000 else torch.cat([x, x]), c, 0.8)
        torch.cuda.synchronize()
        assert ops.workspace_bytes(0) == size
        if n == 50_000:
            assert all(torch.equal(a, b) for a, b in zip(got, before))
    assert torch.cuda.mem_get_info()[0] == free0
    ops.reserve_workspace(0, 1)                       # grow only
    assert ops.workspace_bytes(0) == size

"""The decomposition of a search into workgroups, checked on the CPU: ic_debug_plan gives the
plan the library would launch, and the kernels' own index arithmetic (k_top2_partial /
k_top2_wide: which units of the line of (row tile, stage) pairs a workgroup walks, which piece of
its row tile that is, which record slot a piece writes) is replayed here in Python.  Every stage
of every row tile must be scanned exactly once, pieces must be numbered 0 .. pieces-1 without
gaps, no piece index may reach max_pieces, and the record slots of all (row, list) pairs must be
distinct and inside the partial buffer."""
import ctypes

import numpy as np
import pytest

from image_classification_amd import _lib

FIELDS = ("waves", "stage_k", "rows_per_wg", "stages", "tail_span", "row_tiles", "full_tiles",
          "tail_quota", "max_pieces", "tail_wgs", "wide_pieces", "records", "resident",
          "resident_wgs", "tail_long")


def run_start(p, g):
    """First unit of run g: the first tail_long runs have tail_quota units, the others one less."""
    return g * p["tail_quota"] - max(0, g - p["tail_long"])


def run_of(p, u):
    q, long_units = p["tail_quota"], p["tail_long"] * p["tail_quota"]
    if u < long_units or q < 2:
        return u // q
    return p["tail_long"] + (u - long_units) // (q - 1)


def plan(n, k, d=128, cus=256):
    out = (ctypes.c_int64 * 15)()
    rc = _lib.load().ic_debug_plan(n, k, d, cus, out)
    assert rc == 0, _lib.last_error()
    return dict(zip(FIELDS, (int(v) for v in out)))


def tail_order(n, k, d=128, cus=256):
    buf = (ctypes.c_int32 * 1024)()
    cnt = ctypes.c_int(0)
    rc = _lib.load().ic_debug_plan_order(n, k, d, cus, buf, 1024, ctypes.byref(cnt))
    assert rc == 0, _lib.last_error()
    return [int(buf[i]) for i in range(cnt.value)]


def walk(p, n, order=None):
    """[(row tile, first stage, stages, piece)] of every workgroup run, as the kernels derive them
    from blockIdx."""
    runs = []
    S, span, q = p["stages"], p["tail_span"], p["tail_quota"]
    full = p["full_tiles"]
    if p["wide_pieces"]:
        P = p["wide_pieces"]
        grid = (p["row_tiles"] + 7) // 8 * 8 * P
        for g in range(grid):
            xcd, place = g & 7, g >> 3
            tile = place // P * 8 + xcd
            piece = place % P
            st = piece * q
            left = min(q, S - st)
            if tile * p["rows_per_wg"] >= n:
                continue
            assert left >= 1
            runs.append((tile, st, left, piece))
        return runs
    for g in range(full):
        runs.append((g, 0, S, 0))
    tail_units = (p["row_tiles"] - full) * span
    if order:
        assert sorted(order) == list(range(p["tail_wgs"])), "the hand-out is not a permutation"
    for b in range(p["tail_wgs"]):
        g = order[b] if order else b
        u = run_start(p, g)
        tt = u // span
        rest = tail_units - u
        tile = full + tt
        st = u - tt * span
        left = min(rest, q if g < p["tail_long"] else q - 1)
        if span != S:
            left = min(left, S - st)
        piece = g - run_of(p, tt * span)
        while left > 0:
            item = min(S - st, left)
            assert item >= 1
            runs.append((tile, st, item, piece))
            left -= item
            tile, st, piece = tile + 1, 0, 0
    return runs


def check(n, k, d=128, cus=256):
    p = plan(n, k, d, cus)
    S = p["stages"]
    assert p["rows_per_wg"] == p["waves"] * 32 and p["stage_k"] in (64, 128)
    assert S == -(-k // p["stage_k"]) and p["row_tiles"] == -(-n // p["rows_per_wg"])
    assert 0 <= p["full_tiles"] <= p["row_tiles"]
    covered = np.zeros((p["row_tiles"], S), dtype=np.int32)
    pieces = [set() for _ in range(p["row_tiles"])]
    for tile, st, cnt, piece in walk(p, n, tail_order(n, k, d, cus)):
        assert 0 <= tile < p["row_tiles"] and 0 <= st and st + cnt <= S
        covered[tile, st:st + cnt] += 1
        assert piece not in pieces[tile], "two runs write the same records"
        pieces[tile].add(piece)
    assert (covered == 1).all(), "a stage is scanned %s times" % sorted(set(covered.ravel()))
    split_row0 = min(n, p["full_tiles"] * p["rows_per_wg"])
    for tile, got in enumerate(pieces):
        assert got == set(range(len(got))) and len(got) <= p["max_pieces"]
        if tile < p["full_tiles"]:
            assert got == {0}
        else:
            # what k_rescore takes the number of lists of a tail row tile to be
            tt = tile - p["full_tiles"]
            span = p["tail_span"]
            first = run_of(p, tt * span)
            assert len(got) == run_of(p, (tt + 1) * span - 1) - first + 1
    # record slots (part_slot): lists 0 and 1 for every row, the others only for the tail's rows
    assert p["records"] == 2 * n + (2 * p["max_pieces"] - 2) * (n - split_row0)
    if p["resident"]:
        check_resident(p, n, k)
    return p


def check_resident(p, n, k):
    """k_top2_resident: the codebook stays in LDS, every wave walks 32-row tiles on its own
    (tile = iteration * 16 G + slot, slot = CU first, then SIMD, then wave within the SIMD).  The
    rest of the plan must read as row tiles scanned whole: two lists per row."""
    G = p["resident_wgs"]
    assert k <= 256 and p["waves"] == 16 and p["stage_k"] == 128 and p["max_pieces"] == 1
    assert p["full_tiles"] == p["row_tiles"] and p["tail_wgs"] == 0 and p["records"] == 2 * n
    n_tiles = -(-n // 32)
    assert 1 <= G and 4 * G <= n_tiles + 3        # no workgroup without a tile
    taken = np.zeros(n_tiles, dtype=np.int32)
    per_simd = {}
    for g in range(G):
        for wave in range(16):
            wt = (wave >> 2) * (4 * G) + (wave & 3) * G + g
            while wt < n_tiles:
                taken[wt] += 1
                per_simd[(g, wave & 3)] = per_simd.get((g, wave & 3), 0) + 1
                wt += 16 * G
    assert (taken == 1).all()
    loads = list(per_simd.values()) + [0] * (4 * G - len(per_simd))
    assert max(loads) - min(loads) <= 1, "tiles are not spread evenly over the SIMDs"


@pytest.mark.parametrize("n,k,d", [
    (100_000, 1000, 128),        # one partly filled round: stream-K runs across row tiles
    (1_000_000, 10_000, 128),    # seven whole rounds and a tail
    (4000, 2_000_000 // 64, 128),  # few row tiles, long codebook: aligned pieces (scaled down)
    (1, 1, 128), (31, 33, 128), (513, 64, 128), (256, 8192, 128), (512, 8191, 128),
    (100_000, 600, 1937), (1000, 600, 1937), (70_000, 2100, 512), (300, 5000, 640),
    (40_000, 300, 256), (2500, 600, 2048), (1_000_000, 256, 128), (140_000, 8400, 128),
])
def test_named_shapes(n, k, d):
    check(n, k, d)


def test_random_shapes():
    rng = np.random.default_rng(12)
    for _ in range(400):
        n = int(rng.choice([1, 17, 255, 256, 257, 5000, 70_000, 131_072, 400_000])) + int(rng.integers(0, 600))
        k = int(rng.choice([1, 2, 63, 64, 65, 700, 4000, 8191, 8192, 20_000])) + int(rng.integers(0, 130))
        d = int(rng.choice([128, 128, 128, 36, 129, 256, 512, 640, 1024, 1937, 2048]))
        cus = int(rng.choice([256, 256, 304, 64, 1]))
        if n * -(-k // 64) > 3_000_000:      # keep the coverage matrix small
            n = max(1, 3_000_000 // -(-k // 64))
        check(n, k, d, cus)


def test_wide_rows_share_an_xcd():
    # the pieces of a row tile of the wide search sit at consecutive places of one XCD
    p = plan(100_000, 600, 1937)
    P = p["wide_pieces"]
    assert P == 10 and p["tail_quota"] == 1 and p["full_tiles"] == 0
    by_tile = {}
    grid = (p["row_tiles"] + 7) // 8 * 8 * P
    for g in range(grid):
        tile = (g >> 3) // P * 8 + (g & 7)
        by_tile.setdefault(tile, []).append(g)
    for tile, ids in by_tile.items():
        assert len({g & 7 for g in ids}) == 1
        places = sorted(g >> 3 for g in ids)
        assert places == list(range(places[0], places[0] + P))


def test_tail_runs_are_dealt_to_the_xcds_by_first_stage():
    # BASELINE config 2: 162 row tiles of the last round as 256 runs of 50 stages; the workgroups
    # of one XCD (ids 8 apart) get runs whose first stages follow each other
    n, k = 1_000_000, 10_000
    p = plan(n, k)
    order = tail_order(n, k)
    assert len(order) == p["tail_wgs"] == 256 and p["tail_span"] == p["stages"]
    S = p["stages"]
    last = -1
    for xcd in range(8):
        mine = [order[b] for b in range(len(order)) if (p["full_tiles"] + b) % 8 == xcd]
        # inside an XCD's window the long runs come first, each kind by first stage
        kinds = [g < p["tail_long"] for g in mine]
        assert kinds == sorted(kinds, reverse=True)
        for kind in (True, False):
            starts = [run_start(p, g) % S for g in mine if (g < p["tail_long"]) == kind]
            assert starts == sorted(starts)
        starts = [run_start(p, g) % S for g in mine]
        assert min(starts) >= last
        assert max(starts) - min(starts) <= S // 4    # a window of the codebook, not all of it
        last = max(starts)
    # aligned pieces and the wide search keep the plain order
    assert tail_order(4000, 2_000_000 // 64) == [] and tail_order(100_000, 600, 1937) == []


def test_runs_of_two_lengths_fill_every_slot():
    # BASELINE config 1: 391 row tiles x 16 stages = 6256 units on 512 slots: 112 runs of 13
    # stages and 400 of 12 instead of 482 runs of 13, and no CU (workgroups b and b + 256) gets
    # two long ones
    n, k = 100_000, 1000
    p = plan(n, k)
    assert (p["tail_wgs"], p["tail_quota"], p["tail_long"], p["full_tiles"]) == (512, 13, 112, 0)
    order = tail_order(n, k)
    assert sorted(order) == list(range(512))
    long_at = [order[b] < p["tail_long"] for b in range(512)]
    assert sum(long_at) == 112
    assert not any(long_at[b] and long_at[b + 256] for b in range(256))
    lengths = [13 if g < 112 else 12 for g in range(512)]
    assert sum(lengths) == 391 * 16 and run_start(p, 511) + 12 == 391 * 16
    # a line that divides evenly keeps runs of one length
    q = plan(256 * 256, 64 * 16)
    assert q["tail_long"] == q["tail_wgs"]


def test_small_codebooks_stay_in_lds_from_many_rows_on():
    assert plan(1_000_000, 256)["resident"] == 1 and plan(1_000_000, 256)["resident_wgs"] == 256
    assert plan(200_000, 64)["resident"] == 1
    assert plan(120_000, 256, cus=64)["resident_wgs"] == 64
    assert plan(20_000, 256)["resident"] == 0          # few rows: streamed as any other codebook
    assert plan(1_000_000, 257)["resident"] == 0 and plan(1_000_000, 256, d=256)["resident"] == 0


def test_bad_arguments_are_refused():
    out = (ctypes.c_int64 * 15)()
    lib = _lib.load()
    assert lib.ic_debug_plan(0, 10, 128, 256, out) != 0
    assert lib.ic_debug_plan(10, 0, 128, 256, out) != 0
    assert lib.ic_debug_plan(10, 10, 4096, 256, out) != 0
    assert lib.ic_debug_plan(10, 10, 128, 256, None) != 0

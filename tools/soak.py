#!/usr/bin/env python3
"""Randomised comparison of the MFMA search with the direct float32 search, both on the device:
soak.py [cases] [seed].  Shapes cover both workgroup geometries, K-splits, padded dimensions,
exact ties, duplicated codewords and data far from the origin; indices, distances and the ratio
mask must EQUAL the direct search's on every row, hinted and unhinted searches must be identical,
and the record / merge path must equal the single search."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_classification_amd import ops  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    rng = np.random.default_rng(seed)
    for case in range(cases):
        n = int(rng.choice([1, 31, 64, 513, 4096, 20000, 140000])) + int(rng.integers(0, 300))
        k = int(rng.choice([1, 2, 33, 200, 1000, 2047, 2048, 2100, 5000, 9000])) + int(rng.integers(0, 50))
        d = int(rng.choice([128, 128, 128, 64, 100, 36, 3, 200, 640, 1937]))
        if d > 128:
            n = min(n, 20000)     # wide rows: keep a case in the tens of milliseconds
        kind = int(rng.integers(0, 3))
        if kind == 0:      # real-valued, non-negative, norm a few hundred (SIFT-like range)
            x = torch.rand((n, d), device="cuda", generator=g) * 60
            c = torch.rand((k, d), device="cuda", generator=g) * 60
        elif kind == 1:    # small integers: exact ties everywhere
            x = torch.randint(0, 4, (n, d), device="cuda", generator=g).float()
            c = torch.randint(0, 4, (k, d), device="cuda", generator=g).float()
        else:              # signed, unit scale, duplicated codewords and exact hits
            x = torch.randn((n, d), device="cuda", generator=g)
            c = torch.randn((k, d), device="cuda", generator=g)
            if k > 3:
                c[k // 2] = c[0]
                c[k - 1] = c[1]
            x[: min(n, k)] = c[: min(n, k)]
        if case % 4 == 3:  # the same rows far from the origin: norms dwarf the distances
            shift = float(rng.choice([50.0, 1000.0]))
            x = x + shift
            c = c + shift
        i1, d1, i2, d2, m = ops.assign_top2(x, c, 0.8)
        e1, ed1, e2, ed2, em = ops.assign_top2_exact(x, c, 0.8)
        tag = "case %d: n=%d k=%d d=%d kind=%d" % (case, n, k, d, kind)
        for name, a, b in (("i1", i1, e1), ("i2", i2, e2), ("mask", m, em)):
            bad = int((a != b).sum())
            assert bad == 0, tag + ": %s differs from the direct search on %d rows" % (name, bad)
        for name, a, b in (("d1", d1, ed1), ("d2", d2, ed2)):
            bad = int((a.view(torch.int32) != b.view(torch.int32)).sum())
            assert bad == 0, tag + ": %s differs in bits on %d rows" % (name, bad)
        t1, td1 = ops.assign_top1(x, c)
        assert torch.equal(t1, e1) and torch.equal(td1.view(torch.int32), ed1.view(torch.int32)), \
            tag + ": top-1 search differs from the direct search"
        # hinted search: identical to the unhinted one, whatever the hint
        hint = e1.clone()
        hint[::3] = torch.randint(0, k, (hint[::3].numel(),), device="cuda", generator=g).int()
        hint[1::7] = -5
        h1, hd1 = ops.assign_top1(x, c, hint=hint)
        assert torch.equal(h1, i1) and torch.equal(hd1, d1), tag + ": hinted search differs"
        # index only, with and without the hint: the same indices
        o1, _ = ops.assign_top1(x, c, want_dist=False)
        o2, _ = ops.assign_top1(x, c, want_dist=False, hint=hint)
        assert torch.equal(o1, e1) and torch.equal(o2, e1), tag + ": index-only search differs"
        # record + merge over two ragged index shards
        if k >= 2:
            cut = int(rng.integers(1, k))
            recs = torch.stack([ops.assign_top2_rec(x, c[:cut].contiguous()),
                                ops.assign_top2_rec(x, c[cut:].contiguous())]).contiguous()
            m1, md1, m2, md2, mm = ops.top2_merge(recs, [0, cut], 0.8)
            # shards are merged by Euclidean distance: equal unless two different squared
            # distances round to the same square root
            differ = (m1 != i1.long()) & (md1 != d1)
            assert int(differ.sum()) == 0, tag + ": merged shards differ"
        if case % 20 == 19:
            print("%d cases ok" % (case + 1), flush=True)
    torch.cuda.synchronize()
    print("SOAK_OK %d cases" % cases)


if __name__ == "__main__":
    main()

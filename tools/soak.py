#!/usr/bin/env python3
"""Randomised comparison of the MFMA search with the direct float32 search, both on the device:
soak.py [cases] [seed].  Shapes cover both workgroup geometries, K-splits, padded dimensions,
exact ties and duplicated codewords; indices must agree wherever the direct search's gap between
the candidates exceeds the float32 resolution of the norm expansion, hinted and unhinted searches
must be identical, and the record / merge path must equal the single search."""
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
    worst = 0.0
    for case in range(cases):
        n = int(rng.choice([1, 31, 64, 513, 4096, 20000, 140000])) + int(rng.integers(0, 300))
        k = int(rng.choice([1, 2, 33, 200, 1000, 2047, 2048, 2100, 5000, 9000])) + int(rng.integers(0, 50))
        d = int(rng.choice([128, 128, 128, 64, 100, 36, 3]))
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
        i1, d1, i2, d2, m = ops.assign_top2(x, c, 0.8)
        e1, ed1, e2, ed2, em = ops.assign_top2_exact(x, c, 0.8)
        # resolution of the expansion in squared distance, as a distance gap at this row
        xn = (x * x).sum(1).sqrt()
        cn = (c * c).sum(1).sqrt().max()
        band_sq = 3 * 7.7e-6 * (cn * cn + 2 * xn * cn) + 1e-30
        tag = "case %d: n=%d k=%d d=%d kind=%d" % (case, n, k, d, kind)
        # best: every candidate the direct search ranks below must be clearly worse, else free
        if k >= 2:
            clear1 = (ed2 * ed2 - ed1 * ed1) > band_sq
        else:
            clear1 = torch.ones_like(ed1, dtype=torch.bool)
        bad1 = int(((i1 != e1) & clear1).sum())
        assert bad1 == 0, tag + ": %d best neighbours differ outside the float32 band" % bad1
        same1 = i1 == e1
        # the two kernels add the 128 squared differences in different orders
        assert torch.allclose(d1[same1], ed1[same1], rtol=2e-6, atol=1e-6), \
            tag + ": distances of equal winners differ"
        if kind == 1 and d == 128:
            assert torch.equal(i1, e1) and torch.equal(i2, e2), tag + ": exact data must agree exactly"
        agree = float(same1.float().mean())
        worst = max(worst, 1 - agree)
        # hinted search: identical to the unhinted one, whatever the hint
        hint = e1.clone()
        hint[::3] = torch.randint(0, k, (hint[::3].numel(),), device="cuda", generator=g).int()
        hint[1::7] = -5
        h1, hd1 = ops.assign_top1(x, c, hint=hint)
        assert torch.equal(h1, i1) and torch.equal(hd1, d1), tag + ": hinted search differs"
        # record + merge over two ragged index shards
        if k >= 2:
            cut = int(rng.integers(1, k))
            recs = torch.stack([ops.assign_top2_rec(x, c[:cut].contiguous()),
                                ops.assign_top2_rec(x, c[cut:].contiguous())]).contiguous()
            m1, md1, m2, md2, mm = ops.top2_merge(recs, [0, cut], 0.8)
            ok = (m1 == i1.long())
            assert int((~ok & clear1).sum()) == 0, tag + ": merged shards differ"
        if case % 20 == 19:
            print("%d cases ok, largest disagreement with the direct search inside the band %.2e"
                  % (case + 1, worst), flush=True)
    torch.cuda.synchronize()
    print("SOAK_OK %d cases" % cases)


if __name__ == "__main__":
    main()

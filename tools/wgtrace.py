#!/usr/bin/env python3
"""Timeline of the search kernel's workgroups: wgtrace.py N K  (needs `make trace` in csrc/).

Runs one search with the instrumented library and prints, per CU, when each of its workgroups
entered, passed its first barrier, had its rows in registers and centred (or, with several row
tiles, changed row tile) and left (microseconds from the first entry)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tools", "libic_hip_trace.so")
os.environ["IC_AMD_LIBRARY"] = LIB
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from image_classification_amd import _lib, ops  # noqa: E402
from image_classification_amd.synthetic import sift_like_device  # noqa: E402


def main():
    n, k = int(sys.argv[1]), int(sys.argv[2])
    dev = torch.device("cuda", 0)
    x = sift_like_device(n, seed=1, device=dev)
    c = sift_like_device(k, seed=2, device=dev)
    aux = ops.codebook_aux(c)
    import time
    t_end = time.perf_counter() + 0.06      # the shader clock ramps with sustained load
    while time.perf_counter() < t_end:
        ops.assign_top1(x, c, aux, want_dist=False)
        torch.cuda.synchronize()
    buf = np.zeros((1 << 16, 8), dtype=np.uint64)
    rd = _lib.load().ic_trace_read
    rd.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert rd(buf.ctypes.data, buf.nbytes) == 0
    used = buf[:, 2] != 0
    t = buf[used]
    t0 = t[:, 0].min()
    us = lambda v: (v.astype(np.int64) - int(t0)) / 100.0   # 100 MHz clock
    hw = t[:, 3]
    xcc = (hw >> np.uint64(32)) & np.uint64(0xf)
    lo = hw & np.uint64(0xffffffff)
    cu = (lo >> np.uint64(8)) & np.uint64(0xf)
    sh = (lo >> np.uint64(12)) & np.uint64(0x1)
    se = (lo >> np.uint64(13)) & np.uint64(0x7)
    place = (xcc * 64 + se * 16 + sh * 8 + cu).astype(np.int64)
    start, bar, end = us(t[:, 0]), us(t[:, 1]), us(t[:, 2])
    sw0 = np.where(t[:, 4] != 0, us(t[:, 4]), np.nan)
    sw1 = np.where(t[:, 5] != 0, us(t[:, 5]), np.nan)
    print("workgroups %d  first entry -> last exit %.1f us" % (len(t), end.max()))
    print("entry:  min %.1f  median %.1f  max %.1f" % (start.min(), np.median(start), start.max()))
    print("entry -> first barrier: median %.1f  max %.1f us" % (np.median(bar - start), (bar - start).max()))
    print("duration: median %.1f  max %.1f us" % (np.median(end - start), (end - start).max()))
    cyc = (t[:, 7].astype(np.int64) - t[:, 6].astype(np.int64)) / np.maximum(end - start, 1e-3)
    print("shader clock while the workgroups ran: median %.0f MHz (min %.0f, max %.0f)"
          % (np.median(cyc), cyc.min(), cyc.max()))
    if np.isfinite(sw0).any():
        d = sw1 - sw0
        print("rows landed -> centred (workgroups with several row tiles: the change of row tile): median %.1f  max %.1f us, %d workgroups"
              % (np.nanmedian(d), np.nanmax(d), int(np.isfinite(d).sum())))
    per_cu = {}
    for i in range(len(t)):
        per_cu.setdefault(int(place[i]), []).append((start[i], bar[i], sw0[i], sw1[i], end[i]))
    counts = np.bincount([len(v) for v in per_cu.values()])
    print("CUs used %d; workgroups per CU histogram %s" % (len(per_cu), counts.tolist()))
    busiest = sorted(per_cu.items(), key=lambda kv: -max(e[-1] for e in kv[1]))[:4]
    for pl, evs in busiest:
        print("CU %d:" % pl)
        for e in sorted(evs):
            print("   entry %7.1f  barrier %7.1f  rows landed / centred %7.1f..%7.1f  exit %7.1f" % e)
    ends = np.array([max(e[-1] for e in v) for v in per_cu.values()])
    print("per-CU last exit: min %.1f  median %.1f  max %.1f" % (ends.min(), np.median(ends), ends.max()))


if __name__ == "__main__":
    main()

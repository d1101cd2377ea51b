#!/usr/bin/env python3
"""Benchmark of the descriptor -> codeword assignment path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload bow|knn|kmeans|video|logo|histknn]

One step = one pass of the hot path over one batch of synthetic descriptors resident in HBM.
The default workload is BASELINE config 2, "BoW quantisation + per-image L2-normalised histograms:
1M descriptors x 10k-word codebook, 128-d": ic_assign_top1 over [1M,128] x [10k,128] followed by
ic_hist (1000 images x 1000 descriptors, L2).  With N > 1 (launched by torch.distributed.run, one
rank per GPU) every rank processes its own 1M descriptors -- the path shards by rows and needs no
collective -- so scaling is weak and `value` is the sum over ranks.

The other workloads are the remaining BASELINE configs at their per-GPU size (they are parity /
sizing cases run by hand, not the line the driver records):
  knn     config 3: 4k queries vs a 2M-descriptor index split over N ranks, top-2 + ratio test,
          all-gather + merge of the per-rank records
  kmeans  config 4: one Lloyd iteration over 10M/8 rows, K = 65536, all-reduce of sums/counts
  video   config 5: 2000 frames x 2k descriptors per GPU, K = 10k, power-L2 histograms
  logo    config 3 in the reference's own direction: the 2M descriptors of 20 000 logo images are
          searched in the 4k-descriptor index of the query (top-2 + ratio test), matches are
          compacted per logo and verified by the batched homography RANSAC
  histknn the logo k-NN classifier's search: one query histogram vs 100 000 index histograms of
          10k bins, manhattan, 2 neighbours (HBM-bound: its roofline counts the index bytes)

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the fp32-MFMA distance /
top-2 kernel): algorithmic FLOPs per launch = 2 * N * K * D, divided by its average duration
measured with HIP events recorded on the launch stream inside the timed region; `traffic` is that
kernel's HBM bytes per launch from the newest profiles/*_<workload>_traffic.json (rocprofv3 --pmc
FETCH_SIZE / WRITE_SIZE passes of that workload, read side doubled as gfx950 requires).  `cpu_baseline`
(rank 0, N == 1 only) times the float64 numpy oracle on a bounded row sample of the same workload.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DIM = 128
FP32_MFMA_PEAK_TFLOPS = 157.3  # 256 CU x 256 FLOP/clk x 2.4 GHz
HBM_PEAK_GBS = 8000.0

WORKLOADS = {
    "bow": dict(rows=1_000_000, k=10_000, images=1000, norm="l2",
                name="BASELINE config 2: BoW quantisation + per-image L2 histograms, 1M descriptors "
                     "x 10k-word codebook x 128-d per GPU (1000 images x 1000 descriptors)"),
    "video": dict(rows=4_000_000, k=10_000, images=2000, norm="power_l2",
                  name="BASELINE config 5 (per-GPU share): 2000 frames x 2000 descriptors, 10k words, "
                       "assignment + per-frame power/L2 histograms"),
    "kmeans": dict(rows=1_250_000, k=65_536,
                   name="BASELINE config 4 (per-GPU share): one Lloyd iteration, 1.25M rows x 65536 "
                        "centroids x 128-d, all-reduce of sums/counts"),
    "logo": dict(rows=2_000_000, k=4000, images=20_000,
                 name="BASELINE config 3 (reference direction): 20 000 logo images x 100 descriptors "
                      "searched in a 4k-descriptor query index, 5 % planted matches, top-2 + ratio "
                      "test, per-logo match compaction, homography RANSAC (46 iterations, 3 px)"),
    "knn": dict(rows=4000, k=2_000_000,
                name="BASELINE config 3: 4k query descriptors vs 2M-descriptor index (split over "
                     "ranks), top-2 + ratio test, all-gather + merge"),
    "histknn": dict(rows=100_000, k=10_000,
                    name="histogram-space 2-NN (the logo k-NN classifier): one query histogram vs "
                         "100 000 L1-normalised 10k-bin histograms, manhattan, float64 distances"),
}


def cpu_baseline(x_sample, codebook, per_image, norm):
    """Oracle (numpy float64 restatement of scipy vq + np.histogram) on a row sample."""
    import oracle
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = int(os.environ.get("OMP_NUM_THREADS", "1"))
    t0 = time.perf_counter()
    code, _ = oracle.vq(x_sample, codebook)
    n_img = x_sample.shape[0] // per_image
    oracle.build_histogram([code[i * per_image:(i + 1) * per_image] for i in range(n_img)],
                           codebook.shape[0], norm)
    dt = time.perf_counter() - t0
    one_core = None
    try:   # the same restatement on one BLAS thread, on a tenth of the sample
        from threadpoolctl import threadpool_limits
        ns1 = max(per_image, x_sample.shape[0] // 10 // per_image * per_image)
        with threadpool_limits(limits=1):
            t1 = time.perf_counter()
            oracle.vq(x_sample[:ns1], codebook)
            one_core = ns1 / (time.perf_counter() - t1)
    except Exception:
        pass
    scipy_value = scipy_agree = None
    try:   # the library call the reference itself makes (model/features.py:1166), where installed
        from scipy.cluster.vq import vq as scipy_vq
        ns2 = max(per_image, x_sample.shape[0] // 10 // per_image * per_image)
        xs, cs = x_sample[:ns2].astype(np.float64), codebook.astype(np.float64)
        t2 = time.perf_counter()
        sc_code, _ = scipy_vq(xs, cs)
        scipy_value = ns2 / (time.perf_counter() - t2)
        scipy_agree = float(np.mean(sc_code == code[:ns2]))
    except ImportError:
        pass
    return {"value": x_sample.shape[0] / dt, "unit": "assignments/s", "cores": int(cores),
            "kind": "port", "one_core_value": one_core, "scipy_vq_value": scipy_value,
            "scipy_vq_agreement_with_port": scipy_agree,
            "sample": "%d rows of the workload (same codebook; float64 numpy GEMM + argmin + "
                      "np.histogram), %.1f s" % (x_sample.shape[0], dt)}, code


def cpu_baseline_logo(x, kp_rows, q, kp_q, per, n_logos, n_iter, gpu_inliers, gpu_off):
    """Oracle restatement of the reference's per-logo loop (2-NN + ratio test + its own ransac())
    on the first `n_logos` logo images; reports rows/s and how often accept/reject agrees."""
    import oracle
    from oracle import ransac_ref
    rows = n_logos * per
    xs = x[:rows].cpu().numpy()
    kps = kp_rows[:rows].cpu().numpy()
    qn, kq = q.cpu().numpy(), kp_q.cpu().numpy()
    cores = int(os.environ.get("OMP_NUM_THREADS", "1"))
    rng = np.random.RandomState(1)
    t0 = time.perf_counter()
    accepted = []
    for i in range(n_logos):
        mq, mi = oracle.robust_match_descriptors(qn, [xs[i * per:(i + 1) * per]], 10, 0.8)[0]
        ok = False
        if len(mi) > 8:
            data = np.append(kq[mq], kps[i * per:(i + 1) * per][mi], axis=1)
            fit, mask, _ = ransac_ref.ransac(data, ransac_ref.HomographyModel, 4, 100, 3.0, rng=rng)
            ok = fit is not None
        accepted.append(ok)
    dt = time.perf_counter() - t0
    gpu_ok = gpu_inliers[:n_logos].cpu().numpy() > 0
    return {"value": rows / dt, "unit": "assignments/s", "cores": cores, "kind": "port",
            "sample": "%d logo images (%d rows): float64 numpy 2-NN + ratio test + the reference's "
                      "ransac() restated, %.1f s" % (n_logos, rows, dt),
            "agreement_with_gpu": float(np.mean(gpu_ok == np.asarray(accepted)))}


def measured_traffic(workload, kernel="k_top2_partial"):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary of this
    workload (profiles/rNN_<workload>_traffic.json)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_traffic.json" % workload)))
    if not files:
        return None
    try:
        data = json.load(open(files[-1]))
        for name, rec in data.items():
            if kernel in name:
                return float(rec["hbm_bytes_per_launch"])
    except Exception:
        return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="bow")
    ap.add_argument("--codebook", type=int, default=0,
                    help="override the workload's codebook size (bow / video / kmeans)")
    ap.add_argument("--norm", default="", help="override the histogram norm (bow / video)")
    ap.add_argument("--cpu-sample", type=int, default=400_000,
                    help="rows of the workload the CPU oracle is timed on (0 = skip)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group even with one rank (smoke test)")
    ap.add_argument("--dist-backend", default="nccl",
                    help="'gloo' lets several ranks share one GPU to rehearse the multi-rank code "
                         "path on a one-GPU box (RCCL refuses two ranks on one device)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if args.dist_backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    from image_classification_amd import distributed as D
    from image_classification_amd import ops
    from image_classification_amd.backend import HipBackend
    from image_classification_amd.synthetic import image_offsets, sift_like_device

    w = dict(WORKLOADS[args.workload])
    if args.codebook > 0:
        w["k"] = args.codebook
        w["name"] += " [codebook overridden: %d]" % args.codebook
    if args.norm:
        w["norm"] = args.norm
        w["name"] += " [norm overridden: %s]" % args.norm
    backend = HipBackend(dev)
    n_rows, k_words = w["rows"], w["k"]
    flops_per_launch = None
    units_per_step = n_rows  # what `value` counts per rank and step

    if args.workload in ("bow", "video"):
        cb = sift_like_device(k_words, DIM, n_modes=2048, seed=4, device=dev)
        x = sift_like_device(n_rows, DIM, n_modes=2048, seed=3 + 1000 * rank, device=dev)
        offsets = torch.from_numpy(image_offsets(w["images"], n_rows // w["images"])).to(dev)
        cn = ops.row_sqnorm(cb)
        flops_per_launch = 2.0 * n_rows * k_words * DIM

        def step():
            idx, _ = ops.assign_top1(x, cb, cn, want_dist=False)
            return idx, ops.hist(idx, offsets, k_words, w["norm"])
    elif args.workload == "kmeans":
        x = sift_like_device(n_rows, DIM, n_modes=4096, seed=7 + 1000 * rank, device=dev)
        state = {"c": sift_like_device(k_words, DIM, n_modes=4096, seed=8, device=dev)}
        allreduce = D.make_allreduce() if use_dist else None
        flops_per_launch = 2.0 * n_rows * k_words * DIM

        def step():
            c, labels, _ = D.lloyd(x, state["c"], 1, backend, allreduce=allreduce,
                                   labels0=state.get("labels"))
            state["c"] = c
            state["labels"] = labels
            return labels, c
    elif args.workload == "logo":
        n_img, per = w["images"], n_rows // w["images"]
        g = torch.Generator(device=dev)
        g.manual_seed(11 + rank)
        q = sift_like_device(k_words, DIM, n_modes=4096, seed=5, device=dev)
        kp_q = torch.rand((k_words, 2), dtype=torch.float64, device=dev, generator=g) * 500
        x = sift_like_device(n_rows, DIM, n_modes=4096, seed=6 + 1000 * rank, device=dev)
        kp_rows = torch.rand((n_rows, 2), dtype=torch.float64, device=dev, generator=g) * 500
        # 5 % of all rows are true matches: every second logo really contains the query, 10 of
        # its 100 rows are noisy copies of query descriptors whose keypoints follow one homography
        planted = per // 10
        hom = torch.tensor([[1.1, 0.05, 20.0], [-0.04, 0.95, 10.0], [5e-5, 2e-5, 1.0]],
                           dtype=torch.float64, device=dev)
        rows = (torch.arange(0, n_img, 2, device=dev)[:, None] * per
                + torch.arange(planted, device=dev)[None, :]).reshape(-1)
        src = torch.randint(0, k_words, (rows.numel(),), device=dev, generator=g)
        x[rows] = q[src] + 2.0 * torch.randn((rows.numel(), DIM), device=dev, generator=g)
        t = torch.cat([kp_q[src], torch.ones((rows.numel(), 1), dtype=torch.float64, device=dev)], 1) @ hom.T
        kp_rows[rows] = t[:, :2] / t[:, 2:] + 0.3 * torch.randn((rows.numel(), 2), dtype=torch.float64,
                                                                device=dev, generator=g)
        row_off = torch.from_numpy(image_offsets(n_img, per)).to(dev)
        cn = ops.row_sqnorm(q)
        flops_per_launch = 2.0 * n_rows * k_words * DIM
        n_iter = 46   # int(log(1 - 0.95) / log(1 - 0.5 ** 4)), the reference's defaults

        def step():
            i1, _d1, _i2, _d2, mask = ops.assign_top2(x, q, 0.8, cn)
            out_off, mq, mi, pts = ops.compact_matches(mask, i1, row_off, kp_q, kp_rows)
            res = ops.ransac_homography(pts, out_off, n_iter, 3.0, None, 1, 10, True)
            return res[1], out_off
    elif args.workload == "histknn":
        g = torch.Generator(device=dev)
        g.manual_seed(21 + rank)
        index = torch.empty((n_rows, k_words), dtype=torch.float32, device=dev)
        for r0 in range(0, n_rows, 10_000):   # sparse count histograms, about 1000 words per image
            cnt = torch.poisson(torch.full((min(10_000, n_rows - r0), k_words), 0.1, device=dev),
                                generator=g)
            index[r0:r0 + cnt.shape[0]] = cnt / cnt.sum(1, keepdim=True).clamp_(min=1)
        query = index[n_rows // 2:n_rows // 2 + 1].clone()
        bytes_per_launch = float(n_rows) * k_words * 4

        def step():
            return ops.knn_rows(query, index, 2, "manhattan")
    else:  # knn: the index is split over ranks, the queries are replicated
        bounds = D.shard_bounds(k_words, world)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        index = sift_like_device(hi - lo, DIM, n_modes=4096, seed=6 + 1000 * rank, device=dev)
        q = sift_like_device(n_rows, DIM, n_modes=4096, seed=5, device=dev)
        flops_per_launch = 2.0 * n_rows * (hi - lo) * DIM
        units_per_step = n_rows * (hi - lo)  # query x index pairs examined by this rank

        # the index is static: it is wrapped once, with its squared norms, as the logo DB would be
        index_cb = backend.make_codebook(index)

        def step():
            return D.knn2_index_sharded(q, index_cb, lo, 0.8, backend=backend)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ops.profile_begin(local_rank)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out_a, out_b = step()[:2]
    fence()
    dt = time.perf_counter() - t0
    kern_ms, launches = ops.profile_end(local_rank)

    t = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())

    phases = None
    if args.workload == "kmeans" and rank == 0:
        # one more iteration with a device synchronisation between its phases
        def timed(fn):
            torch.cuda.synchronize()
            t_a = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            return r, (time.perf_counter() - t_a) * 1e3
        cbk = backend.make_codebook(state["c"])
        (lab, _), t_assign = timed(lambda: backend.assign_top1(x, cbk, want_dist=False))
        (sums, counts), t_part = timed(lambda: backend.centroid_partials(x, lab, k_words))
        t_red = 0.0
        if allreduce is not None and args.dist_backend == "nccl" and world == 1:
            _, t_red = timed(lambda: allreduce(sums, counts))
        _, t_fin = timed(lambda: backend.centroid_finalize(sums, counts, state["c"]))
        phases = {"assign_ms": t_assign, "centroid_sums_ms": t_part, "allreduce_ms": t_red,
                  "finalize_ms": t_fin}

    if rank == 0:
        props = torch.cuda.get_device_properties(dev)
        value = world * units_per_step * args.steps / dt_max
        avg_kernel_s = kern_ms / max(1, launches) / 1e3
        is_pairs = args.workload == "knn"
        is_hist = args.workload == "histknn"
        if is_hist:
            achieved = bytes_per_launch / avg_kernel_s / 1e9
            roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS,
                        "traffic": measured_traffic(args.workload, "k_pairdist_few"),
                        "kernel": "k_pairdist_few", "avg_kernel_ms": avg_kernel_s * 1e3,
                        "bytes_per_launch": bytes_per_launch}
        else:
            achieved = flops_per_launch / avg_kernel_s / 1e12
        metric, unit = ("descriptor->codeword assignments/sec", "assignments/s")
        if is_pairs:
            metric, unit = ("query-index descriptor pairs examined/sec", "pairs/s")
        if is_hist:
            metric, unit = ("index histograms compared/sec", "histograms/s")
        out = {
            "metric": metric,
            "value": value,
            "unit": unit,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if not is_pairs else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": w["name"], "rows_per_gpu": n_rows,
                       "codebook": k_words if not is_pairs else int(flops_per_launch / (2.0 * n_rows * DIM)),
                       "dim": k_words if is_hist else DIM,
                       "parallelism": {"bow": "rows sharded, no collective",
                                       "video": "frames sharded, no collective",
                                       "logo": "logo images sharded, no collective",
                                       "kmeans": "rows sharded, RCCL all-reduce of sums+counts per iteration",
                                       "histknn": "replicas (index rows would shard, top-k merge)",
                                       "knn": "index sharded, RCCL all-gather + top-2 merge"}[args.workload]},
            "roofline": roofline if is_hist else {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
                         "traffic": measured_traffic(args.workload),
                         "kernel": "k_top2_partial", "avg_kernel_ms": avg_kernel_s * 1e3,
                         "flops_per_launch": flops_per_launch,
                         "peak_basis": "256 CU x 256 FLOP/clk x 2.4 GHz; this device reports "
                                       "%d CUs%s" % (props.multi_processor_count,
                                                     " at %.0f MHz" % (props.clock_rate / 1e3)
                                                     if hasattr(props, "clock_rate") else "")},
        }
        if world == 1 and args.cpu_sample > 0 and args.workload in ("bow", "video"):
            per_image = n_rows // w["images"]
            ns = min(args.cpu_sample, n_rows)
            ns -= ns % per_image
            base, ref_code = cpu_baseline(x[:ns].cpu().numpy(), cb.cpu().numpy(), per_image, w["norm"])
            base["agreement_with_gpu"] = float(np.mean(ref_code == out_a[:ns].cpu().numpy()))
            out["cpu_baseline"] = base
        if phases is not None:
            out["config"]["phase_ms"] = phases
        if args.workload in ("bow", "video") and args.cpu_sample > 0:
            # every row of the workload against the direct float32 search on the device (no norm
            # expansion; outside the timed region)
            e1, ed1, _e2, ed2, _m = ops.assign_top2_exact(x, cb)
            gap = (ed2 - ed1) > 1e-5 * ed1
            out["parity"] = {"checked_rows": int(n_rows), "checker": "direct float32 search on the device",
                             "agreement": float((out_a == e1).float().mean()),
                             "mismatches_where_gap_exceeds_1e-5": int(((out_a != e1) & gap).sum())}
        if world == 1 and args.cpu_sample > 0 and args.workload == "logo":
            out["cpu_baseline"] = cpu_baseline_logo(
                x, kp_rows, q, kp_q, per, max(1, min(n_img, args.cpu_sample // per // 2)), n_iter,
                out_a, out_b)
        print(json.dumps(out), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

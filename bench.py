#!/usr/bin/env python3
"""Benchmark of the descriptor -> codeword assignment path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of synthetic descriptors resident in HBM:
BASELINE config 2, "BoW quantisation + per-image L2-normalised histograms: 1M descriptors x
10k-word codebook, 128-d" = ic_assign_top1 over [1M,128] x [10k,128] followed by ic_hist (1000
images x 1000 descriptors, L2).  With N > 1 (launched by torch.distributed.run, one rank per GPU)
every rank processes its own 1M descriptors -- the path shards by rows and needs no collective --
so scaling is weak and `value` is the sum over ranks.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the fp32-MFMA distance /
top-2 kernel): algorithmic FLOPs per launch = 2 * N * K * D, divided by its average duration
measured with HIP events recorded on the launch stream inside the timed region.  `cpu_baseline`
(rank 0, N == 1 only) times the float64 numpy oracle on a bounded row sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_ROWS = 1_000_000
K_WORDS = 10_000
DIM = 128
N_IMAGES = 1000
FP32_MFMA_PEAK_TFLOPS = 157.3  # 256 CU x 256 FLOP/clk x 2.4 GHz


def cpu_baseline(x_sample, codebook, per_image):
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
                           codebook.shape[0], "l2")
    dt = time.perf_counter() - t0
    return {"value": x_sample.shape[0] / dt, "unit": "assignments/s", "cores": int(cores),
            "kind": "port",
            "sample": "%d of the %d rows (same codebook, float64 numpy GEMM + argmin + np.histogram), "
                      "%.1f s" % (x_sample.shape[0], N_ROWS, dt)}, code


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=100_000,
                    help="rows of the workload the CPU oracle is timed on (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from image_classification_amd import ops
    from image_classification_amd.synthetic import sift_like_device, image_offsets

    # identical codebook on every rank, rank-specific descriptors
    cb = sift_like_device(K_WORDS, DIM, n_modes=2048, seed=4, device=dev)
    x = sift_like_device(N_ROWS, DIM, n_modes=2048, seed=3 + 1000 * rank, device=dev)
    offsets = torch.from_numpy(image_offsets(N_IMAGES, N_ROWS // N_IMAGES)).to(dev)
    cn = ops.row_sqnorm(cb)

    def step():
        idx, _ = ops.assign_top1(x, cb, cn, want_dist=False)
        return idx, ops.hist(idx, offsets, K_WORDS, "l2")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ops.profile_begin(local_rank)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        idx, h = step()
    fence()
    dt = time.perf_counter() - t0
    kern_ms, launches = ops.profile_end(local_rank)

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())

    if rank == 0:
        value = world * N_ROWS * args.steps / dt_max
        avg_kernel_s = kern_ms / max(1, launches) / 1e3
        flops = 2.0 * N_ROWS * K_WORDS * DIM
        achieved = flops / avg_kernel_s / 1e12
        out = {
            "metric": "descriptor->codeword assignments/sec",
            "value": value,
            "unit": "assignments/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE config 2: BoW quantisation + per-image L2 histograms, "
                                   "1M descriptors x 10k-word codebook x 128-d per GPU "
                                   "(1000 images x 1000 descriptors)",
                       "rows_per_gpu": N_ROWS, "codebook": K_WORDS, "dim": DIM,
                       "images_per_gpu": N_IMAGES, "parallelism": "rows sharded, no collective"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
                         "traffic": None,
                         "kernel": "k_top2_partial", "avg_kernel_ms": avg_kernel_s * 1e3,
                         "flops_per_launch": flops},
        }
        if world == 1 and args.cpu_sample > 0:
            ns = min(args.cpu_sample, N_ROWS)
            per_image = N_ROWS // N_IMAGES
            ns -= ns % per_image
            base, ref_code = cpu_baseline(x[:ns].cpu().numpy(), cb.cpu().numpy(), per_image)
            base["agreement_with_gpu"] = float(np.mean(ref_code == idx[:ns].cpu().numpy()))
            out["cpu_baseline"] = base
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- gate bootstraps/s (NAND, 128-bit params) on N MI355X.

One "step" = one pass of the hot path over BASELINE.json configs[1]: a batch of 1024
independent NAND gates at the 128-bit parameter set (n=700, N=1024) per GPU -- fused gate
prep + blind rotate (700 CMUX steps) + sample extract + key switch -- with keys and inputs
already resident in HBM.  Multi-GPU is weak scaling: every rank owns a full cloud-key replica
and its own 1024-gate shard, no collective on the data path.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (k_blind_rotate):
algorithmic bytes per launch / mean launch duration measured with HIP events on the launch
stream.  `cpu_baseline` (N=1, rank 0) times the C oracle (a port of the Go reference, which
cannot run here: no Go toolchain) on the host cores over a bounded sample of the same gates.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_VECTOR_PEAK_TFLOPS = 78.6   # 256 CUs x 4 SIMDs x 16 fp64 FMA lanes x 2 flop x 2.4 GHz
BATCH = 1024
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # written by tools/prof_pmc.sh (rocprofv3 --pmc passes)


def measured_traffic(kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950
    note in MI355X_MICROARCH.md, WRITE_SIZE as reported, both in KiB units), or None."""
    try:
        rec = json.load(open(PMC_FILE))[kernel]
        return rec["hbm_bytes_per_launch"]
    except Exception:
        return None


def fp64_flops_per_bootstrap(p):
    """n x (2L+2 transforms x 5*(N/2)*log2(N/2) + 4L*(N/2)*8)."""
    M = p.N // 2
    logm = M.bit_length() - 1
    return p.n * ((2 * p.L + 2) * 5 * M * logm + 4 * p.L * M * 8)


def algorithmic_bytes_blind_rotate(p):
    """Per bootstrap, streaming model (SURVEY.md 8d / DESIGN.md): the whole bootstrapping key
    once + the two gate operands + the test vector in, the TRLWE accumulator out."""
    bsk = p.n * 2 * p.L * 2 * p.N * 8
    return bsk + 2 * (p.n + 1) * 4 + 2 * p.N * 4 + 2 * p.N * 4


def algorithmic_bytes_keyswitch(p):
    """Per bootstrap: expected N*t*(1-1/base) key rows + TRLWE in + LWE out."""
    rows = p.N * p.t * (1.0 - 1.0 / p.base)
    return rows * (p.n + 1) * 4 + 2 * p.N * 4 + (p.n + 1) * 4


def effective_cores():
    """Host cores this process may actually use: affinity mask, capped by the cgroup CPU quota
    (the GPU box exposes 256 logical CPUs but a 16-CPU quota; oversubscribing it collapses)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(p128, a, b, bsk_torus, ksk, budget_s=12.0):
    """Oracle (port of the reference) on the host cores: all threads, one bootstrap per thread
    (mirrors trgsw.BatchBlindRotate's goroutine per input, trgsw.go:234-252)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle
    o = Oracle()
    p = o.params("128")
    bsk_f = np.empty(bsk_torus.shape, np.float64)
    flat_t = bsk_torus.reshape(-1, p.N)
    flat_f = bsk_f.reshape(-1, p.N)
    for i in range(flat_t.shape[0]):
        flat_f[i] = o.to_fourier(flat_t[i])
    cores = effective_cores()
    # single-thread latency on 2 gates (comparable to BenchmarkBootstrapNAND, gates_test.go:505-518)
    t0 = time.perf_counter()
    o.gate_batch(p, bsk_f, ksk, "NAND", a[:2], b[:2], nthreads=1)
    one = (time.perf_counter() - t0) / 2
    # all cores: size the sample for ~budget_s of wall time, at least one gate per thread
    per_thread = max(1, int(budget_s / max(one * 1.5, 1e-3)))
    S = min(a.shape[0], cores * per_thread)
    t0 = time.perf_counter()
    _, used = o.gate_batch(p, bsk_f, ksk, "NAND", a[:S], b[:S], nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": S / dt, "unit": "gates/s", "cores": used, "kind": "port",
            "sample": f"{S} of the same {a.shape[0]} NAND gates, one bootstrap per thread; "
                      f"gcc -O2 scalar radix-2 FFT port of the Go reference",
            "single_thread_ms_per_gate": one * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if os.environ.get("TFHE_BENCH_SHARE_GPU"):                    # dry run of the N>1 path on a 1-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    backend = os.environ.get("TFHE_BENCH_BACKEND", "nccl")      # "gloo" only for single-GPU dry runs of the N>1 path
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    def barrier():
        if dist:
            dist.barrier()

    # one rank compiles (no-op when the shipped .so is current), the others wait
    if rank == 0:
        graft.build()
    barrier()
    pkg = graft.load_package()
    p = pkg.params.Security128Bit

    # ---- synthetic random-key data (no secret key needed: throughput is value-independent)
    rs = np.random.RandomState(0x7F4E0002 + rank)

    def rnd(shape):
        return rs.randint(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)

    krs = np.random.RandomState(0x7F4E0002)         # same cloud key on every rank (replicated)
    bsk_torus = krs.randint(0, 2**32, size=(p.n, 2 * p.L, 2, p.N), dtype=np.uint64).astype(np.uint32)
    ksk = krs.randint(0, 2**32, size=(p.ksk_rows, p.n + 1), dtype=np.uint64).astype(np.uint32)
    ksk.reshape(p.N * p.t, p.base, p.n + 1)[:, 0, :] = 0      # k = 0 rows are zero (cloudkey.go:104-106)
    ck = pkg.CloudKey(p, bsk_torus=bsk_torus, ksk=ksk, device=local_rank)
    ctx = ck.ctx
    a_h, b_h = rnd((BATCH, p.n + 1)), rnd((BATCH, p.n + 1))
    a = torch.from_numpy(a_h.view(np.int32)).to(dev)
    b = torch.from_numpy(b_h.view(np.int32)).to(dev)
    out = torch.empty_like(a)
    stream = torch.cuda.current_stream()

    def step():
        ctx.gate_batch_dev("NAND", a, b, None, out, stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ctx.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ctx.timing_enable(False)
    br_n, br_ms = ctx.timing_read(0)
    ks_n, ks_ms = ctx.timing_read(1)
    if dist:
        tt = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = world * BATCH * args.steps / elapsed
        br_avg_ms = br_ms / max(br_n, 1)
        ks_avg_ms = ks_ms / max(ks_n, 1)
        alg = algorithmic_bytes_blind_rotate(p) * BATCH
        achieved = alg / (br_avg_ms * 1e-3) / 1e9
        ks_alg = algorithmic_bytes_keyswitch(p) * BATCH
        line = {
            "metric": "gate bootstraps/sec (NAND, 128-bit params)",
            "value": value, "unit": "gates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: batch of 1024 independent NAND bootstraps per GPU, "
                                   "128-bit params (n=700, N=1024, L=3, Bgbit=6, t=9), keys+inputs resident in HBM",
                       "batch_per_gpu": BATCH, "parallelism": f"batch-shard x{world} (replicated cloud key)"},
            "roofline": {"kernel": "k_blind_rotate", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic("k_blind_rotate"),
                         "algorithmic_bytes_per_launch": alg, "avg_launch_ms": br_avg_ms, "launches": br_n,
                         "note": "all 1024 bootstraps stream the same key, four per workgroup in step: each XCD's L2 fetches "
                                 "it once (traffic = 8 x 68.8 MB), so algorithmic GB/s exceeds the HBM peak; the kernel "
                                 "is fp64-VALU/LDS bound",
                         "fp64_tflops": fp64_flops_per_bootstrap(p) * BATCH / (br_avg_ms * 1e-3) / 1e12,
                         "fp64_vector_peak_tflops": FP64_VECTOR_PEAK_TFLOPS},
            "kernels": {"k_blind_rotate_ms": br_avg_ms, "k_extract_keyswitch_ms": ks_avg_ms,
                        "keyswitch_algorithmic_GBps": ks_alg / (ks_avg_ms * 1e-3) / 1e9 if ks_n else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(p, a_h, b_h, bsk_torus, ksk)
        print(json.dumps(line), flush=True)
    ck.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

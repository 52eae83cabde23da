"""CPU tier: the multi-GPU batch scatter / compute / gather path with world_size 2 over gloo.
The per-rank compute is the oracle here (no GPU in this tier); on the GPU box the same
ShardedGates object is driven by go_tfhe_amd.distributed.gpu_compute (RCCL)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import __graft_entry__ as graft
    from oracle_lib import Oracle
    from conftest import KeySet
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = graft.load_package()
    from go_tfhe_amd.distributed import ShardedGates
    o = Oracle()
    ks = KeySet(o, "128", 0x7F4E0003, n_override=8)          # same seed on every rank = replicated key
    n1 = ks.p.n + 1

    def compute(ops, a, b, c):
        an, bn = a.numpy().view(np.uint32), b.numpy().view(np.uint32)
        cn = c.numpy().view(np.uint32) if c is not None else None
        op = ops if isinstance(ops, str) else ops.numpy()
        out, _ = o.gate_batch(ks.p, ks.bsk, ks.ksk, op, np.ascontiguousarray(an), np.ascontiguousarray(bn),
                              None if cn is None else np.ascontiguousarray(cn), nthreads=2)
        return torch.from_numpy(out.view(np.int32))

    eng = ShardedGates(compute, n1)
    B = 7                                                      # ragged: shards of 3 and 4
    rs = np.random.RandomState(21)
    names = ["AND", "OR", "XOR", "MUX", "NAND", "MUX", "XNOR"]
    if rank == 0:
        a, b, c = (rs.randint(0, 2**32, size=(B, n1), dtype=np.uint64).astype(np.uint32) for _ in range(3))
        ops = np.array([pkg.OPS[x] for x in names], np.uint8)
        ta, tb, tc = (torch.from_numpy(x.view(np.int32)) for x in (a, b, c))
        got = eng.gate_batch(torch.from_numpy(ops), ta, tb, tc)
        want, _ = o.gate_batch(ks.p, ks.bsk, ks.ksk, ops, a, b, c)
        ok = np.array_equal(got.numpy().view(np.uint32), want)
        got2 = eng.gate_batch("NAND", ta, tb)                   # uniform op, no third operand
        want2, _ = o.gate_batch(ks.p, ks.bsk, ks.ksk, "NAND", a, b)
        ok = ok and np.array_equal(got2.numpy().view(np.uint32), want2)
        got3 = eng.gate_batch("NAND", ta[:1], tb[:1])           # batch smaller than the world
        ok = ok and np.array_equal(got3.numpy().view(np.uint32), want2[:1])
        got4 = eng.gate_batch(torch.from_numpy(ops[:6]), ta[:6], tb[:6], tc[:6])     # even shards: strided packing path
        ok = ok and np.array_equal(got4.numpy().view(np.uint32), want[:6])
        pk = eng.pack("NAND", ta, tb)                                                # pre-packed by the caller
        got5 = eng.gate_batch(None, None, None, packed=pk)
        ok = ok and np.array_equal(got5.numpy().view(np.uint32), want2)
        ok = ok and set(eng.last_timing) == {"scatter_s", "compute_s", "gather_s"}
    else:
        for _ in range(5):
            eng.gate_batch(None, None, None, None)
    # ---- a circuit sharded BY CIRCUIT: 2-bit reference-form adder (10 gates, Constant(false) carry-in), 4 circuits
    from go_tfhe_amd.circuits import ripple_carry_adder, adder_constant_wire
    from go_tfhe_amd.distributed import ShardedCircuits
    bits, C = 2, 4
    levels, n_wires, sums, cout = ripple_carry_adder(bits, fold_carry_in=False)

    def run_local(wires):
        w = wires.numpy().view(np.uint32)
        for lvl in levels:
            for (op, x, y, z, out) in lvl:
                res, _ = o.gate_batch(ks.p, ks.bsk, ks.ksk, op, np.ascontiguousarray(w[x]), np.ascontiguousarray(w[y]), nthreads=1)
                w[out] = res
        return wires

    circ = ShardedCircuits(run_local, n_wires, n1)
    in_wires = list(range(2 * bits)) + [adder_constant_wire(bits)]
    out_wires = sums + [cout]
    if rank == 0:
        av, bv = np.array([0, 1, 2, 3]), np.array([3, 3, 1, 2])
        inp = np.zeros((len(in_wires), C, n1), np.uint32)
        for i in range(bits):
            inp[i] = ks.enc((av >> i) & 1)
            inp[bits + i] = ks.enc((bv >> i) & 1)
        inp[2 * bits] = pkg.gates.Constant(False, ks.p)
        res = circ.run(in_wires, out_wires, torch.from_numpy(inp.view(np.int32))).numpy().view(np.uint32)
        got = sum(ks.dec(res[i]).astype(np.int64) << i for i in range(bits)) + (ks.dec(res[bits]).astype(np.int64) << bits)
        ok = ok and np.array_equal(got, av + bv)
        full = torch.zeros((n_wires, C, n1), dtype=torch.int32)                       # unsharded run: identical wires
        full[torch.tensor(in_wires)] = torch.from_numpy(inp.view(np.int32))
        run_local(full)
        ok = ok and np.array_equal(full[torch.tensor(out_wires)].numpy().view(np.uint32), res)
        # UNEVEN circuit counts: 3 circuits on 2 ranks (shares 1 and 2), then 1 circuit (one rank has nothing to do)
        for Cu in (3, 1):
            r_u = circ.run(in_wires, out_wires, torch.from_numpy(np.ascontiguousarray(inp[:, :Cu]).view(np.int32))).numpy().view(np.uint32)
            ok = ok and r_u.shape == (len(out_wires), Cu, n1) and np.array_equal(r_u, res[:, :Cu])
        q.put(ok)
    else:
        for _ in range(3):
            circ.run(in_wires, out_wires)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_gates_world2_gloo(built):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _bench_worker(rank, world, port, q):
    """bench.py's multi-rank plumbing without a GPU: the backend handshake must put EVERY rank on gloo when RCCL cannot
    form a group (no device here), and the timed sharded loop (barriers, max-over-ranks reduce, per-rank records with
    telemetry that degrades to 'unavailable') must run over it with a ShardedGates engine."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import bench
    graft = bench.graft
    graft.load_package()
    from go_tfhe_amd.distributed import ShardedGates
    dist, group, backend = bench.init_distributed(rank, world, torch.device("cpu"), "nccl")
    ok = backend == "gloo" and group is None and dist.get_world_size() == world
    env = bench.DistEnv(dist, group, backend, rank, world, torch.device("cpu"), 0)
    ok = ok and env.cdev == "cpu"
    n1 = 9

    def compute(ops, a, b, c):                                  # stand-in for the local path: out = a + b (mod 2^32)
        return a + b

    eng = ShardedGates(compute, n1)
    B = 5
    if rank == 0:
        a = torch.arange(B * n1, dtype=torch.int32).view(B, n1)
        b = torch.ones((B, n1), dtype=torch.int32)
        run = lambda: eng.gate_batch("NAND", a, b)
    else:
        run = lambda: eng.gate_batch(None, None, None, None)
    res, rec = bench._sharded_timed(env, eng, run, steps=3, warmup=1)
    if rank == 0:
        ok = ok and torch.equal(res, a + b)
        ok = ok and rec["n_gpus"] == world and rec["steps"] == 3 and len(rec["per_rank"]) == world
        ok = ok and [r["rank"] for r in rec["per_rank"]] == list(range(world))
        ok = ok and all(r["telemetry"]["available"] is False for r in rec["per_rank"])       # no SMI device in this tier
        ok = ok and rec["ms_per_step"] >= max(rec["scatter_ms_per_step"], rec["compute_ms_per_step"], rec["gather_ms_per_step"]) > 0
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_backend_handshake_and_sharded_loop_world2_gloo(built):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True

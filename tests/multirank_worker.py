#!/usr/bin/env python3
"""One rank of the multi-device functional test (launched by torch.distributed.run; see tests/test_gpu_multidevice.py and
tools/cpx_functional.sh).  NOT collected by pytest.

Every rank owns ONE device (LOCAL_RANK).  Over RCCL (backend "nccl") it exercises exactly what go-tfhe_amd/distributed.py
offers for SURVEY.md 8(e) -- the fan-out of trgsw.BatchBlindRotate (trgsw/trgsw.go:234-252) across GPUs:

  1. broadcast_cloud_key: rank 0 alone uploads the seeded cloud key; every other rank receives the two header-checked device blobs;
  2. ShardedGates.gate_batch: a ragged mixed batch (all ten gates + MUX, per-item op codes), then a uniform-op batch, held by rank 0;
  3. ShardedCircuits.run: a 4-bit ripple-carry adder x C circuits (C not a multiple of the world size), sharded by circuit;
  4. every rank additionally computes ITS shard locally and compares it, word for word, with the CPU oracle (each rank rebuilds the
     seeded host key for that purpose: a checker, never on the data path).

Rank 0 prints one JSON line: world_size, collective_backend, ranks_verified, devices, and the root-side comparisons with the oracle."""
import datetime
import json
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    from conftest import KeySet, gpu_params, rand_u32

    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    ndev = torch.cuda.device_count()
    if local >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {ndev} device(s) visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=300))
    pkg = graft.load_package()          # the library was built by the launcher; nothing compiles here
    from oracle_lib import Oracle
    from go_tfhe_amd.distributed import ShardedGates, ShardedCircuits, broadcast_cloud_key, gpu_compute, shard_bounds
    from go_tfhe_amd.circuits import ripple_carry_adder, adder_constant_wire, CircuitExecutor

    o = Oracle()
    k = KeySet(o, "128", 0x7F4E0003, n_override=24)            # the same seeded host key on every rank: the checker's
    n1 = k.p.n + 1
    gp = gpu_params(pkg, k.p)
    if rank == 0:
        ck = pkg.CloudKey(gp, bsk_fourier=k.bsk, ksk=k.ksk, device=local)
    else:
        ck = pkg.CloudKey(gp, device=local)                    # no key: it arrives over RCCL
    broadcast_cloud_key(ck.ctx, src=0)
    checks = {}

    # ---- 2. ragged mixed batch held by rank 0
    rs = np.random.RandomState(61)
    B = 8 * world + 5
    a, b, c = (rand_u32(rs, (B, n1)) for _ in range(3))
    ops = rs.randint(0, 11, size=B).astype(np.uint8)
    eng = ShardedGates(gpu_compute(ck.ctx), n1, device=dev)
    if rank == 0:
        ta, tb, tc = (torch.from_numpy(x.view(np.int32)).to(dev) for x in (a, b, c))
        got = eng.gate_batch(torch.from_numpy(ops).to(dev), ta, tb, tc)
        got2 = eng.gate_batch("XNOR", ta, tb)
        torch.cuda.synchronize()
        want, _ = o.gate_batch(k.p, k.bsk, k.ksk, ops, a, b, c)
        want2, _ = o.gate_batch(k.p, k.bsk, k.ksk, "XNOR", a, b)
        checks["sharded_mixed_batch_equals_oracle"] = bool(np.array_equal(got.cpu().numpy().view(np.uint32), want))
        checks["sharded_uniform_batch_equals_oracle"] = bool(np.array_equal(got2.cpu().numpy().view(np.uint32), want2))
    else:
        eng.gate_batch(None, None, None, None)
        eng.gate_batch(None, None, None)

    # ---- 3. adder circuits sharded by circuit
    bits, C = 4, 3 * world + 1
    levels, n_wires, sums, cout = ripple_carry_adder(bits, fold_carry_in=False)
    ex = CircuitExecutor(ck.ctx, levels, n_wires)
    sc = ShardedCircuits(ex.run, n_wires, n1, device=dev)
    in_wires = list(range(2 * bits)) + [adder_constant_wire(bits)]
    out_wires = sums + [cout]
    if rank == 0:
        rs = np.random.RandomState(62)
        av, bv = rs.randint(0, 1 << bits, C), rs.randint(0, 1 << bits, C)
        rng = o.rng(0x7F4E0062)
        inp = np.zeros((len(in_wires), C, n1), np.uint32)
        for i in range(bits):
            inp[i] = o.encrypt_bools(k.p, rng, (av >> i) & 1, k.s0)
            inp[bits + i] = o.encrypt_bools(k.p, rng, (bv >> i) & 1, k.s0)
        inp[2 * bits] = pkg.gates.Constant(False, k.p)
        res = sc.run(in_wires, out_wires, torch.from_numpy(inp.view(np.int32)).to(dev))
        torch.cuda.synchronize()
        r = res.cpu().numpy().view(np.uint32)
        dec = sum(k.dec(np.ascontiguousarray(r[i])).astype(np.int64) << i for i in range(bits + 1))
        checks["adder_sums_decrypt"] = bool(np.array_equal(dec, av + bv))
        c0 = C - 1                                              # a circuit of the last rank's share, gate by gate on the oracle
        ow = {w: inp[j, c0] for j, w in enumerate(in_wires)}
        for lvl in levels:
            for (op, x, y, z, w_out) in lvl:
                ow[w_out] = o.gate(k.p, k.bsk, k.ksk, op, np.ascontiguousarray(ow[x]), np.ascontiguousarray(ow[y]))
        checks["adder_last_circuit_wires_equal_oracle"] = all(bool(np.array_equal(r[j, c0], ow[w])) for j, w in enumerate(out_wires))
    else:
        sc.run(in_wires, out_wires)

    # ---- 4. every rank: its own contiguous shard of the mixed batch, locally, against the oracle; and the key it received
    lo, hi = shard_bounds(B, world, rank)
    mine = ck.ctx.gate_batch(ops[lo:hi], a[lo:hi], b[lo:hi], c[lo:hi])
    want_m, _ = o.gate_batch(k.p, k.bsk, k.ksk, ops[lo:hi], a[lo:hi], b[lo:hi], c[lo:hi])
    ok_local = bool(np.array_equal(mine, want_m))
    ref = pkg.CloudKey(gp, bsk_fourier=k.bsk, ksk=k.ksk, device=local)     # what an upload on this rank would have installed
    key_ok = all(torch.equal(ref.ctx.key_export_dev(w).cpu(), ck.ctx.key_export_dev(w).cpu()) for w in (0, 1))
    ref.close()
    props = torch.cuda.get_device_properties(local)
    info = {"rank": rank, "device": local, "name": props.name, "cus": props.multi_processor_count,
            "shard_equals_oracle": ok_local, "received_key_blobs_equal_an_upload": key_ok}
    flag = torch.tensor([1 if (ok_local and key_ok) else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag)                                       # RCCL sum: how many ranks verified
    per_rank = [None] * world
    dist.all_gather_object(per_rank, info)
    ck.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        ok = all(checks.values()) and int(flag.item()) == world
        print(json.dumps({"world_size": world, "collective_backend": "nccl", "devices_visible": ndev, "ranks_verified": int(flag.item()),
                          "verified": ok, "root_checks": checks, "per_rank": per_rank}), flush=True)
        sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

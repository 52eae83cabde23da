#!/usr/bin/env python3
"""Throughput of BASELINE configs 3, 4, 5 on one GPU (synthetic random keys; values do not affect
timing).  Run on the GPU box: python tools/measure_configs.py > gpurun_out/configs.json"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.build(); pkg = g.load_package()
from go_tfhe_amd.circuits import ripple_carry_adder, CircuitExecutor, count_gates

rs = np.random.RandomState(7)
rnd = lambda shape: rs.randint(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)
out = {}

def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps

# ---- 128-bit key
p = pkg.params.Security128Bit
ck = pkg.CloudKey(p, bsk_torus=rnd((p.n, 2 * p.L, 2, p.N)), ksk=rnd((p.ksk_rows, p.n + 1)))
n1 = p.n + 1
# config 3: 8-bit ripple-carry adder x 256 circuits.  "reference" = the 40-gate circuit exactly as README.md:78-106
# writes it (8 FullAdders from Constant(false): the contractual workload); "folded" = 37 gates (carry-in folded away).
from go_tfhe_amd.circuits import balance_levels, schedule_min_cost
for tag, fold in (("reference40", False), ("folded37", True)):
    levels, nw, sums, cout = ripple_carry_adder(8, fold_carry_in=fold)
    wires = torch.from_numpy(rnd((nw, 256, n1)).view(np.int32)).cuda()
    G = count_gates(levels) * 256
    for sched, lv in (("asap", levels), ("balanced", balance_levels(levels, 1024 // 256)), ("mincost", schedule_min_cost(levels, 256))):
        ex = CircuitExecutor(ck.ctx, lv, nw)
        dt = timed(lambda: ex.run(wires), 5)
        graph = ex.capture(wires)
        dtg = timed(lambda: graph.replay(), 5)
        out[f"config3_adder8_x256_{tag}_{sched}"] = {
            "gates": G, "levels": len(lv), "widths": [len(l) for l in lv], "seconds": dt, "gates_per_s": G / dt,
            "adds_per_s": 256 / dt, "graph_replay_seconds": dtg, "graph_gates_per_s": G / dtg}
        del graph
        ex.release()
# config 5 (one-GPU slice): mixed AND/OR/XOR/MUX stream, 65536 gates
B = 65536
ops = torch.from_numpy(np.array([1, 2, 3, 10], np.uint8)[rs.randint(0, 4, B)]).cuda()
a, b, c = (torch.from_numpy(rnd((B, n1)).view(np.int32)).cuda() for _ in range(3))
o = torch.empty_like(a)
dt = timed(lambda: ck.ctx.gate_batch_dev(ops, a, b, c, o), 3)
nb = int((ops == 10).sum().item()) * 3 + int((ops != 10).sum().item())
out["config5_mixed_stream_64k"] = {"gates": B, "bootstraps": nb, "seconds": dt, "gates_per_s": B / dt, "bootstraps_per_s": nb / dt}
# large uniform batches (how throughput moves with batch size)
for Bb in (1024, 4096, 16384):
    a2, b2 = a[:Bb].contiguous(), b[:Bb].contiguous(); o2 = torch.empty_like(a2)
    dt = timed(lambda: ck.ctx.gate_batch_dev("NAND", a2, b2, None, o2), 5)
    out[f"nand_batch_{Bb}"] = {"seconds": dt, "gates_per_s": Bb / dt}
ck.close()
# ---- config 4: Uint5 PBS batch 512
p = pkg.params.SecurityUint5
ck = pkg.CloudKey(p, bsk_torus=rnd((p.n, 2 * p.L, 2, p.N)), ksk=rnd((p.ksk_rows, p.n + 1)))
cts = torch.from_numpy(rnd((512, p.n + 1)).view(np.int32)).cuda()
lut = torch.from_numpy(rnd((2, p.N)).view(np.int32)).cuda()
o = torch.empty_like(cts)
dt = timed(lambda: ck.ctx.bootstrap_batch_dev(cts, lut, o), 5)
out["config4_pbs_uint5_x512"] = {"pbs": 512, "seconds": dt, "pbs_per_s": 512 / dt,
                                 "blind_rotate_ms": ck.ctx.last_kernel_ms(0), "keyswitch_ms": ck.ctx.last_kernel_ms(1)}
ck.close()
# ---- Uint2 (N=512, one wave per bootstrap): PBS at one co-resident launch (2048) and a quarter of it
p = pkg.params.SecurityUint2
ck = pkg.CloudKey(p, bsk_torus=rnd((p.n, 2 * p.L, 2, p.N)), ksk=rnd((p.ksk_rows, p.n + 1)))
lut = torch.from_numpy(rnd((2, p.N)).view(np.int32)).cuda()
for Bb in (512, 2048, 8192):
    cts = torch.from_numpy(rnd((Bb, p.n + 1)).view(np.int32)).cuda()
    o = torch.empty_like(cts)
    dt = timed(lambda: ck.ctx.bootstrap_batch_dev(cts, lut, o), 5)
    out[f"pbs_uint2_x{Bb}"] = {"pbs": Bb, "seconds": dt, "pbs_per_s": Bb / dt,
                                "blind_rotate_ms": ck.ctx.last_kernel_ms(0), "keyswitch_ms": ck.ctx.last_kernel_ms(1)}
print(json.dumps(out, indent=1))

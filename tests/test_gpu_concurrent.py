"""GPU tier: concurrent submitters on ONE context are combined, not serialised (include/tfhe_hip.h, tfhe_gate_batch).

The reference's scalar gates.* share one evaluator that is not goroutine-safe (gates.go:19-23,136-142); its concurrency is
goroutine fan-out over pooled evaluators (trgsw.go:227-252).  Here any number of threads may call gates.* on one CloudKey:
requests that arrive while a launch is in flight are issued together as ONE gate batch with per-item op codes (flat
combining), and each caller gets exactly the rows a call on its own would have returned."""
import threading
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _adder_inputs(k, pkg, C, bits, seed):
    rs = np.random.RandomState(seed)
    av, bv = rs.randint(0, 256, C), rs.randint(0, 256, C)
    n1 = k.p.n + 1
    a_bits = np.stack([k.enc((av >> i) & 1) for i in range(bits)])        # [bits][C][n1]
    b_bits = np.stack([k.enc((bv >> i) & 1) for i in range(bits)])
    return av, bv, a_bits, b_bits, n1


def test_256_threads_issuing_scalar_gates_are_combined_and_bit_identical(oracle, keys128, ck128, pkg):
    # 256 threads, each adding two encrypted bytes with the reference's 40 scalar gate calls IN SEQUENCE (README.md:78-106),
    # all on one context.  Serialised that is 256 x 40 launches of ~2.5 ms = 25 s; combined it is ~40 launches of <= 256 gates.
    from go_tfhe_amd.circuits import ripple_carry_adder, adder_constant_wire, CircuitExecutor
    g = pkg.gates
    k, ctx = keys128, ck128.ctx
    C, bits = 256, 8
    av, bv, a_bits, b_bits, n1 = _adder_inputs(k, pkg, C, bits, 77)
    const_false = g.Constant(False, k.p)
    sums = np.zeros((C, bits + 1, n1), np.uint32)

    def add_bytes(c):                                    # one caller: the README's FullAdder chain, gate by gate
        carry = const_false
        for i in range(bits):
            x = g.XOR(a_bits[i, c], b_bits[i, c], ck128)
            gen = g.AND(a_bits[i, c], b_bits[i, c], ck128)
            sums[c, i] = g.XOR(x, carry, ck128)
            t = g.AND(x, carry, ck128)
            carry = g.OR(gen, t, ck128)
        sums[c, bits] = carry

    add_bytes(0)                                         # warm-up (and the lone-caller path: no combined launch)
    assert ctx.get_option("combine_launches") == 0
    threads = [threading.Thread(target=add_bytes, args=(c,)) for c in range(C)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt_threads = time.perf_counter() - t0
    launches, carried = ctx.get_option("combine_launches"), ctx.get_option("combine_requests")

    # the same 256 additions through the levelised batch executor: every output word must be identical
    levels, n_wires, sum_w, cout = ripple_carry_adder(bits, fold_carry_in=False)
    wires = np.zeros((n_wires, C, n1), np.uint32)
    wires[:bits] = a_bits
    wires[bits:2 * bits] = b_bits
    wires[adder_constant_wire(bits)] = const_false
    wt = torch.from_numpy(wires.view(np.int32)).cuda()
    ex = CircuitExecutor(ctx, levels, n_wires)
    ex.run(wt.clone()); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ex.run(wt); torch.cuda.synchronize()
    dt_exec = time.perf_counter() - t0
    res = wt.cpu().numpy().view(np.uint32)
    for i, w in enumerate(sum_w):
        assert np.array_equal(sums[:, i], res[w]), f"sum bit {i} differs from the batch path"
    assert np.array_equal(sums[:, bits], res[cout])
    got = sum(k.dec(np.ascontiguousarray(sums[:, i])).astype(np.int64) << i for i in range(bits + 1))
    assert np.array_equal(got, av + bv)

    # combined: far fewer launches than requests, and a wall time in the region of 40 sequential launches, not 10,240.
    # (40 dependent rounds x ~2.6 ms is ~105 ms of kernels against the executor's 17 levels; the rest is 256 Python threads
    # taking turns at the interpreter lock -- the C++ measurement is tools/combine_bench.cpp.)
    assert carried >= 0.9 * 40 * C and launches <= carried / 8, (launches, carried)
    print(f"\n256 threads x 40 scalar gates: {dt_threads * 1e3:.0f} ms in {launches} combined launches carrying {carried} calls "
          f"(+ {40 * C - carried} lone); CircuitExecutor x256: {dt_exec * 1e3:.0f} ms")
    assert dt_threads < 40 * C * 2.4e-3 / 10, "concurrent scalar gates are being serialised"


def test_combined_mixed_requests_equal_serial_results(oracle, keys_small, ck_small, pkg):
    # callers with different shapes at once: uniform ops, per-item ops with MUX (third operand), batches of several rows --
    # each result equals the same call issued alone with combining switched off
    k, ctx = keys_small, ck_small.ctx
    rs = np.random.RandomState(5)
    n1 = k.p.n + 1
    rnd = lambda B: rs.randint(0, 2**32, size=(B, n1), dtype=np.uint64).astype(np.uint32)
    reqs = []
    for i in range(48):
        B = [1, 1, 3, 17][i % 4]
        a, b, c = rnd(B), rnd(B), rnd(B)
        if i % 3 == 0:
            reqs.append(("XOR", a, b, None))
        elif i % 3 == 1:
            reqs.append((np.array([10, 1, 2, 0, 4, 9, 10] * 3, np.uint8)[:B], a, b, c))
        else:
            reqs.append(("MUX", a, b, c))
    ctx.set_option("combine_max", 0)
    want = [ctx.gate_batch(op, a, b, c) for op, a, b, c in reqs]
    ctx.set_option("combine_max", -1)
    before = ctx.get_option("combine_requests")
    got = [None] * len(reqs)
    gate = threading.Barrier(len(reqs))

    def run(i):
        op, a, b, c = reqs[i]
        gate.wait()
        got[i] = ctx.gate_batch(op, a, b, c)

    ts = [threading.Thread(target=run, args=(i,)) for i in range(len(reqs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for i in range(len(reqs)):
        assert np.array_equal(got[i], want[i]), f"request {i} differs from the serial path"
    assert ctx.get_option("combine_requests") > before          # at least some of them travelled together


def test_combined_launch_error_reaches_every_caller_and_the_context_survives(keys_small, ck_small, pkg):
    # a bad op code is refused per call, before queueing: the other callers are not affected
    k, ctx = keys_small, ck_small.ctx
    rs = np.random.RandomState(6)
    n1 = k.p.n + 1
    a = rs.randint(0, 2**32, size=(1, n1), dtype=np.uint64).astype(np.uint32)
    errs, oks = [], []

    def bad():
        try:
            ctx.gate_batch(np.array([77], np.uint8), a, a)
        except pkg.TfheError as e:
            errs.append(str(e))

    def good():
        oks.append(ctx.gate_batch("AND", a, a))

    ts = [threading.Thread(target=bad if i % 2 else good) for i in range(16)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert len(errs) == 8 and all("bad op code" in e for e in errs)
    assert len(oks) == 8 and all(np.array_equal(o, oks[0]) for o in oks)


def test_key_export_survives_the_first_combined_launch_of_a_context(keys_small, pkg):
    # regression: the page-locked header of tfhe_key_export_dev and the combiner's page-locked staging are separate allocations
    # with separate lifetimes (the first combined launch of a context allocates its staging; it must not touch the header)
    from conftest import gpu_params
    k = keys_small
    ck = pkg.CloudKey(gpu_params(pkg, k.p), bsk_fourier=k.bsk, ksk=k.ksk)
    try:
        ctx = ck.ctx
        before = [ctx.key_export_dev(w).cpu() for w in (0, 1)]
        rs = np.random.RandomState(8)
        a = rs.randint(0, 2**32, size=(1, k.p.n + 1), dtype=np.uint64).astype(np.uint32)
        gate = threading.Barrier(8)
        outs = [None] * 8

        def run(i):
            gate.wait()
            outs[i] = ctx.gate_batch("NAND", a, a)

        ts = [threading.Thread(target=run, args=(i,)) for i in range(8)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert ctx.get_option("combine_launches") >= 1
        assert all(np.array_equal(o, outs[0]) for o in outs)
        for w in (0, 1):
            again = ctx.key_export_dev(w).cpu()
            assert torch.equal(again, before[w])
            ctx.key_import_dev(w, again.cuda())
        assert np.array_equal(ctx.gate_batch("NAND", a, a), outs[0])
    finally:
        ck.close()


def test_combiner_stress_random_sizes_ops_and_oversize_calls(keys_small, ck_small, pkg):
    # 24 threads x 12 calls each with random batch sizes (1 ... 40, and some beyond TFHE_OPT_COMBINE_MAX, which take the context for
    # themselves), random per-item op codes incl. MUX, no start barrier: leaders, waiters, promotions, gathering waits and oversize
    # calls interleave freely.  Every result must equal the same call issued alone with combining switched off.
    k, ctx = keys_small, ck_small.ctx
    n1 = k.p.n + 1
    T, CALLS = 24, 12
    rs = np.random.RandomState(99)
    pool = rs.randint(0, 2**32, size=(3, 400, n1), dtype=np.uint64).astype(np.uint32)
    plan = []
    for t in range(T):
        calls = []
        for _ in range(CALLS):
            B = int(rs.choice([1, 1, 1, 2, 7, 40, 130]))
            off = int(rs.randint(0, 400 - B))
            ops = rs.randint(0, 11, size=B).astype(np.uint8)
            calls.append((B, off, ops))
        plan.append(calls)
    ctx.set_option("combine_max", 0)
    want = [[ctx.gate_batch(ops, pool[0, off:off + B], pool[1, off:off + B], pool[2, off:off + B]) for B, off, ops in calls]
            for calls in plan]
    ctx.set_option("combine_max", 64)                   # the 130-gate calls are oversize: they bypass the combiner
    before = ctx.get_option("combine_requests")
    got = [[None] * CALLS for _ in range(T)]
    errors = []

    def run(t):
        try:
            for i, (B, off, ops) in enumerate(plan[t]):
                got[t][i] = ctx.gate_batch(ops, pool[0, off:off + B], pool[1, off:off + B], pool[2, off:off + B])
        except Exception as e:                          # noqa: BLE001 -- reported below, with the thread
            errors.append((t, repr(e)))

    ts = [threading.Thread(target=run, args=(t,)) for t in range(T)]
    for th in ts:
        th.start()
    for th in ts:
        th.join(timeout=120)
    try:
        assert not any(th.is_alive() for th in ts), "a caller never returned (lost wake-up or leadership not handed over)"
        assert not errors, errors
        for t in range(T):
            for i in range(CALLS):
                assert np.array_equal(got[t][i], want[t][i]), (t, i, plan[t][i][0])
        assert ctx.get_option("combine_requests") > before
    finally:
        ctx.set_option("combine_max", -1)


@pytest.mark.parametrize("which", ["80", "uint5"])
def test_concurrent_programmable_bootstraps_are_combined_and_bit_identical(oracle, pkg, request, which):
    # evaluator.BootstrapLUT from many threads on one context (programmable_bootstrap.go:93-115): callers with their OWN table,
    # with per-item tables, and with none (the gate test vector) travel in one launch with one table per item; every result is,
    # word for word, what the same call returns alone with combining switched off.  "80": exact regime (N = 1024, L = 3);
    # "uint5" (n reduced): tolerance regime -- still bit-identical, because a combined launch stays within the kernel shape
    # of a lone call (at most one bootstrap per CU).
    from conftest import KeySet, gpu_params
    if which == "80":
        k = request.getfixturevalue("keys80")
        modulus = 2
    else:
        k = KeySet(oracle, "uint5", 0x7F4E0061, n_override=24, torus=False)
        modulus = 32
    ck = pkg.CloudKey(gpu_params(pkg, k.p), bsk_fourier=k.bsk, ksk=k.ksk)
    try:
        ctx = ck.ctx
        rs = np.random.RandomState(71)
        n1, N = k.p.n + 1, k.p.N
        luts = [oracle.lut_generate(k.p, [(a * x + b) % modulus for x in range(modulus)]) for a, b in ((1, 0), (3, 1), (modulus - 1, 2))]
        T = 40
        reqs = []
        for t in range(T):
            B = [1, 1, 2, 5][t % 4]
            msgs = rs.randint(0, modulus, B)
            cts = np.stack([oracle.encrypt_message(k.p, k.rng, int(m), modulus, k.s0) for m in msgs])
            if t % 5 == 4:
                tv = None                                           # the gate test vector
            elif t % 5 == 3:
                tv = np.stack([luts[(t + i) % 3] for i in range(B)])   # one table per item
            else:
                tv = luts[t % 3]
            reqs.append((cts, tv, msgs))
        ctx.set_option("combine_max", 0)
        want = [ctx.bootstrap_batch(cts, tv) for cts, tv, _ in reqs]
        ctx.set_option("combine_max", -1)
        before_l, before_r = ctx.get_option("combine_launches"), ctx.get_option("combine_requests")
        got = [None] * T
        gate = threading.Barrier(T)

        def run(t):
            gate.wait()
            got[t] = ctx.bootstrap_batch(reqs[t][0], reqs[t][1])

        ts = [threading.Thread(target=run, args=(t,)) for t in range(T)]
        for th in ts:
            th.start()
        for th in ts:
            th.join(timeout=120)
        assert not any(th.is_alive() for th in ts)
        for t in range(T):
            assert np.array_equal(got[t], want[t]), f"request {t} differs from the serial path"
        assert ctx.get_option("combine_launches") > before_l and ctx.get_option("combine_requests") - before_r >= 2
        # and the values are right: a shared-table request decrypts to f(m)
        cts, tv, msgs = reqs[1]
        fa, fb = [(1, 0), (3, 1), (modulus - 1, 2)][1 % 3]
        dec = [oracle.decrypt_message(k.p, modulus, k.s0, np.ascontiguousarray(o)) for o in got[1]]
        assert dec == [int((fa * m + fb) % modulus) for m in msgs]
    finally:
        ck.close()


@pytest.mark.parametrize("op", ["XOR", "MUX"])
def test_combined_gates_at_a_tolerance_regime_shape_equal_lone_calls(oracle, pkg, op):
    # at the shapes whose transforms are not exact (here Uint1: N = 1024, L = 2, Bgbit = 10) kernels of different
    # launch shapes round differently, so a combined GATE launch -- like a combined table bootstrap -- must stay within the
    # kernel shape a lone small call runs (at most one bootstrap per CU, within the four-/eight-wave limits): every caller's
    # words equal the same call issued alone with combining switched off, even when far more gates than CUs are in flight.
    # "MUX" (ADVICE r05): a MUX row puts TWO bootstraps into pass 1 of the launch (its own AND and the ANDNY of the list), so the
    # combiner must budget bootstraps, not rows -- 96 callers x 5 MUX rows = 960 bootstraps, against 256 per lone-shape launch.
    from conftest import KeySet, gpu_params
    k = KeySet(oracle, "uint1", 0x7F4E0071, n_override=24, torus=False)
    ck = pkg.CloudKey(gpu_params(pkg, k.p), bsk_fourier=k.bsk, ksk=k.ksk)
    try:
        ctx = ck.ctx
        n1 = k.p.n + 1
        rs = np.random.RandomState(17)
        T = 96                                          # x 5 gates = 480 > the 256 CUs: more than one lone-shape launch
        rnd = lambda: rs.randint(0, 2**32, size=(5, n1), dtype=np.uint64).astype(np.uint32)
        reqs = [(rnd(), rnd(), rnd() if op == "MUX" else None) for _ in range(T)]
        ctx.set_option("combine_max", 0)
        want = [ctx.gate_batch(op, a, b, c) for a, b, c in reqs]
        ctx.set_option("combine_max", -1)
        before = ctx.get_option("combine_requests")
        got = [None] * T
        gate = threading.Barrier(T)

        def run(i):
            gate.wait()
            got[i] = ctx.gate_batch(op, *reqs[i])

        ts = [threading.Thread(target=run, args=(i,)) for i in range(T)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for i in range(T):
            assert np.array_equal(got[i], want[i]), f"request {i}: a combined launch left the lone call's kernel shape"
        assert ctx.get_option("combine_requests") > before
    finally:
        ck.close()


def test_mid_size_host_batches_from_several_threads_overlap_and_equal_lone_calls(oracle, keys_small, ck_small, pkg):
    # gates.Batch* on host memory from several goroutines: batches of more than one row per CU are not combined, their uploads / kernels /
    # downloads overlap across callers in two buffer slots (gate_batch_overlapped).  Every result must equal the same call issued alone --
    # and the oracle -- whatever ran beside it: uniform ops, per-item ops with MUX, ragged sizes around the CU count and the launch size.
    import torch
    k, ctx = keys_small, ck_small.ctx
    n1 = k.p.n + 1
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rs = np.random.RandomState(123)
    rnd = lambda B: rs.randint(0, 2**32, size=(B, n1), dtype=np.uint64).astype(np.uint32)
    T, CALLS = 5, 4
    plan = []
    for t in range(T):
        calls = []
        for i in range(CALLS):
            B = int(rs.choice([cus + 1, 300, 777, 1024, 1025, 2100]))
            ops = "XNOR" if (t + i) % 3 == 0 else rs.randint(0, 11, size=B).astype(np.uint8)
            calls.append((ops, rnd(B), rnd(B), rnd(B)))
        plan.append(calls)
    want = [[ctx.gate_batch(ops, a, b, c if not isinstance(ops, str) else None) for ops, a, b, c in calls] for calls in plan]
    got = [[None] * CALLS for _ in range(T)]
    errors = []

    def run(t):
        try:
            for i, (ops, a, b, c) in enumerate(plan[t]):
                got[t][i] = ctx.gate_batch(ops, a, b, c if not isinstance(ops, str) else None)
        except Exception as e:                          # noqa: BLE001
            errors.append((t, repr(e)))

    ts = [threading.Thread(target=run, args=(t,)) for t in range(T)]
    for th in ts:
        th.start()
    for th in ts:
        th.join(timeout=300)
    assert not any(th.is_alive() for th in ts) and not errors, errors
    for t in range(T):
        for i in range(CALLS):
            assert np.array_equal(got[t][i], want[t][i]), (t, i)
    ops, a, b, c = plan[0][1]
    ref, _ = oracle.gate_batch(k.p, k.bsk, k.ksk, ops, a[:40], b[:40], c[:40] if not isinstance(ops, str) else None)
    assert np.array_equal(got[0][1][:40], ref)


def test_mid_size_host_bootstrap_batches_from_several_threads_equal_lone_calls(oracle, keys_small, ck_small, pkg):
    # evaluator.BootstrapLUT over host batches from several goroutines (one shared table, one table per item, or the gate test vector): batches of
    # more than one sample per CU overlap across callers like the gate batches above (bootstrap_batch_overlapped); same words as lone calls and the oracle
    import torch
    k, ctx = keys_small, ck_small.ctx
    n1, N = k.p.n + 1, k.p.N
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rs = np.random.RandomState(321)
    w32 = lambda shape: rs.randint(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)
    T, CALLS = 4, 3
    plan = []
    for t in range(T):
        calls = []
        for i in range(CALLS):
            B = int(rs.choice([cus + 1, 400, 1025]))
            tv = [None, w32((2, N)), w32((B, 2, N))][(t + i) % 3]
            calls.append((w32((B, n1)), tv))
        plan.append(calls)
    want = [[ctx.bootstrap_batch(cts, tv) for cts, tv in calls] for calls in plan]
    got = [[None] * CALLS for _ in range(T)]
    errors = []

    def run(t):
        try:
            for i, (cts, tv) in enumerate(plan[t]):
                got[t][i] = ctx.bootstrap_batch(cts, tv)
        except Exception as e:                          # noqa: BLE001
            errors.append((t, repr(e)))

    ts = [threading.Thread(target=run, args=(t,)) for t in range(T)]
    for th in ts:
        th.start()
    for th in ts:
        th.join(timeout=300)
    assert not any(th.is_alive() for th in ts) and not errors, errors
    for t in range(T):
        for i in range(CALLS):
            assert np.array_equal(got[t][i], want[t][i]), (t, i)
    cts, tv = plan[1][0]
    tvi = k.tv if tv is None else tv
    for i in (0, 7, len(cts) - 1):
        table = tvi if tvi.ndim == 2 else tvi[i]
        assert np.array_equal(got[1][0][i], oracle.bootstrap(k.p, k.bsk, k.ksk, cts[i], np.ascontiguousarray(table)))

"""GPU tier: BASELINE configs 3 and 5 as parity-test cases.
  config 3: 8-bit ripple-carry adder (40 gates) x 256 circuits, 128-bit params -> sums decrypt
            to (a+b) mod 256 for all 256; levelised into 17 batched launches.
  config 5: mixed AND/OR/XOR/MUX stream; a 4096-gate prefix is compared bit-for-bit with the
            oracle on a sampled subset, the full decrypt is checked on every item."""
import numpy as np
import pytest
import torch

from conftest import gpu_params, rand_u32

pytestmark = pytest.mark.gpu


def test_adder_structure(pkg):
    from go_tfhe_amd.circuits import ripple_carry_adder, count_gates
    levels, n_wires, sums, cout = ripple_carry_adder(8)
    assert len(levels) == 15 + 0 or len(levels) >= 15       # 1 + 2*7 dependency levels
    assert count_gates(levels) == 2 * 8 + 3 * 7               # 37: the folded carry-in saves 3 of the 40
    ref_levels, nw, sums, cout = ripple_carry_adder(8, fold_carry_in=False)
    assert count_gates(ref_levels) == 40 and len(ref_levels) == 17      # README.md:78-106 as written
    assert [len(l) for l in ref_levels] == [16] + [2, 1] * 8


def test_adder_reference_form_40_gates_x256_128bit(oracle, keys128, ck128, pkg):
    # BASELINE config 3 exactly as the reference writes it: 8 FullAdders chained from carry := Constant(false)
    # (README.md:78-106; the constant is the trivial sample with body 1 - 1/8 = 0xE0000001, gates.go:61-69), 256
    # circuits at once.  Sums decrypt to (a+b) mod 256; one circuit is re-done gate by gate on the oracle and
    # every wire is the identical ciphertext; a graph replay of the captured level loop reproduces the run.
    from go_tfhe_amd.circuits import ripple_carry_adder, adder_constant_wire, CircuitExecutor
    k = keys128
    C, bits = 256, 8
    levels, n_wires, sums, cout = ripple_carry_adder(bits, fold_carry_in=False)
    rs = np.random.RandomState(43)
    av, bv = rs.randint(0, 256, C), rs.randint(0, 256, C)
    n1 = k.p.n + 1
    wires = np.zeros((n_wires, C, n1), np.uint32)
    for i in range(bits):
        wires[i] = k.enc((av >> i) & 1)
        wires[bits + i] = k.enc((bv >> i) & 1)
    cw = adder_constant_wire(bits)
    const_false = pkg.gates.Constant(False, k.p)
    assert const_false[k.p.n] == 0xE0000001 and not const_false[:k.p.n].any()
    wires[cw] = const_false
    wt = torch.from_numpy(wires.view(np.int32)).cuda()
    ex = CircuitExecutor(ck128.ctx, levels, n_wires)
    ex.run(wt)
    torch.cuda.synchronize()
    res = wt.cpu().numpy().view(np.uint32)
    got = np.zeros(C, np.int64)
    for i, w in enumerate(sums):
        got |= k.dec(res[w]).astype(np.int64) << i
    assert np.array_equal(got, (av + bv) % 256)
    assert np.array_equal(k.dec(res[cout]).astype(np.int64), (av + bv) >> 8)
    c0 = 101
    ow = {w: wires[w, c0] for w in list(range(2 * bits)) + [cw]}
    for lvl in levels:
        for (op, x, y, z, out) in lvl:
            ow[out] = oracle.gate(k.p, k.bsk, k.ksk, op, np.ascontiguousarray(ow[x]), np.ascontiguousarray(ow[y]))
    for w, v in ow.items():
        assert np.array_equal(res[w, c0], v), w
    # the same level loop captured into a HIP graph and replayed on fresh inputs
    wt2 = torch.from_numpy(wires.view(np.int32)).cuda()
    graph = ex.capture(wt2)
    wt2.copy_(torch.from_numpy(wires.view(np.int32)))       # capture ran the circuit once: restore the inputs
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(wt2, wt)
    del graph
    ex.release()                                             # the context is shared with later, larger batches


def test_adder_8bit_x256_128bit(oracle, keys128, ck128, pkg):
    from go_tfhe_amd.circuits import ripple_carry_adder, CircuitExecutor, count_gates
    k = keys128
    C, bits = 256, 8
    levels, n_wires, sums, cout = ripple_carry_adder(bits)
    rs = np.random.RandomState(41)
    av, bv = rs.randint(0, 256, C), rs.randint(0, 256, C)
    n1 = k.p.n + 1
    wires = np.zeros((n_wires, C, n1), np.uint32)
    for i in range(bits):
        wires[i] = k.enc((av >> i) & 1)
        wires[bits + i] = k.enc((bv >> i) & 1)
    wt = torch.from_numpy(wires.view(np.int32)).cuda()
    ex = CircuitExecutor(ck128.ctx, levels, n_wires)
    ex.run(wt)
    torch.cuda.synchronize()
    res = wt.cpu().numpy().view(np.uint32)
    got = np.zeros(C, np.int64)
    for i, w in enumerate(sums):
        got |= k.dec(res[w]).astype(np.int64) << i
    carry = k.dec(res[cout]).astype(np.int64)
    assert np.array_equal(got, (av + bv) % 256)
    assert np.array_equal(carry, (av + bv) >> 8)
    # the slack-balanced schedule (balance_levels: at most one full launch per level, same depth) computes
    # bit-identical ciphertexts on every wire
    from go_tfhe_amd.circuits import balance_levels
    bal = balance_levels(levels, 1024 // C)
    assert len(bal) == len(levels) and count_gates(bal) == count_gates(levels) and max(len(l) for l in bal) <= 4
    wt2 = torch.from_numpy(wires.view(np.int32)).cuda()
    CircuitExecutor(ck128.ctx, bal, n_wires).run(wt2)
    torch.cuda.synchronize()
    assert torch.equal(wt2, wt)
    # ... and so does the cost-aware schedule (schedule_min_cost: gates moved between levels by the measured launch costs)
    from go_tfhe_amd.circuits import schedule_min_cost
    mc = schedule_min_cost(levels, C)
    assert len(mc) == len(levels) and count_gates(mc) == count_gates(levels)
    wt3 = torch.from_numpy(wires.view(np.int32)).cuda()
    CircuitExecutor(ck128.ctx, mc, n_wires).run(wt3)
    torch.cuda.synchronize()
    assert torch.equal(wt3, wt)
    # one circuit re-done gate by gate on the oracle: identical ciphertexts on every wire it wrote
    c0 = 17
    ow = {w: wires[w, c0] for w in range(2 * bits)}
    for lvl in levels:
        for (op, x, y, z, out) in lvl:
            ow[out] = oracle.gate(k.p, k.bsk, k.ksk, op, np.ascontiguousarray(ow[x]), np.ascontiguousarray(ow[y]))
    for w in sums + [cout]:
        assert np.array_equal(res[w, c0], ow[w]), w


def test_mixed_stream_4096_128bit(oracle, keys128, ck128, pkg):
    k = keys128
    B = 4096
    rs = np.random.RandomState(42)
    pool_bits = rs.randint(0, 2, 64)
    pool = k.enc(pool_bits)                                    # operands drawn from a pool of encrypted bits
    ia, ib, ic = rs.randint(0, 64, B), rs.randint(0, 64, B), rs.randint(0, 64, B)
    names = np.array(["AND", "OR", "XOR", "MUX"])[rs.randint(0, 4, B)]
    ops = np.array([pkg.OPS[x] for x in names], np.uint8)
    a, b, c = pool[ia], pool[ib], pool[ic]
    out = pkg.gates.gate_stream(ops, a, b, ck128, c)
    A, Bb, Cc = pool_bits[ia].astype(bool), pool_bits[ib].astype(bool), pool_bits[ic].astype(bool)
    want = np.where(names == "AND", A & Bb, np.where(names == "OR", A | Bb, np.where(names == "XOR", A ^ Bb, np.where(A, Bb, Cc))))
    assert np.array_equal(k.dec(out), want)
    sample = [0, 1, 2, 3, 1000, 2047, 4095] + list(np.where(names == "MUX")[0][:3])
    ref, _ = oracle.gate_batch(k.p, k.bsk, k.ksk, ops[sample], np.ascontiguousarray(a[sample]),
                               np.ascontiguousarray(b[sample]), np.ascontiguousarray(c[sample]))
    assert np.array_equal(out[sample], ref)
    # the device-pointer entry point never sees the op codes on the host (the MUX items are compacted on the GPU):
    # same ciphertexts, and a second stream with a different MUX density through the same context
    ad, bd, cd_ = (torch.from_numpy(np.ascontiguousarray(v).view(np.int32)).cuda() for v in (a, b, c))
    od = torch.empty_like(ad)
    ck128.ctx.gate_batch_dev(torch.from_numpy(ops).cuda(), ad, bd, cd_, od)
    torch.cuda.synchronize()
    assert np.array_equal(od.cpu().numpy().view(np.uint32), out)
    ck128.ctx.sync()                                         # nothing was rejected
    for ops2 in (np.full(300, pkg.OPS["MUX"], np.uint8), np.array([pkg.OPS["XOR"]] * 299 + [pkg.OPS["MUX"]], np.uint8)):
        o2 = torch.empty_like(ad[:300])
        ck128.ctx.gate_batch_dev(torch.from_numpy(ops2).cuda(), ad[:300].contiguous(), bd[:300].contiguous(), cd_[:300].contiguous(), o2)
        torch.cuda.synchronize()
        want2 = ck128.ctx.gate_batch(ops2, a[:300], b[:300], c[:300])
        assert np.array_equal(o2.cpu().numpy().view(np.uint32), want2)
    om = torch.empty_like(ad[:64])
    ck128.ctx.gate_batch_dev("MUX", ad[:64].contiguous(), bd[:64].contiguous(), cd_[:64].contiguous(), om)
    torch.cuda.synchronize()
    assert np.array_equal(om.cpu().numpy().view(np.uint32), ck128.ctx.gate_batch("MUX", a[:64], b[:64], c[:64]))


def test_mux_stream_captured_into_a_graph(oracle, keys128, ck128, pkg):
    # A per-item-op batch with MUX items is enqueue-only too (the split happens on the device), so it records into a HIP
    # graph after tfhe_ctx_reserve has sized the intermediate buffers; replays on NEW operand values and a different MUX
    # pattern in the same device buffers give what the eager call gives.
    k = keys128
    B = 700
    rs = np.random.RandomState(61)
    n1 = k.p.n + 1
    bufs = [torch.empty((B, n1), dtype=torch.int32, device="cuda") for _ in range(4)]
    ops_d = torch.empty(B, dtype=torch.uint8, device="cuda")

    def fill():
        a, b, c = (rand_u32(rs, (B, n1)) for _ in range(3))
        ops = np.array([pkg.OPS[x] for x in ("AND", "MUX", "XNOR")], np.uint8)[rs.randint(0, 3, B)]
        for t, v in zip(bufs[:3], (a, b, c)):
            t.copy_(torch.from_numpy(v.view(np.int32)))
        ops_d.copy_(torch.from_numpy(ops))
        return ops, a, b, c

    ck128.ctx.reserve(B, with_mux=True)
    fill()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        ck128.ctx.gate_batch_dev(ops_d, bufs[0], bufs[1], bufs[2], bufs[3], side)       # warm-up on the capture stream
    side.synchronize()
    with torch.cuda.graph(graph, stream=side):
        ck128.ctx.gate_batch_dev(ops_d, bufs[0], bufs[1], bufs[2], bufs[3], torch.cuda.current_stream())
    for _ in range(2):
        ops, a, b, c = fill()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(bufs[3].cpu().numpy().view(np.uint32), ck128.ctx.gate_batch(ops, a, b, c))
    ck128.ctx.sync()
    del graph
    ck128.ctx.set_option("frozen", 0)                        # the graph is gone; later tests on this context run larger batches


def test_dev_path_reports_bad_op_codes_at_sync(ck128, keys128, pkg):
    # tfhe_gate_batch_dev never copies the op codes back, so it cannot refuse them up front: the kernels record
    # the problem and tfhe_ctx_sync reports it once.  The host-pointer call still validates before issuing work.
    k = keys128
    rs = np.random.RandomState(5)
    a = torch.from_numpy(rand_u32(rs, (8, k.p.n + 1)).view(np.int32)).cuda()
    out = torch.empty_like(a)
    bad = torch.tensor([0, 1, 2, 3, 4, 11, 6, 7], dtype=torch.uint8).cuda()
    ck128.ctx.gate_batch_dev(bad, a, a, a, out)
    torch.cuda.synchronize()
    with pytest.raises(pkg.TfheError):
        ck128.ctx.sync()
    ck128.ctx.sync()                                         # reported once, then cleared
    muxes = torch.full((8,), pkg.OPS["MUX"], dtype=torch.uint8).cuda()
    ck128.ctx.gate_batch_dev(muxes, a, a, None, out)         # MUX without a third operand
    torch.cuda.synchronize()
    with pytest.raises(pkg.TfheError):
        ck128.ctx.sync()
    with pytest.raises(pkg.TfheError):
        ck128.ctx.gate_batch(np.array([11], np.uint8), a[:1].cpu().numpy().view(np.uint32), a[:1].cpu().numpy().view(np.uint32))


def test_mixed_stream_per_gpu_share_131072(oracle, keys128, ck128, pkg):
    # BASELINE config 5 (SURVEY 8d numbering; BASELINE.json configs[4]: 1M mixed gates over 8 GPUs) at one GPU's full share, 2^20 / 8 gates, checked through
    # size-independent properties: every output decrypts to its truth-table value, a second run is
    # bit-identical (chunked launches and the three MUX passes are deterministic), a sample equals the oracle.
    k = keys128
    B = (1 << 20) // 8
    rs = np.random.RandomState(43)
    pool_bits = rs.randint(0, 2, 64)
    pool = k.enc(pool_bits)
    ia, ib, ic = rs.randint(0, 64, B), rs.randint(0, 64, B), rs.randint(0, 64, B)
    names = np.array(["AND", "OR", "XOR", "MUX"])[rs.randint(0, 4, B)]
    ops = np.array([pkg.OPS[x] for x in names], np.uint8)
    a, b, c = pool[ia], pool[ib], pool[ic]
    out = pkg.gates.gate_stream(ops, a, b, ck128, c)
    A, Bb, Cc = pool_bits[ia].astype(bool), pool_bits[ib].astype(bool), pool_bits[ic].astype(bool)
    want = np.where(names == "AND", A & Bb, np.where(names == "OR", A | Bb, np.where(names == "XOR", A ^ Bb, np.where(A, Bb, Cc))))
    # vectorised tlwe decrypt (tlwe/tlwe.go:64-73): phase = b - <a, s> mod 2^32, bit = int32(phase) >= 0;
    # the convention is pinned against the oracle's decrypt on a random sample below
    n = k.p.n
    phase = (out[:, n].astype(np.uint64) - (out[:, :n].astype(np.uint64) @ k.s0.astype(np.uint64))) & 0xFFFFFFFF
    bits = phase.astype(np.uint32).view(np.int32) >= 0
    probe = rs.randint(0, B, 64)
    assert np.array_equal(bits[probe], k.dec(out[probe]))
    assert np.array_equal(bits, want)
    again = pkg.gates.gate_stream(ops, a, b, ck128, c)
    assert np.array_equal(again, out)
    sample = [0, 1, B // 2, B - 1] + list(np.where(names == "MUX")[0][-2:])
    ref, _ = oracle.gate_batch(k.p, k.bsk, k.ksk, ops[sample], np.ascontiguousarray(a[sample]),
                               np.ascontiguousarray(b[sample]), np.ascontiguousarray(c[sample]))
    assert np.array_equal(out[sample], ref)


def test_growth_during_or_after_capture_is_refused_not_undefined(pkg, keys_small):
    """include/tfhe_hip.h, tfhe_ctx_reserve: the intermediate buffers are grow-only and growing re-allocates.  A "_dev"
    call that would have to grow one while its stream is being captured must fail with a message naming
    tfhe_ctx_reserve (not a raw HIP error), and once a call HAS been captured the context is frozen: a later, larger
    call is refused as well (a re-allocation would leave the graph replaying into freed memory) until the caller
    clears the flag.  A context reserved up front captures and replays."""
    import torch
    k = keys_small
    ck = pkg.CloudKey(gpu_params(pkg, k.p), bsk_fourier=k.bsk, ksk=k.ksk)
    ctx = ck.ctx
    a = torch.from_numpy(k.enc([0, 1] * 32).view(np.int32)).cuda()
    b = torch.from_numpy(k.enc([1, 1] * 32).view(np.int32)).cuda()
    out = torch.zeros_like(a)
    ctx.reserve(4)                                     # too small for the 64-gate batch below
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with pytest.raises(pkg.TfheError, match="tfhe_ctx_reserve"):
        with torch.cuda.graph(graph, stream=side):
            ctx.gate_batch_dev("NAND", a, b, None, out, side)
    torch.cuda.synchronize()
    assert ctx.get_option("frozen") == 0               # the refused call enqueued nothing: no graph holds any address
    ctx.gate_batch_dev("NAND", a, b, None, out)        # un-captured: grows
    ctx.sync()
    want = ~(np.array([0, 1] * 32, bool) & np.array([1, 1] * 32, bool))
    assert np.array_equal(k.dec(out.cpu().numpy().view(np.uint32)), want)
    # reserved up front: captures, replays, and stays usable for smaller batches afterwards
    ctx.reserve(64)
    out.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        ctx.gate_batch_dev("NAND", a, b, None, out, side)
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(k.dec(out.cpu().numpy().view(np.uint32)), want)
    ctx.gate_batch_dev("AND", a[:8].contiguous(), b[:8].contiguous(), None, out[:8])
    ctx.sync()
    assert ctx.get_option("frozen") == 1               # a call WAS captured: the graph holds the buffers' addresses
    big = torch.cat([a, a, a])
    with pytest.raises(pkg.TfheError, match="frozen"):
        ctx.gate_batch_dev("NAND", big, torch.cat([b, b, b]), None, torch.zeros_like(big))
    # a captured call with bad arguments is refused without side effects on a fresh context either
    del graph
    ctx.set_option("frozen", 0)                        # the graph is gone: growth is allowed again
    big_out = torch.zeros_like(big)
    ctx.gate_batch_dev("NAND", big, torch.cat([b, b, b]), None, big_out)
    ctx.sync()
    assert np.array_equal(k.dec(big_out.cpu().numpy().view(np.uint32)), np.concatenate([want] * 3))
    ck.close()


def test_executor_capture_reserves_for_its_widest_level_and_release_unfreezes(pkg, keys_small):
    """CircuitExecutor.capture sizes the context for the widest level of ITS schedule (not for a full 65,536-item slab with MUX
    buffers, ~1 GB, as it did until round 4); release() clears the frozen flag once the graphs are gone."""
    import torch
    from go_tfhe_amd.circuits import ripple_carry_adder, adder_constant_wire, CircuitExecutor
    k = keys_small
    ck = pkg.CloudKey(gpu_params(pkg, k.p), bsk_fourier=k.bsk, ksk=k.ksk)
    bits, C = 2, 4
    levels, n_wires, sums, cout = ripple_carry_adder(bits, fold_carry_in=False)
    n1 = k.p.n + 1
    wires = np.zeros((n_wires, C, n1), np.uint32)
    av, bv = np.array([0, 1, 2, 3]), np.array([3, 3, 1, 2])
    for i in range(bits):
        wires[i] = k.enc((av >> i) & 1)
        wires[bits + i] = k.enc((bv >> i) & 1)
    wires[adder_constant_wire(bits)] = pkg.gates.Constant(False, k.p)
    wt = torch.from_numpy(wires.view(np.int32)).cuda()
    ex = CircuitExecutor(ck.ctx, levels, n_wires)
    free0 = torch.cuda.mem_get_info()[0]
    graph = ex.capture(wt)
    assert free0 - torch.cuda.mem_get_info()[0] < 256 << 20, "capture() reserved far more than this circuit's widest level needs"
    graph.replay()
    torch.cuda.synchronize()
    res = wt.cpu().numpy().view(np.uint32)
    got = sum(k.dec(res[w]).astype(np.int64) << i for i, w in enumerate(sums)) + (k.dec(res[cout]).astype(np.int64) << bits)
    assert np.array_equal(got, av + bv)
    assert ck.ctx.get_option("frozen") == 1
    del graph
    ex.release()
    assert ck.ctx.get_option("frozen") == 0
    ck.close()


def test_sync_waits_for_every_stream_even_destroyed_ones(pkg, keys_small):
    """tfhe_ctx_sync waits on context-owned events recorded behind each "_dev" call, on whichever stream it was issued;
    the caller may have destroyed the stream since (the library keeps no stream handles)."""
    import torch
    k = keys_small
    ck = pkg.CloudKey(gpu_params(pkg, k.p), bsk_fourier=k.bsk, ksk=k.ksk)
    bits = [0, 1, 1, 0] * 8
    a = torch.from_numpy(k.enc(bits).view(np.int32)).cuda()
    b = torch.from_numpy(k.enc([1] * 32).view(np.int32)).cuda()
    ck.ctx.reserve(32)
    outs = []
    torch.cuda.synchronize()
    for i in range(3):
        st = torch.cuda.Stream()
        o = torch.zeros_like(a)
        ck.ctx.gate_batch_dev("AND", a, b, None, o, st)
        outs.append(o)
        del st                                          # torch returns the stream to its pool; we never touch it again
    ck.ctx.sync()                                       # no torch synchronisation: the context's own events
    for o in outs:
        assert np.array_equal(k.dec(o.cpu().numpy().view(np.uint32)), np.array(bits, bool))
    ck.close()


def test_two_executors_on_one_context_release_is_refcounted_and_threads_run_on_a_frozen_context(pkg, keys_small):
    """(i) release() of one executor must not un-freeze a context another executor's graph still replays on;
    (ii) a second capture that does not fit the frozen buffers fails with an error naming max_instances; (iii) threads issuing
    scalar gates on a frozen context get what lone calls return -- the combiner re-issues a batch one by one when only the
    COMBINATION needs buffers the frozen context may not grow."""
    import threading
    import torch
    from go_tfhe_amd.circuits import ripple_carry_adder, adder_constant_wire, CircuitExecutor
    k = keys_small
    ck = pkg.CloudKey(gpu_params(pkg, k.p), bsk_fourier=k.bsk, ksk=k.ksk)
    ctx = ck.ctx
    bits, C = 2, 2
    levels, n_wires, sums, cout = ripple_carry_adder(bits, fold_carry_in=False)
    n1 = k.p.n + 1
    wires = np.zeros((n_wires, C, n1), np.uint32)
    av, bv = np.array([1, 2]), np.array([3, 1])
    for i in range(bits):
        wires[i] = k.enc((av >> i) & 1)
        wires[bits + i] = k.enc((bv >> i) & 1)
    wires[adder_constant_wire(bits)] = pkg.gates.Constant(False, k.p)
    wt1 = torch.from_numpy(wires.view(np.int32)).cuda()
    wt2 = wt1.clone()
    ex1, ex2 = CircuitExecutor(ctx, levels, n_wires), CircuitExecutor(ctx, levels, n_wires)
    g1 = ex1.capture(wt1)
    g2 = ex2.capture(wt2)                                   # same size: fits the frozen buffers
    assert ctx.get_option("frozen") == 1 and ctx._graphs_alive == 2
    # (ii) a much wider capture on the frozen context: refused, and the message names the remedy
    wide = torch.from_numpy(np.zeros((n_wires, 4096, n1), np.int32)).cuda()
    with pytest.raises(RuntimeError, match="max_instances"):
        CircuitExecutor(ctx, levels, n_wires).capture(wide)
    del wide
    # (iii) 64 threads x scalar gates on the frozen context: the reserved scratch holds a handful of bootstraps, the combined
    # total would need more -> every caller must still get the lone-call result, none a 'frozen' error
    rs = np.random.RandomState(3)
    a = rs.randint(0, 2**32, size=(64, n1), dtype=np.uint64).astype(np.uint32)
    b = rs.randint(0, 2**32, size=(64, n1), dtype=np.uint64).astype(np.uint32)
    ctx.set_option("combine_max", 0)
    want = [ctx.gate_batch("NAND", a[i:i + 1], b[i:i + 1]) for i in range(64)]
    ctx.set_option("combine_max", -1)
    got, errs = [None] * 64, []
    gate = threading.Barrier(64)

    def run(i):
        gate.wait()
        try:
            for _ in range(3):
                got[i] = ctx.gate_batch("NAND", a[i:i + 1], b[i:i + 1])
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=run, args=(i,)) for i in range(64)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not errs, errs[:3]
    assert all(np.array_equal(got[i], want[i]) for i in range(64))
    # (i) refcount: releasing ex1 leaves the context frozen for ex2's graph, which still replays correctly
    del g1
    ex1.release()
    assert ctx.get_option("frozen") == 1 and ctx._graphs_alive == 1
    g2.replay()
    torch.cuda.synchronize()
    res = wt2.cpu().numpy().view(np.uint32)
    val = sum(k.dec(res[w]).astype(np.int64) << i for i, w in enumerate(sums)) + (k.dec(res[cout]).astype(np.int64) << bits)
    assert np.array_equal(val, av + bv)
    del g2
    ex2.release()
    assert ctx.get_option("frozen") == 0 and ctx._graphs_alive == 0
    ck.close()

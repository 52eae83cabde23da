#!/usr/bin/env python3
"""Differential fuzzer: libtfhe_hip.so against the C oracle, for as many minutes as asked.

    python tests/fuzz_gpu.py --minutes 10 --seed 1 [--log gpurun_out/fuzz.txt]

Every case draws a parameter set, a (reduced) LWE dimension, a batch size weighted towards the dispatch boundaries of
the library (1, the CU count, the slab and chunk sizes, +-1 around each), the entry point (gates with one op / one op per
item incl. MUX / programmable bootstraps through one table or one per item / blind rotate + key switch on their own; host
pointers or device pointers; T concurrent threads of dependent scalar-sized calls; now and then on a tfhe_ctx_clone_to replica, now
and then at the FULL LWE dimension; random levelised circuits through the CircuitExecutor, re-levelled and hipGraph-captured;
the same key brought in as a torus-form upload / imported blobs / a host-staged clone) and the kernel-dispatch options (TFHE_OPT_QUAD_MAX / OCT_MAX / KS_MFMA_MIN), then runs the same
words through the oracle:

  * N = 1024, L = 3, Bgbit = 6 sets (80 / 110 / 128-bit): the inputs are ARBITRARY words (no valid encryption needed: the
    transforms are exact there) with edge rows mixed in, and every output word must be IDENTICAL;
  * Uint sets (tolerance regime): valid encryptions of random messages; the key switch on its own must be identical (integer
    work), a programmable bootstrap must decrypt to table[message], one CMUX step from the engine's own state must sit
    within 2^9 words of the exact-integer increment, its output must be the (exact) key switch of its own accumulator, and
    the accumulator's phase must stay within a 16-sigma noise bound of the oracle's.

Test infrastructure (imports oracle/ through tests/oracle_lib.py); the product never sees it.  Exit status 1 on the first
mismatch, after writing the case's seed and a repro line."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))           # this file lives in tests/: it is test infrastructure (it drives the oracle)

import __graft_entry__ as graft  # noqa: E402

OPS2 = ["NAND", "AND", "OR", "XOR", "XNOR", "NOR", "ANDNY", "ANDYN", "ORNY", "ORYN"]
EXACT = ["80", "110", "128"]
FULL_N = {"80": 550, "110": 630, "128": 700}
UINT = {"uint1": 2, "uint2": 4, "uint3": 8, "uint4": 16, "uint5": 32}


class Key:
    def __init__(self, o, pkg, name, n, seed):
        import torch  # noqa: F401  (device memory for the _dev entry points)
        self.o, self.name, self.seed = o, name, seed
        self.p = o.params(name).small(n)
        rng = o.rng(seed)
        self.rng = rng
        self.s0, self.s1 = o.keygen_secret(self.p, rng)
        self.bsk_torus, self.bsk = o.keygen_bsk(self.p, rng, self.s0, self.s1, torus=True, fourier=True)
        self.ksk = o.keygen_ksk(self.p, rng, self.s0, self.s1)
        self.tv = o.gate_testvec(self.p)
        p = self.p
        self.ck = pkg.CloudKey(pkg.Params(n=p.n, N=p.N, Nbit=p.Nbit, L=p.L, Bgbit=p.Bgbit, basebit=p.basebit, t=p.t),
                               bsk_fourier=self.bsk, ksk=self.ksk)
        self.ctx = self.ck.ctx
        self.replica = None

    def clone(self):
        if self.replica is None:
            self.replica = self.ck.clone_to(0)              # tfhe_ctx_clone_to (device-to-device on a one-GPU box)
        return self.replica.ctx

    def close(self):
        if self.replica is not None:
            self.replica.close()
        self.ck.close()


def words(rs, shape):
    return rs.randint(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)


def edge_rows(rs, a):
    """Overwrite a few rows with the inputs mod-switch and rotation edge cases come from."""
    B = a.shape[0]
    for v in (0, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF, 0x001FFFFF, 0x00100000, 0xFFF00000):
        if B and rs.rand() < 0.5:
            a[rs.randint(B)] = v
    if B and rs.rand() < 0.5:
        a[rs.randint(B), -1] = rs.choice([0, 0xFFFFFFFF, 0xFFEFFFFF, 0xFFF00000, 0x000FFFFF, 0x00100000])
    return a


def pick_batch(rs, cus, heavy):
    marks = [1, 2, 3, cus // 2, cus, 2 * cus, 3 * cus, 4 * cus, 8 * cus, 1000, 1024, 4096]
    r = rs.rand()
    if r < 0.45:
        b = rs.choice(marks) + rs.randint(-2, 3)
    elif r < 0.8:
        b = rs.randint(1, 2 * cus + 2)
    else:
        b = rs.randint(1, 4200)
    b = int(max(1, b))
    return min(b, 1200) if heavy else b


def set_options(rs, ctx, log):
    opts = {"quad_max": rs.choice([-1, -1, 0, 1, 7, 64, 300, 5000]),
            "oct_max": rs.choice([-1, -1, 0, 1, 5, 64, 300]),
            "ks_mfma_min": rs.choice([-1, -1, 0, 1, 24, 100, 10**6])}
    for k, v in opts.items():
        ctx.set_option(k, int(v))
    log.append("opts=" + ",".join(f"{k}:{int(v)}" for k, v in opts.items()))


def case_exact(rs, o, K, log):
    import torch
    p, ctx = K.p, K.ctx
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    kind = rs.choice(["gate1", "gates", "gates_dev", "pbs", "pbs_items", "rotate_ks", "threads", "seams"], p=[.18, .22, .14, .1, .1, .1, .08, .08])
    B = pick_batch(rs, cus, heavy=p.n > 40)
    if p.n > 100:
        B = min(B, 300)                          # a full-size set: the oracle is the clock
    log.append(f"kind={kind} B={B}")
    if rs.rand() < 0.1:
        ctx = K.clone()
        log.append("on-a-clone")
    set_options(rs, ctx, log)
    n1 = p.n + 1
    if kind == "seams":
        return seams(rs, o, K, ctx, min(B, 200), log, exact=True)
    if kind == "threads":
        # concurrent callers of the host-pointer entry point (flat combining): every caller must get what a lone call gets
        import threading
        T = int(rs.randint(2, 33))
        sizes = rs.randint(1, 9, T)
        if rs.rand() < 0.3:                        # some callers with mid-size batches: not combined, overlapped across callers (gate_batch_overlapped)
            T = min(T, 6)
            sizes = sizes[:T]
            for t in range(T):
                if rs.rand() < 0.5:
                    sizes[t] = int(rs.randint(cus + 1, cus + (60 if p.n > 40 else 400)))
        log.append(f"T={T}")
        jobs = []
        for t in range(T):
            b = int(sizes[t])
            jobs.append((str(rs.choice(OPS2 + ["MUX"])), *(edge_rows(rs, words(rs, (b, n1))) for _ in range(3))))
        got, errs = [None] * T, []

        def work(t):
            try:
                op, a, b, c = jobs[t]
                r = a
                for _ in range(3):                 # dependent calls, so that the threads meet again and again
                    r = ctx.gate_batch(op, r, b, c if op == "MUX" else None)
                got[t] = r
            except Exception as e:                 # noqa: BLE001
                errs.append(repr(e))
        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        [x.start() for x in th]
        [x.join() for x in th]
        if errs:
            return False, "a concurrent caller failed: " + errs[0]
        for t in range(T):
            op, a, b, c = jobs[t]
            r = a
            for _ in range(3):
                r, _ = o.gate_batch(p, K.bsk, K.ksk, op, r, b, c if op == "MUX" else None)
            if not np.array_equal(got[t], r):
                return False, f"concurrent caller {t} ({op}) got different words"
        return True, ""
    if kind in ("gate1", "gates", "gates_dev"):
        a, b, c = (edge_rows(rs, words(rs, (B, n1))) for _ in range(3))
        if kind == "gate1":
            op = str(rs.choice(OPS2 + ["MUX"]))
            log.append("op=" + op)
            want, _ = o.gate_batch(p, K.bsk, K.ksk, op, a, b, c if op == "MUX" else None)
            got = ctx.gate_batch(op, a, b, c if op == "MUX" else None)
        else:
            ops = rs.randint(0, 11, B).astype(np.uint8)
            want, _ = o.gate_batch(p, K.bsk, K.ksk, ops, a, b, c)
            if kind == "gates":
                got = ctx.gate_batch(ops, a, b, c)
            else:
                dev = lambda x: torch.from_numpy(x.view(np.int32)).cuda()
                d_out = torch.empty((B, n1), dtype=torch.int32, device="cuda")
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    d_ops, d_a, d_b, d_c = torch.from_numpy(ops).cuda(), dev(a), dev(b), dev(c)
                    ctx.gate_batch_dev(d_ops, d_a, d_b, d_c, d_out, stream=s)
                s.synchronize()
                ctx.sync()
                got = d_out.cpu().numpy().view(np.uint32)
        return np.array_equal(got, want), "gate words differ"
    if kind in ("pbs", "pbs_items"):
        cts = edge_rows(rs, words(rs, (B, n1)))
        tv = words(rs, (B, 2, p.N)) if kind == "pbs_items" else words(rs, (2, p.N))
        if rs.rand() < 0.3:
            tv[..., 0, :] = 0                    # a look-up table proper: A = 0 (lut/generator.go)
        want, _ = o.bootstrap_batch(p, K.bsk, K.ksk, cts, tv)
        if rs.rand() < 0.4:
            # the device-pointer entry points on a side stream: blind rotate and key switch as two enqueued calls, or the fused one
            dev = lambda x: torch.from_numpy(x.view(np.int32)).cuda()
            s = torch.cuda.Stream()
            d_out = torch.empty((B, n1), dtype=torch.int32, device="cuda")
            with torch.cuda.stream(s):
                d_cts, d_tv = dev(cts), dev(tv)
                if rs.rand() < 0.5:
                    log.append("dev-two-calls")
                    d_acc = torch.empty((B, 2, p.N), dtype=torch.int32, device="cuda")
                    ctx.blind_rotate_batch_dev(d_cts, d_tv, d_acc, stream=s)
                    ctx.extract_keyswitch_batch_dev(d_acc, d_out, stream=s)
                else:
                    log.append("dev-fused")
                    ctx.bootstrap_batch_dev(d_cts, d_tv, d_out, stream=s)
            s.synchronize()
            ctx.sync()
            got = d_out.cpu().numpy().view(np.uint32)
        else:
            got = ctx.bootstrap_batch(cts, tv)
        return np.array_equal(got, want), "bootstrap words differ"
    cts = edge_rows(rs, words(rs, (min(B, 600), n1)))
    nsteps = int(rs.choice([-1, 0, 1, p.n // 2]))
    log.append(f"nsteps={nsteps}")
    acc = ctx.blind_rotate_batch(cts, K.tv, nsteps=nsteps)
    for i in rs.choice(len(cts), size=min(len(cts), 24), replace=False):
        if not np.array_equal(acc[i], o.blind_rotate(p, K.bsk, cts[i], K.tv, nsteps)):
            return False, f"accumulator {i} differs"
    trl = words(rs, (B, 2, p.N))
    got = ctx.extract_keyswitch_batch(trl)
    for i in rs.choice(B, size=min(B, 64), replace=False):
        if not np.array_equal(got[i], o.key_switch(p, K.ksk, o.sample_extract(trl[i]))):
            return False, f"key switch {i} differs"
    return True, ""


def seams(rs, o, K, ctx, B, log, exact):
    """The trgsw / trlwe seams with caller-supplied operands (round 6: tfhe_external_product_with, tfhe_cmux_with, tfhe_sample_extract_batch,
    tfhe_keyswitch_batch): a TRGSW operand handed over free-standing, the caller's own decomposition offset, any extraction index.
    exact: word equality with the oracle (N = 1024, L = 3); else one product within the stated 2^9 words of the oracle's fp64 product at the
    cloud key's offset, the integer seams exact as everywhere."""
    p = K.p
    j = int(rs.randint(p.n))
    std = o.offset(p)
    off = std if (not exact or rs.rand() < 0.6) else int(rs.randint(0, 2**32, dtype=np.uint64))
    log.append(f"gsw={j} off={'std' if off == std else hex(off)}")
    t0, t1 = words(rs, (B, 2, p.N)), words(rs, (B, 2, p.N))
    ep = ctx.external_product_with(K.bsk[j], t0, offset=off)
    cm = ctx.cmux_with(K.bsk[j], t0, t1, offset=off)
    for i in rs.choice(B, size=min(B, 6), replace=False):
        we, wc = o.external_product_at_offset(p, K.bsk[j], t0[i], off), o.cmux_at_offset(p, K.bsk[j], t0[i], t1[i], off)
        if exact:
            if not np.array_equal(ep[i], we) or not np.array_equal(cm[i], wc):
                return False, f"external product / CMUX {i} with a free-standing operand differs"
        else:
            for got, want in ((ep[i], we), (cm[i], wc)):
                e = (got.astype(np.int64) - want.astype(np.int64)) % 2**32
                if int(np.minimum(e, 2**32 - e).max()) > 2**9:
                    return False, f"external product / CMUX {i} more than 2^9 words from the oracle's"
    k = int(rs.choice([0, 1, p.N - 1, rs.randint(p.N)]))
    log.append(f"k={k}")
    ext = ctx.sample_extract_batch(t0, k)
    for i in rs.choice(B, size=min(B, 16), replace=False):
        if not np.array_equal(ext[i], o.sample_extract(np.ascontiguousarray(t0[i]), k)):
            return False, f"sample extract {i} at index {k} differs"
    lwe1 = edge_rows(rs, words(rs, (B, p.N + 1)))
    ks = ctx.keyswitch_batch(lwe1)
    for i in rs.choice(B, size=min(B, 24), replace=False):
        if not np.array_equal(ks[i], o.key_switch(p, K.ksk, lwe1[i])):
            return False, f"key switch of extracted sample {i} differs"
    return True, ""


def case_keys(rs, o, pkg, K, log):
    """The same cloud key brought onto the GPU another way -- uploaded in torus form (the engine transforms it: cloudkey.go:88-110 does
    that on the CPU), or as the exported blobs imported into a fresh context (host or device blobs), or as a clone staged through host
    memory -- must give the same words as the Fourier-form upload (and therefore as the oracle, which the other kinds check)."""
    import torch
    p, n1 = K.p, K.p.n + 1
    how = str(rs.choice(["torus-upload", "blob-import", "blob-import-dev", "host-staged-clone"]))
    B = int(rs.choice([1, 3, 40, 257, 600]))
    log.append(f"kind=keys B={B} how={how}")
    P = pkg.Params(n=p.n, N=p.N, Nbit=p.Nbit, L=p.L, Bgbit=p.Bgbit, basebit=p.basebit, t=p.t)
    other = None
    try:
        if how == "torus-upload":
            other = pkg.CloudKey(P, bsk_torus=K.bsk_torus, ksk=K.ksk)
            ctx = other.ctx
        elif how == "host-staged-clone":
            K.ctx.set_option("clone_force_host", 1)
            try:
                other = K.ck.clone_to(0)
            finally:
                K.ctx.set_option("clone_force_host", 0)
            ctx = other.ctx
            if ctx.get_option("clone_path") != 3:
                return False, "the forced host-staged clone did not take the host-staged path"
        else:
            ctx = pkg.Context(P)
            other = ctx
            for which in (0, 1):
                if how == "blob-import":
                    ctx.key_import(which, K.ctx.key_export(which))
                else:
                    ctx.key_import_dev(which, K.ctx.key_export_dev(which))
            torch.cuda.synchronize()
        ops = rs.randint(0, 11, B).astype(np.uint8)
        a, b, c = (edge_rows(rs, words(rs, (B, n1))) for _ in range(3))
        got, want = ctx.gate_batch(ops, a, b, c), K.ctx.gate_batch(ops, a, b, c)
        if K.name in UINT:
            return True, ""                      # tolerance regime: the torus upload's transform rounds differently; nothing to assert word-wise
        return np.array_equal(got, want), f"words differ between the key brought by {how} and the Fourier-form upload"
    finally:
        if other is not None:
            other.close()


def case_circuit(rs, o, K, log):
    """A random levelised circuit (any op incl. MUX, operands from any earlier wire) for C instances through the CircuitExecutor --
    as given, re-levelled by balance_levels / schedule_min_cost, eager or as a captured hipGraph replayed -- against the oracle
    evaluating the ORIGINAL gate list one level at a time: every wire identical (wires are single-assignment, so any topological
    levelling must compute the same words)."""
    import torch
    from go_tfhe_amd.circuits import CircuitExecutor, balance_levels, schedule_min_cost
    p, ctx = K.p, K.ctx
    n1 = p.n + 1
    W = int(rs.randint(2, 9))
    D = int(rs.randint(1, 7))
    C = int(rs.choice([1, 2, 7, 64, 130]))
    how = str(rs.choice(["as-given", "balanced", "min-cost", "captured", "captured-min-cost"]))
    log.append(f"kind=circuit C={C} depth={D} how={how}")
    set_options(rs, ctx, log)
    levels, nxt = [], W
    for _ in range(D):
        lvl = []
        avail = nxt                                 # operands come from wires of EARLIER levels
        for _ in range(int(rs.randint(1, 7))):
            op = str(rs.choice(OPS2 + ["MUX"]))
            i0, i1, i2 = (int(x) for x in rs.randint(0, avail, 3))
            lvl.append((op, i0, i1, i2 if op == "MUX" else None, nxt))
            nxt += 1
        levels.append(lvl)
    log.append(f"gates={sum(len(l) for l in levels)}")
    run_levels = levels
    if how == "balanced":
        run_levels = balance_levels(levels, int(rs.choice([1, 4, 1000])))
    elif how in ("min-cost", "captured-min-cost"):
        run_levels = schedule_min_cost(levels, C)
    if sorted(g[4] for l in run_levels for g in l) != list(range(W, nxt)):
        return False, "the re-levelled circuit lost or duplicated a gate"
    inputs = edge_rows(rs, words(rs, (W * C, n1))).reshape(W, C, n1)
    wires = torch.zeros((nxt, C, n1), dtype=torch.int32, device="cuda")
    wires[:W] = torch.from_numpy(inputs.view(np.int32)).cuda()
    ex = CircuitExecutor(ctx, run_levels, nxt)
    if how.startswith("captured"):
        graph = ex.capture(wires)
        try:
            wires[W:] = 0                           # the replay must recompute every wire from the inputs
            graph.replay()
            torch.cuda.synchronize()
            ctx.sync()
        finally:
            del graph
            ex.release()
    else:
        ex.run(wires)
        torch.cuda.synchronize()
        ctx.sync()
    got = wires.cpu().numpy().view(np.uint32)
    val = {w: inputs[w] for w in range(W)}
    for lvl in levels:
        for op, i0, i1, i2, out in lvl:
            val[out], _ = o.gate_batch(p, K.bsk, K.ksk, op, np.ascontiguousarray(val[i0]), np.ascontiguousarray(val[i1]),
                                       np.ascontiguousarray(val[i2]) if i2 is not None else None)
    for w in range(W, nxt):
        if not np.array_equal(got[w], val[w]):
            return False, f"wire {w} differs"
    return True, ""


def case_extended(rs, o, K, ext, log):
    """Programmable bootstraps through an EXTENDED lookup table (polyExtendFactor 2: the persistent eight-wave kernel; 4 and 9: one launch
    per CMUX step) -- the reference specifies these sets and skips them (params/uint_params_test.go:29-31), so the yardstick is the table:
    every item must decrypt to table[message]; one table for all items or one per item; batch sizes around the CU count."""
    from go_tfhe_amd.lut import Generator
    p, ctx = K.p, K.ctx
    m = {2: 64, 4: 128, 9: 256}[ext]
    B = int(rs.choice([1, 3, 64, 255, 256, 257, 300]))
    per_item = bool(rs.rand() < 0.3) and B <= 64
    log.append(f"kind=extended ext={ext} m={m} B={B} per_item={int(per_item)}")
    gen = Generator(p, m, polyExtendFactor=ext)
    shifts = rs.randint(0, m, B if per_item else 1)
    mul = int(rs.choice([1, 3, 5, 7]))
    f = lambda x, s: (mul * x + int(s)) % m
    luts = np.stack([gen.GenLookUpTableExtended(lambda x, s=s: f(x, s)) for s in shifts])
    msgs = rs.randint(0, m, B)
    rng = o.rng(int(rs.randint(1, 2**31)))
    cts = np.stack([o.encrypt_message(p, rng, int(x), m, K.s0) for x in msgs])
    out = ctx.bootstrap_extended_batch(cts, luts if per_item else luts[0])
    dec = np.array([o.decrypt_message(p, m, K.s0, np.ascontiguousarray(r)) for r in out])
    want = np.array([f(int(x), shifts[i] if per_item else shifts[0]) for i, x in enumerate(msgs)])
    bad = np.nonzero(dec != want)[0]
    return bad.size == 0, f"{bad.size} of {B} items decrypt to the wrong entry (first: item {bad[0] if bad.size else -1})"


def case_uint(rs, o, K, log):
    import torch
    p, ctx, m = K.p, K.ctx, UINT[K.name]
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    B = min(pick_batch(rs, cus, heavy=True), 700)
    log.append(f"kind=uint B={B} m={m}")
    set_options(rs, ctx, log)
    trl = words(rs, (B, 2, p.N))
    got = ctx.extract_keyswitch_batch(trl)
    for i in rs.choice(B, size=min(B, 32), replace=False):
        if not np.array_equal(got[i], o.key_switch(p, K.ksk, o.sample_extract(trl[i]))):
            return False, f"key switch {i} differs"
    if rs.rand() < 0.2:
        ok, why = seams(rs, o, K, ctx, min(B, 40), log, exact=False)
        if not ok:
            return ok, why
    table = rs.randint(0, m, m)
    tv = o.lut_generate(p, [int(x) for x in table])
    msgs = rs.randint(0, m, B)
    rng = o.rng(int(rs.randint(1, 2**31)))              # per case, so that --case reproduces it
    cts = np.stack([o.encrypt_message(p, rng, int(x), m, K.s0) for x in msgs])
    out = ctx.bootstrap_batch(cts, tv)
    dec = np.array([o.decrypt_message(p, m, K.s0, r) for r in out])
    if not np.array_equal(dec, table[msgs]):
        return False, "a programmable bootstrap decrypts to the wrong entry"
    # Tolerance regime: the first decomposition digit that rounds the other way adds a different key row, whose mask is uniform, so
    # from then on the engine's and the oracle's accumulators are DIFFERENT ENCRYPTIONS of (nearly) the same phase -- their words
    # are unrelated, and behind the key switch even the phases part by two independent rounding noises (measured 2^-7.8 at Uint2,
    # 2^-13.3 at Uint5: sqrt(2) x sqrt(N/2) x 2^-(basebit t)/sqrt(12), as the theory says).  What IS comparable:
    #   (1) ONE step from the engine's own previous state against the exact-integer CMUX increment, within the stated 2^9 words per
    #       coefficient (the oracle's own fp64 path measures ~400 at the L = 1 sets, the engine ~370);
    #   (2) the key switch, integer work: the oracle's key switch of the ENGINE's accumulator must equal the engine's output;
    #   (3) as a sanity bound, the accumulator's phase under the ring key at the extracted coefficient: apart by the decomposition
    #       and rounding noise of the steps only, sigma ~ 2^-Bgbit sqrt((1 + N/2)/12) per step and implementation; the measured
    #       distribution over 3,000 samples has that deviation but mixture tails (99.9 % at 5 sigma), hence 16 sigma sqrt(2 n).
    acc = ctx.blind_rotate_batch(cts, tv)
    s1 = K.s1.astype(np.int64)
    ph1 = lambda ext: (int(ext[-1]) - int((ext[:-1].astype(np.int64) * s1).sum())) % 2**32
    bound = 16.0 * 2.0 ** -p.Bgbit * ((1 + p.N / 2) / 12.0) ** 0.5 * (2.0 * p.n) ** 0.5 + 2.0 ** -20
    for i in rs.choice(B, size=min(B, 8), replace=False):
        ext = o.sample_extract(acc[i])
        if not np.array_equal(o.key_switch(p, K.ksk, ext), out[i]):
            return False, f"item {i}: the fused path's output is not the key switch of its own accumulator"
        want = o.sample_extract(o.blind_rotate(p, K.bsk, cts[i], tv))
        d = ((ph1(ext) - ph1(want) + 2**31) % 2**32 - 2**31) / 2.0**32
        if abs(d) > bound:
            return False, f"item {i}: accumulator phase {d:+.2e} away from the oracle's (bound {bound:.2e})"
    step = int(rs.randint(p.n))
    a0 = ctx.blind_rotate_batch(cts, tv, nsteps=step)
    a1 = ctx.blind_rotate_batch(cts, tv, nsteps=step + 1)
    sh = 32 - p.Nbit - 1
    for i in rs.choice(B, size=min(B, 3), replace=False):
        at = int(((int(cts[i, step]) + (1 << (sh - 1))) & 0xFFFFFFFF) >> sh)             # evaluator.go:122 (wraps)
        diff = np.stack([o.poly_mul_xk(a0[i, q], at) - a0[i, q] for q in range(2)])
        exact = a0[i] + o.external_product_exact(p, K.bsk_torus[step], diff)
        e = (a1[i].astype(np.int64) - exact.astype(np.int64)) % 2**32
        err = int(np.minimum(e, 2**32 - e).max())
        if err > 2**9:
            return False, f"item {i}: step {step} is {err} words away from the exact CMUX of the engine's own previous state"
    return True, ""


def run(seconds, seed, say, only_case=None):
    """Runs cases for `seconds` (or the one case `only_case`); returns (cases, stats); raises AssertionError with the repro
    line on the first mismatch."""
    graft.build()
    pkg = graft.load_package()
    from oracle_lib import Oracle
    o = Oracle()
    keys = {}

    def key(name, n):
        if (name, n) not in keys:
            keys[(name, n)] = Key(o, pkg, name, n, 0x7F4E0F00 + 97 * n + sum(map(ord, name)))
        return keys[(name, n)]

    t0 = time.time()
    stats = {}
    k = only_case if only_case is not None else 0
    try:
        while True:
            rs = np.random.RandomState((seed * 1000003 + k) % 2**32)
            log = []
            if rs.rand() < 0.85:
                name, n = str(rs.choice(EXACT)), int(rs.choice([1, 2, 5, 16, 24, 33, 64]))
                if rs.rand() < 0.04:
                    n = FULL_N[name]                       # the parameter set as the reference ships it
                if n <= 64 and rs.rand() < 0.05:
                    ok, why = case_keys(rs, o, pkg, key(name, n), log)
                elif n <= 33 and rs.rand() < 0.12:
                    ok, why = case_circuit(rs, o, key(name, n), log)
                else:
                    ok, why = case_exact(rs, o, key(name, n), log)
            elif rs.rand() < 0.25:
                ext = int(rs.choice([2, 4, 9]))
                name, n = ("uint5" if ext == 2 else "uint7"), (4 if ext == 9 else int(rs.choice([4, 12])))
                ok, why = case_extended(rs, o, key(name, n), ext, log)
            else:
                name, n = str(rs.choice(list(UINT))), int(rs.choice([4, 12]))
                ok, why = case_uint(rs, o, key(name, n), log)
            kind = log[0].split()[0]
            stats[kind] = stats.get(kind, 0) + 1
            say(f"case {k}: set={name} n={n} {' '.join(log)} -> {'ok' if ok else 'MISMATCH: ' + why}")
            if not ok:
                raise AssertionError(f"case {k} ({name}, n={n}, {' '.join(log)}): {why}; "
                                     f"REPRO: python tests/fuzz_gpu.py --seed {seed} --case {k}")
            k += 1
            if only_case is not None or time.time() - t0 > seconds:
                break
    finally:
        for K in keys.values():
            K.close()
    say(f"{k if only_case is None else 1} cases in {time.time() - t0:.0f} s, 0 mismatches; by kind: {stats}; contexts: {len(keys)}")
    return k, stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=2.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--log", default=None)
    ap.add_argument("--case", type=int, default=None, help="run this one case (repro)")
    a = ap.parse_args()
    out = open(a.log, "w") if a.log else sys.stdout
    say = lambda s: (out.write(s + "\n"), out.flush())
    try:
        run(a.minutes * 60, a.seed, say, a.case)
    except AssertionError as e:
        say(str(e))
        sys.exit(1)


if __name__ == "__main__":
    main()

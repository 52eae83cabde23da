"""Levelised gate circuits over the batch engine.

A circuit is a list of levels; a level is a list of gates (op, in0, in1[, in2], out) over wire
ids.  All gates of one level, for all C independent circuit instances, go to the GPU as ONE
tfhe_gate_batch_dev call with per-item op codes; wires live in one device tensor
[n_wires][C][n+1].  This is the reference's README ripple-carry adder (README.md:78-115,
examples/EXAMPLES_GUIDE.md:134-144: FullAdder = XOR, AND, AND, XOR, OR -> 5 gates/bit) turned
from 40 sequential gates.* calls into 17 dependency levels of batched bootstraps.
"""
import numpy as np

from ._binding import OPS


def ripple_carry_adder(bits, fold_carry_in=True):
    """Wires: a[i] = i, b[i] = bits+i, sum[i] = 2*bits+i, carry-out = 3*bits.
    Returns (levels, n_wires, sum_wires, carry_wire).

    fold_carry_in=False is the circuit exactly as the reference writes it (README.md:78-106): `bits` FullAdders
    of five gates each -- XOR(a,b), AND(a,b), AND(a^b, cin), XOR(a^b, cin), OR(a&b, (a^b)&cin) -- chained
    from carry := gates.Constant(false): 5*bits gates (40 for 8 bits) in 1 + 2*bits levels (17).  The constant
    lives on wire adder_constant_wire(bits); fill it with gates.Constant(False, params) (= the reference's
    trivial sample with body 1 - 1/8 = 0xE0000001, gates.go:61-69) before running.

    fold_carry_in=True (default) drops the three gates that only combine with that constant: s0 = a0 XOR b0,
    c1 = a0 AND b0 -- 5*bits - 3 gates (37), 2*bits - 1 levels (15).  Same decrypted sums; sum[0] and the carries
    are different ciphertexts (one bootstrap fewer on their path)."""
    a = lambda i: i
    b = lambda i: bits + i
    s = lambda i: 2 * bits + i
    nxt = [3 * bits + 2]                      # 3*bits + 1 is reserved for the constant-false wire

    def new():
        nxt[0] += 1
        return nxt[0] - 1

    levels = []
    x = [new() for _ in range(bits)]          # a_i XOR b_i
    g = [new() for _ in range(bits)]          # a_i AND b_i
    lvl = []
    for i in range(bits):
        lvl.append(("XOR", a(i), b(i), None, s(0) if i == 0 and fold_carry_in else x[i]))
        # folded: a 1-bit adder's carry-out IS a_0 AND b_0: route it to the carry wire directly
        lvl.append(("AND", a(i), b(i), None, 3 * bits if bits == 1 and fold_carry_in else g[i]))
    levels.append(lvl)
    carry = g[0] if fold_carry_in else adder_constant_wire(bits)
    for i in range(1 if fold_carry_in else 0, bits):
        t = new()
        # sum_i = x_i XOR c_i ; t = x_i AND c_i        (one level)
        levels.append([("XOR", x[i], carry, None, s(i)), ("AND", x[i], carry, None, t)])
        # c_{i+1} = g_i OR t                              (next level)
        c_out = 3 * bits if i == bits - 1 else new()
        levels.append([("OR", g[i], t, None, c_out)])
        carry = c_out
    return levels, nxt[0], [s(i) for i in range(bits)], 3 * bits


def adder_constant_wire(bits):
    """Wire that holds gates.Constant(false) for ripple_carry_adder(bits, fold_carry_in=False)."""
    return 3 * bits + 1


def count_gates(levels):
    return sum(len(l) for l in levels)


def count_bootstraps(levels):
    return sum(3 if g[0] == "MUX" else 1 for l in levels for g in l)


def balance_levels(levels, width):
    """Re-level a circuit WITHOUT lengthening its critical path so that levels are at most `width`
    bootstraps wide wherever slack allows.  Every gate keeps a level between its earliest (all
    producers done) and latest (no consumer delayed) position; each level first takes the gates
    that must run now, then fills up to `width` with the ready gates of least slack.  One launch
    costs a fixed ~2.4 ms of blind-rotate latency plus ~1.1 ms per 256 bootstraps up to one full
    launch, so the cheapest schedule is the one with the fewest launches: a ripple-carry adder's
    first level (all a_i XOR b_i, a_i AND b_i at once: four launches for 256 circuits) is spread
    under the narrow carry-chain levels instead.  Wires are single-assignment, so any topological
    levelling computes bit-identical results."""
    gates = [g for lvl in levels for g in lvl]
    D = len(levels)
    producer = {g[4]: i for i, g in enumerate(gates)}
    deps = [[producer[w] for w in (g[1], g[2], g[3]) if w is not None and w in producer] for g in gates]
    users = [[] for _ in gates]
    for i, d in enumerate(deps):
        for j in d:
            users[j].append(i)
    alap = [D - 1] * len(gates)
    for i in reversed(range(len(gates))):           # the given order is topological
        for u in users[i]:
            alap[i] = min(alap[i], alap[u] - 1)
    weight = [3 if g[0] == "MUX" else 1 for g in gates]
    done_at = [None] * len(gates)
    out = []
    pending = set(range(len(gates)))
    for l in range(D):
        ready = sorted((i for i in pending if all(done_at[j] is not None and done_at[j] < l for j in deps[i])),
                       key=lambda i: (alap[i], i))
        lvl, used = [], 0
        for i in ready:
            if alap[i] <= l or used + weight[i] <= width:
                lvl.append(i); used += weight[i]
        for i in lvl:
            done_at[i] = l
            pending.discard(i)
        out.append([gates[i] for i in lvl])
    assert not pending, "balance_levels: unschedulable gates (input order not topological?)"
    return [lvl for lvl in out if lvl]


def launch_cost_ms(bootstraps, full=1024):
    """Measured time of one level of `bootstraps` gate bootstraps on one MI355X (blind rotate + key switch, 128-bit set; round 4:
    profiles/r04_b_oct_floor.txt, r04_f_fold8_ab.txt, one box each): a step function of the launch shape -- up to 256 (one
    bootstrap per CU) run the eight-wave kernel, up to 512 the two-wave kernel with two bootstraps per four-wave workgroup (one per
    CU), then three two-wave workgroups per CU, then two four-wave workgroups per CU (both with phase priorities); longer levels
    are full launches plus a tail."""
    steps = ((256, 2.30), (512, 3.98), (768, 4.61), (1024, 5.50))
    n_full, rem = divmod(int(bootstraps), full)
    t = n_full * steps[-1][1]
    if rem:
        t += next(c for lim, c in steps if rem <= lim)
    return t


def schedule_min_cost(levels, instances, cost=launch_cost_ms):
    """Re-level a circuit for `instances` parallel instances so that the SUM of the levels' launch costs is small, without
    lengthening its critical path.  Starts from balance_levels' schedule and moves single gates between the levels their
    dependencies allow while that lowers the total (deterministic local search): with a step-shaped cost it pays to
    fill the levels that are launched anyway up to a shape boundary and to keep the carry-chain levels at the smallest
    shape.  8-bit adder x 256: 73.6 -> 72.4 ms by the cost model.  Any topological levelling computes bit-identical
    results (wires are single-assignment)."""
    D = len(levels)
    start = balance_levels(levels, max(1, 1024 // max(1, instances)))
    gates = [g for lvl in levels for g in lvl]              # the given order is topological
    index = {g[4]: i for i, g in enumerate(gates)}
    deps = [[index[w] for w in (g[1], g[2], g[3]) if w is not None and w in index] for g in gates]
    users = [[] for _ in gates]
    for i, d in enumerate(deps):
        for j in d:
            users[j].append(i)
    weight = [3 if g[0] == "MUX" else 1 for g in gates]
    at = [0] * len(gates)
    for l, lvl in enumerate(start):
        for g in lvl:
            at[index[g[4]]] = l
    load = [0] * D
    for i, l in enumerate(at):
        load[l] += weight[i] * instances
    total = sum(cost(b) for b in load if b)
    improved = True
    while improved:
        improved = False
        for i in range(len(gates)):
            lo = max([at[j] + 1 for j in deps[i]] + [0])
            hi = min([at[u] - 1 for u in users[i]] + [D - 1])
            w = weight[i] * instances
            for cand in range(lo, hi + 1):
                src = at[i]
                if cand == src:
                    continue
                delta = (cost(load[src] - w) if load[src] - w else 0.0) + cost(load[cand] + w) \
                    - cost(load[src]) - (cost(load[cand]) if load[cand] else 0.0)
                if delta < -1e-9:
                    load[src] -= w; load[cand] += w; at[i] = cand
                    total += delta
                    improved = True
    out = [[] for _ in range(D)]
    for i, l in enumerate(at):
        out[l].append(gates[i])
    return [lvl for lvl in out if lvl]


class CircuitExecutor:
    """Runs a levelised circuit for C instances at once on one GPU context (torch tensors)."""

    def __init__(self, ctx, levels, n_wires):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.levels = levels
        self.n_wires = n_wires
        self.n1 = ctx.params.n + 1
        dev = torch.device("cuda", ctx.device)
        self._plan = []
        for lvl in levels:
            ops = np.array([OPS[g[0]] for g in lvl], np.uint8)
            i0 = torch.tensor([g[1] for g in lvl], device=dev)
            i1 = torch.tensor([g[2] for g in lvl], device=dev)
            has_c = any(g[3] is not None for g in lvl)
            i2 = torch.tensor([g[3] if g[3] is not None else g[1] for g in lvl], device=dev) if has_c else None
            out = torch.tensor([g[4] for g in lvl], device=dev)
            uniform = lvl[0][0] if len(set(g[0] for g in lvl)) == 1 else None
            self._plan.append((ops, uniform, i0, i1, i2, out))
        self._op_cache = {}                      # (level, C) -> per-item op codes on the device
        self._captured = False

    def capture(self, wires, max_instances=None):
        """Record run(wires) into a HIP graph (torch.cuda.CUDAGraph) and return it; replay() re-runs the whole
        circuit on whatever the wire tensor holds at that time.  tfhe_gate_batch_dev only enqueues (no read-back,
        no synchronisation), so the level loop captures as is.  Before capturing, the context's intermediate buffers
        are sized for the WIDEST LEVEL of this schedule at `max_instances` circuit instances (default: the instances
        of `wires`; pass a larger number if the same context will later run, or capture, wider batches -- the library
        freezes the buffers at the first captured call and refuses any later call that would have to grow them until
        release() is called; include/tfhe_hip.h, tfhe_ctx_reserve).  One un-captured run fills the op-code cache and
        warms torch's allocator (neither may allocate during capture)."""
        torch = self.torch
        C = int(max_instances or wires.shape[1])
        widest = max(sum(3 if g[0] == "MUX" else 1 for g in lvl) for lvl in self.levels)      # a MUX item is three bootstraps
        try:
            self.ctx.reserve(widest * C, with_mux=any(g[0] == "MUX" for lvl in self.levels for g in lvl))
        except Exception as e:
            if self.ctx.get_option("frozen"):
                # another executor's graph already froze this context at a smaller size: say what to do about it here,
                # not only in the library's words
                raise RuntimeError(
                    f"the context is frozen by {getattr(self.ctx, '_graphs_alive', 0)} captured graph(s) and its buffers are too small "
                    f"for this schedule's widest level ({widest} bootstraps x {C} instances): capture the FIRST graph of a shared "
                    f"context with max_instances large enough for every later one, or release() the earlier executors first "
                    f"({e})") from e
            raise
        self.run(wires)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.run(wires)
        # graphs alive per CONTEXT, not per executor: several executors may capture on one context, and the context may only
        # be un-frozen (buffers allowed to move) when the last of them has let go
        self.ctx._graphs_alive = getattr(self.ctx, "_graphs_alive", 0) + 1
        self._captured_count = getattr(self, "_captured_count", 0) + 1
        self._captured = True
        return graph

    def release(self):
        """Call once every graph captured through THIS executor has been destroyed.  The context counts the captures of all
        its executors; its intermediate buffers are un-frozen (TFHE_OPT_FROZEN = 0: later, larger batches may grow them
        again) only when the last executor holding graphs has released -- another executor's replays keep their addresses."""
        self.ctx.sync()
        mine = getattr(self, "_captured_count", 0)
        alive = max(0, getattr(self.ctx, "_graphs_alive", 0) - mine)
        self.ctx._graphs_alive = alive
        self._captured_count = 0
        self._captured = False
        if alive == 0:
            self.ctx.set_option("frozen", 0)

    def run(self, wires, stream=None):
        """wires: int32 tensor [n_wires][C][n+1] with the input wires filled; updated in place.
        Everything -- the operand gathers, the gate kernels and the result scatter -- is enqueued on `stream`
        (default: torch's current stream), so the temporaries also belong to that stream's allocator pool."""
        torch = self.torch
        C = wires.shape[1]
        stream = stream or torch.cuda.current_stream()
        with torch.cuda.stream(stream):
            for li, (ops, uniform, i0, i1, i2, out) in enumerate(self._plan):
                G = i0.shape[0]
                a = wires.index_select(0, i0).reshape(G * C, self.n1)
                b = wires.index_select(0, i1).reshape(G * C, self.n1)
                c = wires.index_select(0, i2).reshape(G * C, self.n1) if i2 is not None else None
                res = torch.empty_like(a)
                if uniform is not None:
                    self.ctx.gate_batch_dev(uniform, a, b, c, res, stream)
                else:
                    op_t = self._op_cache.get((li, C))
                    if op_t is None:
                        op_t = self._op_cache[(li, C)] = torch.from_numpy(np.repeat(ops, C)).to(a.device)
                    self.ctx.gate_batch_dev(op_t, a, b, c, res, stream)
                wires.index_copy_(0, out, res.reshape(G, C, self.n1))
        return wires

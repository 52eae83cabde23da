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


def ripple_carry_adder(bits):
    """Wires: a[i] = i, b[i] = bits+i, sum[i] = 2*bits+i, carry-out = 3*bits.
    Returns (levels, n_wires, sum_wires, carry_wire).  carry-in = Constant(false) is folded away
    as in the reference example's first half adder: s0 = a0 XOR b0, c1 = a0 AND b0."""
    a = lambda i: i
    b = lambda i: bits + i
    s = lambda i: 2 * bits + i
    nxt = [3 * bits + 1]

    def new():
        nxt[0] += 1
        return nxt[0] - 1

    levels = []
    x = [new() for _ in range(bits)]          # a_i XOR b_i
    g = [new() for _ in range(bits)]          # a_i AND b_i
    lvl = []
    for i in range(bits):
        lvl.append(("XOR", a(i), b(i), None, s(0) if i == 0 else x[i]))
        lvl.append(("AND", a(i), b(i), None, g[i]))
    levels.append(lvl)
    carry = g[0]
    for i in range(1, bits):
        t = new()
        # sum_i = x_i XOR c_i ; t = x_i AND c_i        (one level)
        levels.append([("XOR", x[i], carry, None, s(i)), ("AND", x[i], carry, None, t)])
        # c_{i+1} = g_i OR t                              (next level)
        c_out = 3 * bits if i == bits - 1 else new()
        levels.append([("OR", g[i], t, None, c_out)])
        carry = c_out
    return levels, nxt[0], [s(i) for i in range(bits)], 3 * bits


def count_gates(levels):
    return sum(len(l) for l in levels)


def count_bootstraps(levels):
    return sum(3 if g[0] == "MUX" else 1 for l in levels for g in l)


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

    def run(self, wires, stream=None):
        """wires: int32 tensor [n_wires][C][n+1] with the input wires filled; updated in place."""
        torch = self.torch
        C = wires.shape[1]
        stream = stream or torch.cuda.current_stream()
        for ops, uniform, i0, i1, i2, out in self._plan:
            G = i0.shape[0]
            a = wires.index_select(0, i0).reshape(G * C, self.n1)
            b = wires.index_select(0, i1).reshape(G * C, self.n1)
            c = wires.index_select(0, i2).reshape(G * C, self.n1) if i2 is not None else None
            res = torch.empty_like(a)
            if uniform is not None:
                self.ctx.gate_batch_dev(uniform, a, b, c, res, stream)
            else:
                op_t = torch.from_numpy(np.repeat(ops, C)).to(a.device)
                self.ctx.gate_batch_dev(op_t, a, b, c, res, stream)
            wires.index_copy_(0, out, res.reshape(G, C, self.n1))
        return wires

"""gates.* of the reference (gates/gates.go:26-126,156-312) on top of the HIP engine.

Same names and argument meaning as the Go package: a Ciphertext is one LWE sample, a numpy
uint32 vector of n+1 words with the body last (tlwe.go:11-33); `ck` is a CloudKey.  Scalar
gates are a batch of one; Batch* take a sequence of pairs like the reference's
[][2]*Ciphertext.  Errors raise TfheError where the reference panics.
"""
import numpy as np

from ._binding import OPS


def _one(op, a, b, ck, c=None):
    out = ck.ctx.gate_batch(op, np.asarray(a)[None, :], np.asarray(b)[None, :],
                            None if c is None else np.asarray(c)[None, :])
    return out[0]


def NAND(a, b, ck): return _one("NAND", a, b, ck)      # gates.go:26-31
def OR(a, b, ck): return _one("OR", a, b, ck)          # gates.go:34-37
def AND(a, b, ck): return _one("AND", a, b, ck)        # gates.go:40-43
def XOR(a, b, ck): return _one("XOR", a, b, ck)        # gates.go:46-49
def XNOR(a, b, ck): return _one("XNOR", a, b, ck)      # gates.go:52-58
def NOR(a, b, ck): return _one("NOR", a, b, ck)        # gates.go:72-76
def ANDNY(a, b, ck): return _one("ANDNY", a, b, ck)    # gates.go:79-83
def ANDYN(a, b, ck): return _one("ANDYN", a, b, ck)    # gates.go:86-90
def ORNY(a, b, ck): return _one("ORNY", a, b, ck)      # gates.go:93-97
def ORYN(a, b, ck): return _one("ORYN", a, b, ck)      # gates.go:100-104
def MUX(a, b, c, ck): return _one("MUX", a, b, ck, c)  # gates.go:107-114


def NOT(a):
    """gates.go:117-119: negation of every word, no bootstrap."""
    return (0 - np.asarray(a, dtype=np.uint32)).astype(np.uint32)


def Copy(a):
    """gates.go:122-126"""
    return np.array(a, dtype=np.uint32, copy=True)


def Constant(value, params):
    """gates.go:61-69 (trivial sample; reproduces the reference's `1 - mu` for false)."""
    mu = np.uint32(0x20000000)
    ct = np.zeros(params.n + 1, np.uint32)
    ct[params.n] = mu if value else np.uint32((1 - int(mu)) & 0xFFFFFFFF)
    return ct


def _batch(op, inputs, ck):
    a = np.stack([np.asarray(p[0], dtype=np.uint32) for p in inputs])
    b = np.stack([np.asarray(p[1], dtype=np.uint32) for p in inputs])
    return list(ck.ctx.gate_batch(op, a, b))


def BatchNAND(inputs, ck): return _batch("NAND", inputs, ck)   # gates.go:156-182
def BatchAND(inputs, ck): return _batch("AND", inputs, ck)     # gates.go:185-208
def BatchOR(inputs, ck): return _batch("OR", inputs, ck)       # gates.go:211-234
def BatchXOR(inputs, ck): return _batch("XOR", inputs, ck)     # gates.go:237-260
def BatchNOR(inputs, ck): return _batch("NOR", inputs, ck)     # gates.go:263-286


def BatchXNOR(inputs, ck):
    """gates.go:289-312, with the sign of the tested scalar XNOR."""
    return _batch("XNOR", inputs, ck)


def gate_stream(ops, a, b, ck, c=None):
    """Mixed stream of gates (BASELINE config 5): ops is a sequence of names or op codes."""
    codes = np.array([OPS[o] if isinstance(o, str) else int(o) for o in ops], dtype=np.uint8)
    return ck.ctx.gate_batch(codes, a, b, c)

"""Lookup tables for the programmable bootstrap -- host mirror of the reference's `lut` package
(lut/lut.go:13-45, lut/encoder.go:10-107, lut/generator.go:10-173) and of utils.F64ToTorus /
TorusToF64 (utils/utils.go:11-19).  Plain numpy on the host: a table is built once per function and
handed to Evaluator.BootstrapLUT / tfhe_bootstrap_batch as a TRLWE test vector.

Differences from the reference, on purpose: the ring degree comes from an explicit Params object
instead of the process-global params.GetTRGSWLv1(); tables are filled with array operations instead
of per-coefficient loops.  Values are identical (tests/test_abi_and_host.py pins them against the
oracle and the committed golden table).
"""
import numpy as np

TWO32 = 4294967296.0


def f64_to_torus(d):
    """utils.F64ToTorus (utils.go:11-14): the fractional part (sign kept, as math.Mod) times 2^32,
    truncated toward zero, wrapped to 32 bits.  Scalar or array."""
    v = (np.fmod(np.asarray(d, np.float64), 1.0) * TWO32).astype(np.int64) & 0xFFFFFFFF
    return v.astype(np.uint32) if v.ndim else np.uint32(v)


def torus_to_f64(t):
    """utils.TorusToF64 (utils.go:17-19)."""
    return np.asarray(t, np.uint32).astype(np.float64) / TWO32


def div_round(a, b):
    """generator.go:171-173: integer division rounding half up (non-negative operands)."""
    return (a + b // 2) // b


class LookUpTable:
    """lut.LookUpTable (lut.go:13-45): a TRLWE sample [2][N]; row 0 = A (zero for generated tables),
    row 1 = B.  `poly` is what the bootstrap entry points take as the test vector."""

    def __init__(self, N):
        self.poly = np.zeros((2, N), np.uint32)

    A = property(lambda self: self.poly[0])
    B = property(lambda self: self.poly[1])

    def Copy(self):
        out = LookUpTable(self.poly.shape[1])
        out.poly[...] = self.poly
        return out

    def CopyFrom(self, other):
        self.poly[...] = other.poly

    def Clear(self):
        self.poly[...] = 0


class Encoder:
    """lut.Encoder (encoder.go:10-107): message i of a modulus-m space sits at i * scale on the torus,
    scale = 1/(2m) unless given."""

    def __init__(self, messageModulus, scale=None):
        self.MessageModulus = int(messageModulus)
        self.Scale = 1.0 / (2 * self.MessageModulus) if scale is None else float(scale)

    def Encode(self, message):
        return self.EncodeWithCustomScale(message, self.Scale)

    def EncodeWithCustomScale(self, message, scale):
        m = np.mod(np.asarray(message, np.int64), self.MessageModulus)        # Go's % plus the negative fix-up
        return f64_to_torus(m.astype(np.float64) * scale)

    def Decode(self, value):
        m = np.mod((torus_to_f64(value) / self.Scale + 0.5).astype(np.int64), self.MessageModulus)
        return m if m.ndim else int(m)

    def DecodeBool(self, value):
        d = self.Decode(value)
        return d != 0


class Generator:
    """lut.Generator (generator.go:10-173) for one parameter set.  The reference only builds tables with
    LookUpTableSize = N (polyExtendFactor = 1, generator.go:19-20); polyExtendFactor > 1 gives the EXTENDED tables
    its Uint6/7/8 parameter sets are specified for (params.go:399-402,440-443,481-484) and that it leaves
    unimplemented: the same construction over LookUpTableSize = polyExtendFactor * N positions, handed to the
    engine de-interleaved (GenLookUpTableExtended, tfhe_bootstrap_extended_batch)."""

    def __init__(self, params, messageModulus, scale=None, polyExtendFactor=1):
        self.PolyDegree = int(params.N)
        self.PolyExtendFactor = int(polyExtendFactor)
        self.LookUpTableSize = self.PolyDegree * self.PolyExtendFactor
        self.Encoder = Encoder(messageModulus, scale)

    def _layout(self, modulus):
        """Message index and sign for every coefficient of the rotated table (generator.go:62-93):
        message x owns raw positions [div_round(x*N, m), div_round((x+1)*N, m)); the table is read
        from position offset = div_round(N, 2m) on, and the part that wrapped around is negated."""
        N = self.LookUpTableSize
        bounds = div_round(np.arange(modulus + 1, dtype=np.int64) * N, modulus)
        offset = div_round(N, 2 * modulus)
        src = (np.arange(N) + offset) % N
        msg = np.searchsorted(bounds, src, side="right") - 1
        return msg, np.arange(N) >= N - offset

    def _table(self, values, modulus):
        msg, neg = self._layout(modulus)
        b = np.asarray(values, np.uint32)[msg]
        return np.where(neg, (0 - b.astype(np.int64)) & 0xFFFFFFFF, b).astype(np.uint32)

    def _fill(self, values, modulus, out):
        if self.PolyExtendFactor != 1:
            raise ValueError("a LookUpTable holds N coefficients: use GenLookUpTableExtended for polyExtendFactor > 1")
        out.poly[0] = 0
        out.poly[1] = self._table(values, modulus)
        return out

    def GenLookUpTableExtended(self, f, full=False):
        """The table of f over LookUpTableSize = ext * N positions as [ext][2][N] uint32: component k holds the
        coefficients of Y^(i*ext + k) of the big polynomial (A parts zero), the layout tfhe_bootstrap_extended_batch
        takes.  ext = 1 gives GenLookUpTable(f).poly[None].  full=True: f returns raw torus values."""
        m, ext, N = self.Encoder.MessageModulus, self.PolyExtendFactor, self.PolyDegree
        vals = [int(f(x)) & 0xFFFFFFFF for x in range(m)] if full else self.Encoder.Encode([int(f(x)) for x in range(m)])
        big = self._table(vals, m)                         # [ext * N]
        out = np.zeros((ext, 2, N), np.uint32)
        out[:, 1, :] = big.reshape(N, ext).T
        return out

    def GenLookUpTable(self, f):
        return self.GenLookUpTableAssign(f, LookUpTable(self.PolyDegree))

    def GenLookUpTableAssign(self, f, lutOut):
        m = self.Encoder.MessageModulus
        return self._fill(self.Encoder.Encode([int(f(x)) for x in range(m)]), m, lutOut)

    def GenLookUpTableFull(self, f):
        return self.GenLookUpTableFullAssign(f, LookUpTable(self.PolyDegree))

    def GenLookUpTableFullAssign(self, f, lutOut):
        m = self.Encoder.MessageModulus
        return self._fill([int(f(x)) & 0xFFFFFFFF for x in range(m)], m, lutOut)

    def GenLookUpTableCustom(self, f, messageModulus, scale):
        enc = Encoder(messageModulus, scale)
        return self._fill(enc.Encode([int(f(x)) for x in range(messageModulus)]), messageModulus,
                          LookUpTable(self.PolyDegree))

    def ModSwitch(self, x):
        """generator.go:159-168: torus -> [0, LookUpTableSize), round half away from zero like math.Round."""
        scaled = float(np.uint32(x)) / TWO32 * self.LookUpTableSize
        return int(np.floor(scaled + 0.5)) % self.LookUpTableSize

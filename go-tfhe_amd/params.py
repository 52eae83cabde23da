"""Parameter sets of the reference (params/params.go:83-112,117-146,151-180,362-391).

The reference selects one of these through the process-global params.CurrentSecurityLevel
(params.go:47); here a set is an explicit value captured by the context.
"""
import ctypes as C


class Params(C.Structure):
    """Mirror of tfhe_params (include/tfhe_hip.h)."""
    _fields_ = [("n", C.c_int32), ("N", C.c_int32), ("Nbit", C.c_int32), ("L", C.c_int32),
                ("Bgbit", C.c_int32), ("basebit", C.c_int32), ("t", C.c_int32)]

    @property
    def base(self):
        return 1 << self.basebit

    @property
    def ksk_rows(self):
        return self.N * self.t * self.base

    def replace(self, **kw):
        vals = {f: getattr(self, f) for f, _ in self._fields_}
        vals.update(kw)
        return Params(**vals)

    def __repr__(self):
        return "Params(" + ", ".join(f"{f}={getattr(self, f)}" for f, _ in self._fields_) + ")"


Security80Bit = Params(n=550, N=1024, Nbit=10, L=3, Bgbit=6, basebit=2, t=7)
Security110Bit = Params(n=630, N=1024, Nbit=10, L=3, Bgbit=6, basebit=2, t=8)
Security128Bit = Params(n=700, N=1024, Nbit=10, L=3, Bgbit=6, basebit=2, t=9)
SecurityUint1 = Params(n=700, N=1024, Nbit=10, L=2, Bgbit=10, basebit=2, t=8)      # params.go:194-232
SecurityUint2 = Params(n=687, N=512, Nbit=9, L=1, Bgbit=18, basebit=4, t=3)        # params.go:236-265 (run at rank 1, as the reference does)
SecurityUint3 = Params(n=820, N=1024, Nbit=10, L=1, Bgbit=23, basebit=6, t=2)      # params.go:277-313
SecurityUint4 = Params(n=820, N=2048, Nbit=11, L=1, Bgbit=22, basebit=5, t=3)      # params.go:318-354
SecurityUint5 = Params(n=1071, N=2048, Nbit=11, L=1, Bgbit=22, basebit=6, t=3)     # params.go:362-398
SecurityUint6 = Params(n=1071, N=2048, Nbit=11, L=1, Bgbit=22, basebit=6, t=3)     # params.go:403-439
SecurityUint7 = Params(n=1160, N=2048, Nbit=11, L=1, Bgbit=22, basebit=7, t=3)     # params.go:444-480
SecurityUint8 = Params(n=1160, N=2048, Nbit=11, L=1, Bgbit=22, basebit=7, t=3)     # params.go:485-521

BY_NAME = {"80": Security80Bit, "110": Security110Bit, "128": Security128Bit, "uint1": SecurityUint1, "uint2": SecurityUint2,
           "uint3": SecurityUint3, "uint4": SecurityUint4, "uint5": SecurityUint5, "uint6": SecurityUint6,
           "uint7": SecurityUint7, "uint8": SecurityUint8}

"""evaluator.Evaluator of the reference (evaluator/evaluator.go, programmable_bootstrap.go)
bound to a GPU context.  Method names follow the Go type; the *Assign forms write into a
caller-provided array, the others return a fresh one (the reference's pooled-result aliasing,
evaluator.go:150-151, is not reproduced: every output owns its storage).
The bsk / ksk / decompositionOffset arguments of the Go signatures are the ones already
resident in the CloudKey's context.
"""
import numpy as np


class Evaluator:
    def __init__(self, ck):
        self.ck = ck
        self.ctx = ck.ctx
        self.params = ck.params

    # evaluator.go:50-81.  ctFourierGGSW: an index into the resident bootstrapping key, or ANY TRGSW sample as the reference hands it over
    # ([2L][2][N] float64 in its FourierPoly layout; then decompositionOffset, default the cloud key's, is a kernel operand)
    def ExternalProductAssign(self, ctFourierGGSW, ct_in, ct_out, decompositionOffset=None):
        if np.ndim(ctFourierGGSW) == 0:
            ct_out[...] = self.ctx.external_product_batch(int(ctFourierGGSW), np.asarray(ct_in)[None])[0]
        else:
            ct_out[...] = self.ctx.external_product_with(ctFourierGGSW, np.asarray(ct_in)[None], decompositionOffset)[0]

    # evaluator.go:85-106 : ctOut = ct0 + ctCond (x) (ct1 - ct0)
    def CMuxAssign(self, ctCond, ct0, ct1, ct_out, decompositionOffset=None):
        ct_out[...] = self.ctx.cmux_with(ctCond, np.asarray(ct0)[None], np.asarray(ct1)[None], decompositionOffset)[0]

    # evaluator.go:110-135 (nsteps < n stops the CMUX chain early: CMuxAssign seam, :85-106)
    def BlindRotateAssign(self, ct_in, testvec, ct_out, nsteps=-1):
        ct_out[...] = self.ctx.blind_rotate_batch(np.asarray(ct_in)[None], testvec, nsteps)[0]

    # evaluator.go:139-148
    def BootstrapAssign(self, ct_in, testvec, ct_out):
        ct_out[...] = self.ctx.bootstrap_batch(np.asarray(ct_in)[None], testvec)[0]

    # evaluator.go:152-157
    def Bootstrap(self, ct_in, testvec=None):
        return self.ctx.bootstrap_batch(np.asarray(ct_in)[None], testvec)[0]

    # programmable_bootstrap.go:54-69,93-115 : the LUT is a TRLWE test vector [2][N]
    # (a lut.LookUpTable or its [2][N] array)
    def BootstrapLUT(self, ct_in, lut):
        return self.ctx.bootstrap_batch(np.asarray(ct_in)[None], getattr(lut, "poly", lut))[0]

    def BootstrapLUTAssign(self, ct_in, lut, ct_out):
        ct_out[...] = self.BootstrapLUT(ct_in, lut)

    # programmable_bootstrap.go:16-52 : build the table for f over [0, messageModulus), then bootstrap
    def BootstrapFunc(self, ct_in, f, messageModulus):
        from .lut import Generator
        return self.BootstrapLUT(ct_in, Generator(self.ctx.params, messageModulus).GenLookUpTable(f))

    def BootstrapFuncAssign(self, ct_in, f, messageModulus, ct_out):
        ct_out[...] = self.BootstrapFunc(ct_in, f, messageModulus)

    # Extended tables (LookUpTableSize = polyExtendFactor * N): what the Uint6/7/8 sets are specified for and the
    # reference leaves out (params.go:399-402, params/UINT_STATUS.md:12-30).  lut: [ext][2][N] from
    # lut.Generator(params, modulus, polyExtendFactor=ext).GenLookUpTableExtended(f).
    def BootstrapLUTExtended(self, ct_in, lut):
        return self.ctx.bootstrap_extended_batch(np.asarray(ct_in)[None], lut)[0]

    def BootstrapFuncExtended(self, ct_in, f, messageModulus, polyExtendFactor):
        from .lut import Generator
        gen = Generator(self.ctx.params, messageModulus, polyExtendFactor=polyExtendFactor)
        return self.BootstrapLUTExtended(ct_in, gen.GenLookUpTableExtended(f))

    def BatchBootstrapLUTExtended(self, cts, lut):
        return self.ctx.bootstrap_extended_batch(cts, lut)

    # one table, many ciphertexts in one launch (the batch form of BootstrapLUT)
    def BatchBootstrapLUT(self, cts, lut):
        return self.ctx.bootstrap_batch(cts, getattr(lut, "poly", lut))

    # batch forms (trgsw.go:234-252)
    def BatchBlindRotate(self, cts, testvec=None):
        return self.ctx.blind_rotate_batch(cts, testvec)

    def BatchBootstrap(self, cts, testvec=None):
        return self.ctx.bootstrap_batch(cts, testvec)

    # gates_helper.go:10-63 are fused into the kernel; these return the prepared sample for
    # callers that want the Prepare + Bootstrap seam.
    @staticmethod
    def _prep(a, b, sa, sb, cst):
        a = np.asarray(a, dtype=np.uint32)
        b = np.asarray(b, dtype=np.uint32)
        out = (np.uint32(sa) * a + np.uint32(sb) * b).astype(np.uint32)
        out[-1] = np.uint32((int(out[-1]) + cst) & 0xFFFFFFFF)
        return out

    def PrepareNAND(self, a, b): return self._prep(a, b, 0xFFFFFFFF, 0xFFFFFFFF, 0x20000000)
    def PrepareAND(self, a, b): return self._prep(a, b, 1, 1, 0xE0000000)
    def PrepareOR(self, a, b): return self._prep(a, b, 1, 1, 0x20000000)
    def PrepareXOR(self, a, b): return self._prep(a, b, 1, 2, 0x40000000)

"""The functions of the reference's trgsw package that lie on the bootstrap path (trgsw/trgsw.go:108-312, trgsw/keyswitch.go:10-37),
with its names and argument order, on a GPU-resident CloudKey: what gates.Batch* and the evaluator are themselves built on
(SURVEY.md 8(b) seam 3).  A TRGSW operand is [2L][2][N] float64 in the reference's FourierPoly layout and travels with the call; the
bootstrapping / key-switching keys are the ones resident in `ck` (the Go shim finds them by identity: shim/go/trgsw).  polyEval
arguments of the Go signatures have no counterpart here."""
import numpy as np


def ExternalProductWithFFT(trgswFFT, trlweIn, decompositionOffset, ck):
    """trgsw.go:108 : trgswFFT (x) trlweIn ([2][N])."""
    return ck.ctx.external_product_with(trgswFFT, np.asarray(trlweIn)[None], decompositionOffset)[0]


def CMUX(in1, in2, cond, decompositionOffset, ck):
    """trgsw.go:173 : in1 where cond encrypts 0, in2 where it encrypts 1."""
    return ck.ctx.cmux_with(cond, np.asarray(in1)[None], np.asarray(in2)[None], decompositionOffset)[0]


def _check_offset(ck, decompositionOffset):
    if int(decompositionOffset) != ck.ctx.decomposition_offset():
        raise ValueError("decompositionOffset is not the offset of the cloud key's parameters (cloudkey.go:60-71)")


def BlindRotate(src, blindRotateTestvec, decompositionOffset, ck):
    """trgsw.go:197 : all n CMUX steps of one sample."""
    _check_offset(ck, decompositionOffset)
    return ck.ctx.blind_rotate_batch(np.asarray(src)[None], blindRotateTestvec)[0]


def BatchBlindRotate(srcs, blindRotateTestvec, decompositionOffset, ck):
    """trgsw.go:234 : one goroutine per input there, one launch here."""
    _check_offset(ck, decompositionOffset)
    return ck.ctx.blind_rotate_batch(srcs, blindRotateTestvec)


def IdentityKeySwitching(src, ck):
    """trgsw.go:285 : an extracted level-1 sample ([N+1]) under the level-0 key."""
    return ck.ctx.keyswitch_batch(np.asarray(src)[None])[0]


def IdentityKeySwitchingAssign(src, ck, output):
    """trgsw/keyswitch.go:10."""
    output[...] = IdentityKeySwitching(src, ck)

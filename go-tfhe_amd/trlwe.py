"""trlwe.SampleExtractIndex[Assign] of the reference (trlwe/trlwe.go:114-128, trlwe/trlwe_ops.go:10-21) on the GPU, for any index k.
Inside a bootstrap the engine fuses the extraction at index 0 into the key switch; this is the seam on its own."""
import numpy as np


def SampleExtractIndex(trlwe, k, ck):
    """[2][N] -> the level-1 LWE sample of coefficient k, [N+1] with the body last."""
    return ck.ctx.sample_extract_batch(np.asarray(trlwe)[None], k)[0]


def SampleExtractIndexAssign(trlwe, k, ck, output):
    output[...] = SampleExtractIndex(trlwe, k, ck)

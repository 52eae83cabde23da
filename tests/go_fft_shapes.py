"""Test infrastructure: the reference's fftInPlace / ifftInPlace restated stage shape by stage shape.

oracle/tfhe_oracle.c collapses the four hand-unrolled stage shapes of poly/fourier_transform.go:178-247 (first stage /
middle stages / second-to-last / last, over blocks of eight doubles [4 re | 4 im], poly.go:54-62) into ONE generic loop
over complex slots, and the three shapes of the inverse (:258-347) likewise.  That is the one place where the oracle
RESTRUCTURES the reference instead of restating it, and the slot order tfhe_load_bsk_fourier relies on rests on it.  This
module keeps the reference's own structure -- the same loops over the same flat array of doubles, the same block offsets,
the same order in which the twiddle table is consumed -- in plain Python floats (IEEE double, no fused multiply-add: what
Go does on amd64), so that tests/test_oracle_pins.py can hold the oracle's loop to it BITWISE.

It pins the oracle's restructuring, not the reference: parity with the Go binary itself stays unpinned.
"""
import cmath
import math


def bit_reverse_in_place(data):
    """poly_evaluator.go:146-165"""
    n = len(data)
    if n <= 1:
        return
    j = 0
    for i in range(n):
        if i < j:
            data[i], data[j] = data[j], data[i]
        m = n >> 1
        while m > 0 and j >= m:
            j -= m
            m >>= 1
        j += m


def gen_twiddle_factors(N):
    """poly_evaluator.go:114-143 for a transform of N complex points (the evaluator passes ring degree / 2)."""
    tw_fft = [cmath.exp(complex(0, -2 * math.pi * i / N)) for i in range(N // 2)]
    tw_inv_fft = [cmath.exp(-complex(0, -2 * math.pi * i / N)) for i in range(N // 2)]
    bit_reverse_in_place(tw_fft)
    bit_reverse_in_place(tw_inv_fft)
    tw, tw_inv = [], []
    m, t = 1, N // 2
    while m <= N // 2:
        fold = cmath.exp(complex(0, 2 * math.pi * t / (4 * N)))
        tw.extend(tw_fft[i] * fold for i in range(m))
        m, t = m << 1, t >> 1
    m, t = N // 2, 1
    while m >= 1:
        fold = cmath.exp(complex(0, -2 * math.pi * t / (4 * N)))
        tw_inv.extend(tw_inv_fft[i] * fold for i in range(m))
        m, t = m >> 1, t << 1
    return tw, tw_inv


def _butterfly(uR, uI, vR, vI, wR, wI):
    """fourier_transform.go:170-174"""
    vwR = vR * wR - vI * wI
    vwI = vR * wI + vI * wR
    return uR + vwR, uI + vwI, uR - vwR, uI - vwI


def _inv_butterfly(uR, uI, vR, vI, wR, wI):
    """fourier_transform.go:250-255"""
    uR, uI, vR, vI = uR + vR, uI + vI, uR - vR, uI - vI
    return uR, uI, vR * wR - vI * wI, vR * wI + vI * wR


def fold(p):
    """convertPolyToFourierPolyAssign (fourier_transform.go:64-85): p[ii..ii+3] -> re, p[ii+N/2..] -> im, by blocks of 8."""
    N = len(p)
    out = [0.0] * N
    s32 = lambda x: float(x - (1 << 32) if x >= (1 << 31) else x)
    i = ii = 0
    while i < N:
        for k in range(4):
            out[i + k] = s32(int(p[ii + k]))
            out[i + 4 + k] = s32(int(p[ii + N // 2 + k]))
        i, ii = i + 8, ii + 4
    return out


def fft_in_place(c, tw):
    """fftInPlace (fourier_transform.go:178-247) on the flat array of N doubles."""
    N = len(c)
    w = 0
    # first stage: one twiddle, u block j, v block j + N/2
    wR, wI = tw[w].real, tw[w].imag
    w += 1
    for j in range(0, N // 2, 8):
        u, v = j, j + N // 2
        for k in range(4):
            c[u + k], c[u + 4 + k], c[v + k], c[v + 4 + k] = _butterfly(c[u + k], c[u + 4 + k], c[v + k], c[v + 4 + k], wR, wI)
    # middle stages
    t = N // 2
    m = 2
    while m <= N // 16:
        t >>= 1
        for i in range(m):
            j1 = 2 * i * t
            j2 = j1 + t
            wR, wI = tw[w].real, tw[w].imag
            w += 1
            for j in range(j1, j2, 8):
                u, v = j, j + t
                for k in range(4):
                    c[u + k], c[u + 4 + k], c[v + k], c[v + 4 + k] = _butterfly(c[u + k], c[u + 4 + k], c[v + k], c[v + 4 + k], wR, wI)
        m <<= 1
    # second-to-last stage: inside one block, slots (0, 2) and (1, 3), one twiddle per block
    for j in range(0, N, 8):
        wR, wI = tw[w].real, tw[w].imag
        w += 1
        r, im = j, j + 4
        c[r + 0], c[im + 0], c[r + 2], c[im + 2] = _butterfly(c[r + 0], c[im + 0], c[r + 2], c[im + 2], wR, wI)
        c[r + 1], c[im + 1], c[r + 3], c[im + 3] = _butterfly(c[r + 1], c[im + 1], c[r + 3], c[im + 3], wR, wI)
    # last stage: slots (0, 1) and (2, 3), two twiddles per block
    for j in range(0, N, 8):
        wR0, wI0, wR1, wI1 = tw[w].real, tw[w].imag, tw[w + 1].real, tw[w + 1].imag
        w += 2
        r, im = j, j + 4
        c[r + 0], c[im + 0], c[r + 1], c[im + 1] = _butterfly(c[r + 0], c[im + 0], c[r + 1], c[im + 1], wR0, wI0)
        c[r + 2], c[im + 2], c[r + 3], c[im + 3] = _butterfly(c[r + 2], c[im + 2], c[r + 3], c[im + 3], wR1, wI1)
    assert w == len(tw)


def ifft_in_place(c, tw_inv):
    """ifftInPlace (fourier_transform.go:258-347) including the division by N/2 in its last stage."""
    N = len(c)
    w = 0
    for j in range(0, N, 8):                                   # first stage (reverse of the last forward stage)
        wR0, wI0, wR1, wI1 = tw_inv[w].real, tw_inv[w].imag, tw_inv[w + 1].real, tw_inv[w + 1].imag
        w += 2
        r, im = j, j + 4
        c[r + 0], c[im + 0], c[r + 1], c[im + 1] = _inv_butterfly(c[r + 0], c[im + 0], c[r + 1], c[im + 1], wR0, wI0)
        c[r + 2], c[im + 2], c[r + 3], c[im + 3] = _inv_butterfly(c[r + 2], c[im + 2], c[r + 3], c[im + 3], wR1, wI1)
    for j in range(0, N, 8):                                   # second stage
        wR, wI = tw_inv[w].real, tw_inv[w].imag
        w += 1
        r, im = j, j + 4
        c[r + 0], c[im + 0], c[r + 2], c[im + 2] = _inv_butterfly(c[r + 0], c[im + 0], c[r + 2], c[im + 2], wR, wI)
        c[r + 1], c[im + 1], c[r + 3], c[im + 3] = _inv_butterfly(c[r + 1], c[im + 1], c[r + 3], c[im + 3], wR, wI)
    t = 8                                                      # middle stages
    m = N // 16
    while m >= 2:
        for i in range(m):
            j1 = 2 * i * t
            j2 = j1 + t
            wR, wI = tw_inv[w].real, tw_inv[w].imag
            w += 1
            for j in range(j1, j2, 8):
                u, v = j, j + t
                for k in range(4):
                    c[u + k], c[u + 4 + k], c[v + k], c[v + 4 + k] = _inv_butterfly(c[u + k], c[u + 4 + k], c[v + k], c[v + 4 + k], wR, wI)
        t <<= 1
        m >>= 1
    scale = float(N // 2)                                      # last stage with scaling
    wR, wI = tw_inv[w].real, tw_inv[w].imag
    w += 1
    for j in range(0, N // 2, 8):
        u, v = j, j + N // 2
        for k in range(4):
            c[u + k], c[u + 4 + k], c[v + k], c[v + 4 + k] = _inv_butterfly(c[u + k], c[u + 4 + k], c[v + k], c[v + 4 + k], wR, wI)
        for k in range(8):
            c[u + k] /= scale
            c[v + k] /= scale
    assert w == len(tw_inv)

"""GPU parity: wave-native negacyclic FFT vs the oracle's restatement of
poly/fourier_transform.go (ToFourierPolyAssign / ToPolyAssignUnsafe)."""
import numpy as np
import pytest

from conftest import rand_u32

pytestmark = pytest.mark.gpu


def test_to_fourier_matches_reference_layout(oracle, ck_small):
    rs = np.random.RandomState(1)
    polys = rand_u32(rs, (5, 1024))
    polys[0] = 0
    polys[1] = 0; polys[1][0] = 1            # constant 1 -> all-ones spectrum
    polys[2] = 0; polys[2][1] = 1            # X        -> the evaluation points themselves
    got = ck_small.ctx.to_fourier_batch(polys)
    for k in range(polys.shape[0]):
        want = oracle.to_fourier(polys[k])
        scale = max(1.0, np.abs(want).max())
        # fp64 tolerance: both are O(log N)-ulp accurate transforms of values < 2^31 * N
        assert np.abs(got[k] - want).max() <= 1e-11 * scale, k


def test_round_trip_exact(oracle, ck_small):
    # reference bound is diff <= 10 (poly/poly_test.go:10-33); we get 0 on 32-bit inputs
    rs = np.random.RandomState(2)
    polys = rand_u32(rs, (16, 1024))
    back = ck_small.ctx.to_poly_batch(ck_small.ctx.to_fourier_batch(polys))
    assert np.array_equal(back, polys)


def test_inverse_of_reference_spectrum(oracle, ck_small):
    # spectra produced by the ORACLE's forward transform invert exactly on the GPU
    rs = np.random.RandomState(3)
    polys = rand_u32(rs, (4, 1024))
    spectra = np.stack([oracle.to_fourier(p) for p in polys])
    assert np.array_equal(ck_small.ctx.to_poly_batch(spectra), polys)


def test_fft_product_equals_exact_integer_product(oracle, ck_small):
    # digits (6-bit) x torus polynomial through the GPU transforms == schoolbook mod 2^32
    rs = np.random.RandomState(4)
    dig = (rs.randint(-32, 32, size=(3, 1024)).astype(np.int32)).view(np.uint32)
    key = rand_u32(rs, (3, 1024))
    fd = ck_small.ctx.to_fourier_batch(dig)
    fk = ck_small.ctx.to_fourier_batch(key)

    def cmul(a, b):           # reference FourierPoly blocks: [4 re | 4 im]
        a = a.reshape(-1, 2, 4); b = b.reshape(-1, 2, 4)
        out = np.empty_like(a)
        out[:, 0] = a[:, 0] * b[:, 0] - a[:, 1] * b[:, 1]
        out[:, 1] = a[:, 0] * b[:, 1] + a[:, 1] * b[:, 0]
        return out.reshape(-1)

    prod = np.stack([cmul(fd[i], fk[i]) for i in range(3)])
    got = ck_small.ctx.to_poly_batch(prod)
    for i in range(3):
        assert np.array_equal(got[i], oracle.negacyclic_exact(dig[i], key[i]))

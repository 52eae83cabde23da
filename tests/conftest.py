import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    graft.build()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    from oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def pkg(built):
    return graft.load_package()


class KeySet:
    """Seeded secret + cloud key for one parameter set (made by the oracle harness)."""

    def __init__(self, o, name, seed, n_override=None, torus=True):
        self.o = o
        self.name = name
        p = o.params(name)
        if n_override:
            p = p.small(n_override)
        self.p = p
        self.rng = o.rng(seed)
        self.s0, self.s1 = o.keygen_secret(p, self.rng)
        self.bsk_torus, self.bsk = o.keygen_bsk(p, self.rng, self.s0, self.s1, torus=torus, fourier=True)
        self.ksk = o.keygen_ksk(p, self.rng, self.s0, self.s1)
        self.tv = o.gate_testvec(p)

    def enc(self, bits):
        return self.o.encrypt_bools(self.p, self.rng, bits, self.s0)

    def dec(self, cts):
        return self.o.decrypt_bools(self.p, self.s0, cts)


@pytest.fixture(scope="session")
def keys80(oracle):
    return KeySet(oracle, "80", 0x7F4E0001)


@pytest.fixture(scope="session")
def keys128(oracle):
    return KeySet(oracle, "128", 0x7F4E0002)


@pytest.fixture(scope="session")
def keys_small(oracle):
    """128-bit ring/gadget with a short LWE dimension: full pipeline in milliseconds.
    (Not a secure set; bit-parity does not depend on n.)"""
    return KeySet(oracle, "128", 0x7F4E0003, n_override=24)


def gpu_params(pkg, p):
    return pkg.Params(n=p.n, N=p.N, Nbit=p.Nbit, L=p.L, Bgbit=p.Bgbit, basebit=p.basebit, t=p.t)


@pytest.fixture(scope="session")
def ck80(pkg, keys80):
    ck = pkg.CloudKey(gpu_params(pkg, keys80.p), bsk_fourier=keys80.bsk, ksk=keys80.ksk)
    yield ck
    ck.close()


@pytest.fixture(scope="session")
def ck128(pkg, keys128):
    ck = pkg.CloudKey(gpu_params(pkg, keys128.p), bsk_fourier=keys128.bsk, ksk=keys128.ksk)
    yield ck
    ck.close()


@pytest.fixture(scope="session")
def ck_small(pkg, keys_small):
    ck = pkg.CloudKey(gpu_params(pkg, keys_small.p), bsk_fourier=keys_small.bsk, ksk=keys_small.ksk)
    yield ck
    ck.close()


def rand_u32(rs, shape):
    return rs.randint(0, 2**32, size=shape, dtype=np.uint64).astype(np.uint32)

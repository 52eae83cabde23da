"""cloudkey.CloudKey (cloudkey/cloudkey.go:16-31) as a GPU-resident object.

The reference's CloudKey holds {DecompositionOffset, BlindRotateTestvec, KeySwitchingKey,
BootstrappingKey} as Go pointer graphs.  Here the same four things live on one GPU inside a
Context: offset and gate test vector are derived from the parameters at context creation; the
two keys are either uploaded once from flat arrays (what a cgo shim would flatten them to) or
generated on the GPU from the two binary secret keys (NewCloudKey below).
"""
from ._binding import Context


class CloudKey:
    def __init__(self, params, bsk_fourier=None, bsk_torus=None, ksk=None, device=0):
        self.params = params
        self.ctx = Context(params, device)
        if bsk_fourier is not None:
            self.ctx.load_bsk_fourier(bsk_fourier)
        elif bsk_torus is not None:
            self.ctx.load_bsk_torus(bsk_torus)
        if ksk is not None:
            self.ctx.load_ksk(ksk)

    def close(self):
        self.ctx.close()

    @classmethod
    def NewCloudKey(cls, params, key_lv0, key_lv1, alpha_lv0, alpha_lv1, seed=None, device=0):
        """cloudkey.NewCloudKey(secretKey) (cloudkey.go:24-31) generated ON the GPU
        (tfhe_keygen_cloud_seeded): nothing but the two binary secret keys crosses PCIe.
        seed=None draws 128 bits from the OS entropy source, like the reference's auto-seeded generator; a fixed
        integer seed makes the cloud key reproducible and is for tests only -- the seed is secret key material
        (every mask and noise sample of the published key follows from it)."""
        ck = cls(params, device=device)
        ck.ctx.keygen_cloud(key_lv0, key_lv1, alpha_lv0, alpha_lv1, seed)
        return ck

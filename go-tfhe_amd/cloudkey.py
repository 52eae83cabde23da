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

    def clone_to(self, device):
        """A replica of this cloud key on GPU `device`, copied GPU to GPU behind the C ABI (tfhe_ctx_clone_to)."""
        other = CloudKey.__new__(CloudKey)
        other.params = self.params
        other.ctx = self.ctx.clone_to(device)
        return other


class CloudKeySet:
    """One cloud key on several GPUs of a node, used from ONE process: the in-process form of "replicate the read-only
    keys, shard the batch" (trgsw.BatchBlindRotate, trgsw.go:234-252, fans a batch out over goroutines sharing the keys).
    Every replica is made by tfhe_ctx_clone_to, i.e. GPU to GPU (over xGMI between two GPUs); `src` stays the caller's.
    `devices` may repeat an index (two contexts on one GPU are two independent submitters: that is how a
    one-GPU box exercises this).  Mirrors shim/go/gpu.CloudKeySet and host/tfhe_gpu.hpp cloudkey::CloudKeySet."""

    def __init__(self, src, devices):
        if not devices:
            raise ValueError("CloudKeySet needs at least one device")
        self.params = src.params
        self.replicas = [src.clone_to(d) for d in devices]

    def __len__(self):
        return len(self.replicas)

    def __getitem__(self, i):
        return self.replicas[i]

    def shards(self, B):
        """Contiguous index ranges [g*B/G, (g+1)*B/G)."""
        G = len(self.replicas)
        return [(g * B // G, (g + 1) * B // G) for g in range(G)]

    def gate_batch(self, ops, a, b, c=None):
        """gates.Batch* over all replicas: contiguous shards, one thread per replica, results in index order."""
        import threading
        import numpy as np
        a = np.ascontiguousarray(a, dtype=np.uint32)
        b = np.ascontiguousarray(b, dtype=np.uint32)
        out = np.empty_like(a)
        errs = []

        def run(g, lo, hi):
            try:
                o = ops if isinstance(ops, (str, int)) else ops[lo:hi]
                out[lo:hi] = self.replicas[g].ctx.gate_batch(o, a[lo:hi], b[lo:hi], None if c is None else c[lo:hi])
            except Exception as e:      # noqa: BLE001 -- re-raised on the calling thread
                errs.append(e)

        ts = [threading.Thread(target=run, args=(g, lo, hi)) for g, (lo, hi) in enumerate(self.shards(len(a))) if hi > lo]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise errs[0]
        return out

    def close(self):
        for r in self.replicas:
            r.close()

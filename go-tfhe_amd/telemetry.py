"""Shader clock and socket power of ONE GPU, sampled by a background thread while a measurement runs.

The blind rotate runs power-limited on an MI355X (~1.36 kW of the 1.4 kW board power, shader clock ~2.26 GHz instead of
2.4: profiles/r03_p_clock_power.txt), so a throughput figure -- and above all a multi-GPU scaling curve, where eight
boards share one chassis -- has to carry the clock and the power it was measured at, per rank.  This binds the few
librocm_smi64 entry points needed directly (ctypes; the `rocm-smi` command takes ~0.5 s per reading, too coarse for a
0.1 s timed region) and matches the SMI device to the HIP device by PCI address, not by index.

Measurement infrastructure only: nothing on the hot path depends on it, and every failure (library missing, no
permission inside a container, unknown device) degrades to `{"available": False, "error": ...}`.
"""
import ctypes
import threading
import time

_RSMI_MAX_FREQ = 33
_RSMI_CLK_TYPE_SYS = 0


class _Freqs(ctypes.Structure):
    _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32), ("current", ctypes.c_uint32),
                ("frequency", ctypes.c_uint64 * _RSMI_MAX_FREQ)]


_lib = None
_lib_err = None
_lock = threading.Lock()


def _load():
    global _lib, _lib_err
    with _lock:
        if _lib is not None or _lib_err is not None:
            return _lib
        try:
            lib = ctypes.CDLL("librocm_smi64.so")
            rc = lib.rsmi_init(ctypes.c_uint64(0))
            if rc != 0:
                raise OSError(f"rsmi_init returned {rc}")
            _lib = lib
        except Exception as e:                      # noqa: BLE001 -- any failure means "no telemetry", never a crash
            _lib_err = f"{type(e).__name__}: {e}"
        return _lib


def smi_index_for_pci(domain, bus, device):
    """Index of the SMI device with this PCI address (rsmi_dev_pci_id_get: domain << 32 | bus << 8 | device << 3 | fn), or None."""
    lib = _load()
    if lib is None:
        return None
    n = ctypes.c_uint32(0)
    if lib.rsmi_num_monitor_devices(ctypes.byref(n)) != 0:
        return None
    for i in range(n.value):
        bdf = ctypes.c_uint64(0)
        if lib.rsmi_dev_pci_id_get(ctypes.c_uint32(i), ctypes.byref(bdf)) != 0:
            continue
        v = bdf.value
        if ((v >> 32) & 0xFFFFFFFF, (v >> 8) & 0xFF, (v >> 3) & 0x1F) == (domain, bus, device):
            return i
    return None


def smi_index_for_torch_device(index):
    """SMI index of torch's cuda:<index> (by PCI address; falls back to the same index when the address is unknown)."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(index)
        i = smi_index_for_pci(int(pr.pci_domain_id), int(pr.pci_bus_id), int(pr.pci_device_id))
        if i is not None:
            return i
    except Exception:                               # noqa: BLE001
        pass
    return index if _load() is not None else None


def read_once(smi_index):
    """(sclk MHz or None, socket power W or None) right now."""
    lib = _load()
    if lib is None or smi_index is None:
        return None, None
    mhz = watts = None
    f = _Freqs()
    if lib.rsmi_dev_gpu_clk_freq_get(ctypes.c_uint32(smi_index), ctypes.c_int(_RSMI_CLK_TYPE_SYS), ctypes.byref(f)) == 0 \
            and f.current < min(f.num_supported, _RSMI_MAX_FREQ):
        mhz = f.frequency[f.current] / 1e6
    p = ctypes.c_uint64(0)
    if lib.rsmi_dev_current_socket_power_get(ctypes.c_uint32(smi_index), ctypes.byref(p)) == 0:
        watts = p.value / 1e6
    else:
        if lib.rsmi_dev_power_ave_get(ctypes.c_uint32(smi_index), ctypes.c_uint32(0), ctypes.byref(p)) == 0:
            watts = p.value / 1e6
    return mhz, watts


def summarize(samples):
    """min / mean / max of the (sclk, power) samples taken while a region ran."""
    clk = [s[0] for s in samples if s[0] is not None]
    pw = [s[1] for s in samples if s[1] is not None]
    out = {"available": bool(clk or pw), "samples": len(samples)}
    if clk:
        out.update(sclk_mhz_min=min(clk), sclk_mhz_mean=sum(clk) / len(clk), sclk_mhz_max=max(clk))
    if pw:
        out.update(power_w_min=min(pw), power_w_mean=sum(pw) / len(pw), power_w_max=max(pw))
    return out


class Sampler:
    """`with Sampler(device_index) as s: <timed region>` then `s.summary()`.

    One reading is taken at entry and one at exit whatever the region's length, the rest every `period_s`."""

    def __init__(self, torch_device_index=0, period_s=0.01):
        self.idx = smi_index_for_torch_device(torch_device_index)
        self.period = period_s
        self.samples = []
        self._stop = threading.Event()
        self._thr = None

    def _run(self):
        while not self._stop.is_set():
            self.samples.append(read_once(self.idx))
            self._stop.wait(self.period)

    def __enter__(self):
        if self.idx is not None:
            self._thr = threading.Thread(target=self._run, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *exc):
        if self._thr is not None:
            self._stop.set()
            self._thr.join(timeout=2.0)
            self.samples.append(read_once(self.idx))

    def summary(self):
        if self.idx is None:
            return {"available": False, "error": _lib_err or "no SMI device for this GPU"}
        s = summarize(self.samples)
        s["smi_index"] = self.idx
        s["period_ms"] = self.period * 1e3
        return s


if __name__ == "__main__":                          # python go-tfhe_amd/telemetry.py : one reading per device
    lib = _load()
    print("librocm_smi64:", "ok" if lib is not None else _lib_err)
    if lib is not None:
        n = ctypes.c_uint32(0)
        lib.rsmi_num_monitor_devices(ctypes.byref(n))
        for i in range(n.value):
            t0 = time.perf_counter()
            r = read_once(i)
            print(i, r, f"{(time.perf_counter() - t0) * 1e3:.2f} ms per reading")

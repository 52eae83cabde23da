"""Runs tools/ubench_mfma_f64.bin one configuration at a time, each long enough to reach the power-limited clock, with the
shader clock and socket power sampled meanwhile (go-tfhe_amd/telemetry.py), and prints the verdict of VERDICT r04 item 4a:
does one VALU wave + one MFMA wave per SIMD deliver >= 1.25 x the DFT-8 throughput of two VALU waves?

usage: python tools/run_ubench_mfma_f64.py [seconds per configuration, default 6]"""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("telemetry", os.path.join(ROOT, "go-tfhe_amd", "telemetry.py"))
tel = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tel)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    exe = os.path.join(ROOT, "tools", "ubench_mfma_f64.bin")
    idx = 0 if tel._load() is not None else None
    rows = {}
    for cfg in ("V2", "VM", "V1", "M1", "M2", "VVM", "V2"):             # V2 twice: first and last, to bracket drift
        s = tel.Sampler.__new__(tel.Sampler)
        s.idx, s.period, s.samples, s._stop, s._thr = idx, 0.05, [], tel.threading.Event(), None
        with s:
            r = subprocess.run([exe, cfg, str(seconds)], capture_output=True, text=True, timeout=120)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(cfg, "FAILED", r.stdout[-300:], r.stderr[-300:])
            continue
        rec = json.loads(line[-1])
        # the calibration pass runs first; the clock / power figures cover both passes (the timed one is >= 2/3 of the window)
        rec["telemetry"] = s.summary()
        key = cfg if cfg not in rows else cfg + "_again"
        rows[key] = rec
        t = rec["telemetry"]
        print(f"{key:9s} {rec['total_sets_per_us_per_cu']:7.3f} DFT-8 sets/us/CU  (VALU {rec['valu_sets_per_us_per_cu']:.3f} + MFMA {rec['mfma_sets_per_us_per_cu']:.3f})"
              f"  {t.get('sclk_mhz_mean', float('nan')):6.0f} MHz  {t.get('power_w_mean', float('nan')):6.0f} W   placement {rec['placement_valu_mfma_per_simd']}"
              f"  MFMA {rec['mfma_tflops_chip']:.1f} Tflop/s")
    if "V2" in rows and "VM" in rows:
        base = max(rows["V2"]["total_sets_per_us_per_cu"], rows.get("V2_again", rows["V2"])["total_sets_per_us_per_cu"])
        ratio = rows["VM"]["total_sets_per_us_per_cu"] / base
        print(f"VM / V2 = {ratio:.3f}  -> {'ADOPT candidate (>= 1.25)' if ratio >= 1.25 else 'KILLED (< 1.25): the DFT stays on the VALU'}")
        if "VVM" in rows:
            print(f"VVM / V2 = {rows['VVM']['total_sets_per_us_per_cu'] / base:.3f}  (three waves per SIMD: not reachable at the kernels' 256 VGPRs)")
    print(json.dumps(rows))


if __name__ == "__main__":
    main()

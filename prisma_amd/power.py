"""Socket power / shader clock sampling beside a timed region (bench.py's `avg_power_w`, `joules_per_frame`; tools/overlap_bench.py).

Reads the amdgpu hwmon files of the device (`power1_average` / `power1_input` in microwatts, `freq1_input` in Hz) from a sampler thread; when
the box exposes no such files, falls back to polling `amd-smi metric --power --clock --json` (a few samples per second).  A diagnostic: every
method returns None / empty results when nothing is readable, and the callers then report null."""
import glob
import json
import os
import subprocess
import threading
import time


def _pci_bdf(device: int = 0):
    """PCI address ("0000:05:00.0") of HIP device `device` of this process, or None.  The box shows every GPU of the host under /sys while
    the process may only see one of them, so the hwmon directory has to be found through the bus address, not by card index."""
    try:
        import ctypes
        hip = None
        for name in ("libamdhip64.so.7", "libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):      # the copy already mapped (torch's or the band library's) first
            try:
                hip = ctypes.CDLL(name)
                break
            except OSError:
                continue
        buf = ctypes.create_string_buffer(64)
        if hip is None or hip.hipDeviceGetPCIBusId(buf, 64, int(device)) != 0:
            return None
        return buf.value.decode().lower()
    except Exception:      # noqa: BLE001 - diagnostic only
        return None


def _hwmon_files(device: int = 0):
    bdf = _pci_bdf(device)
    if bdf:
        cards = sorted(glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*"))
        device = 0
    else:
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"))
    out = []
    for h in cards:
        p = next((os.path.join(h, n) for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(h, n))), None)
        f = os.path.join(h, "freq1_input") if os.path.exists(os.path.join(h, "freq1_input")) else None
        if p:
            out.append((p, f))
    return out[device] if device < len(out) else None


def _read_num(path):
    try:
        with open(path) as fh:
            return float(fh.read().strip())
    except (OSError, ValueError):
        return None


def _smi_sample():
    """(watts, MHz) from amd-smi's JSON, or (None, None)"""
    try:
        r = subprocess.run(["/opt/rocm/bin/amd-smi", "metric", "--power", "--clock", "--json"], capture_output=True, text=True, timeout=5)
        d = json.loads(r.stdout)
        d = d[0] if isinstance(d, list) else d
        if "gpu_data" in d:
            d = d["gpu_data"][0]
        pw = d.get("power", {}).get("socket_power", {})
        w = float(pw.get("value")) if isinstance(pw, dict) else float(pw)
        clk = d.get("clock", {})
        mhz = [float(v["clk"]["value"]) for k, v in clk.items() if k.startswith("gfx") and isinstance(v, dict) and isinstance(v.get("clk"), dict)]
        return w, (sum(mhz) / len(mhz) if mhz else None)
    except Exception:      # noqa: BLE001 - diagnostic only
        return None, None


class PowerSampler:
    """with PowerSampler() as ps: ...; ps.window(t0, t1) -> {"avg_power_w", "avg_sclk_mhz", "samples", "source"}"""

    def __init__(self, device: int = 0, interval_s: float = 0.01):
        self.files = _hwmon_files(device)
        self.interval = interval_s if self.files else 0.25
        self.samples = []          # (perf_counter, watts, mhz)
        self._stop = threading.Event()
        self._th = None
        self.source = "hwmon" if self.files else "amd-smi"

    def _run(self):
        while not self._stop.is_set():
            t = time.perf_counter()
            if self.files:
                uw = _read_num(self.files[0])
                hz = _read_num(self.files[1]) if self.files[1] else None
                w, mhz = (uw / 1e6 if uw is not None else None), (hz / 1e6 if hz is not None else None)
            else:
                w, mhz = _smi_sample()
            if w is not None:
                self.samples.append((t, w, mhz))
            self._stop.wait(self.interval)

    def __enter__(self):
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._th:
            self._th.join(timeout=10)
        return False

    def window(self, t0: float, t1: float):
        s = [x for x in self.samples if t0 <= x[0] <= t1]
        if not s:
            return {"avg_power_w": None, "avg_sclk_mhz": None, "samples": 0, "source": self.source}
        clk = [x[2] for x in s if x[2] is not None]
        return {"avg_power_w": round(sum(x[1] for x in s) / len(s), 1), "avg_sclk_mhz": round(sum(clk) / len(clk), 1) if clk else None,
                "samples": len(s), "source": self.source}

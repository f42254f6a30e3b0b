"""Socket power and shader clock of ONE GPU, sampled on the box the code runs on.

The amdgpu driver publishes both per device under sysfs: hwmon `power1_input` (PPT, microwatts; `power1_average` on older
drivers) and `freq1_input` (sclk, Hz).  A read costs 10-200 microseconds (tools/probe_power.py), so a thread can sample at
~1 kHz beside a measurement without a subprocess.  The device is found through its PCI bus id (dm_device_pci_bus_id: the
HIP device index says nothing about the sysfs card number on a box that shows only one of its eight GPUs to the container).
No reference counterpart: this is measurement plumbing for bench.py's `roofline.power`."""
from __future__ import annotations

import glob
import os
import threading
import time
from typing import Optional


def hwmon_dir(pci_bus_id: str) -> Optional[str]:
    """hwmon directory of the amdgpu device with this PCI bus id ("0000:05:00.0"), or None."""
    for base in ("/sys/bus/pci/devices/%s" % pci_bus_id.lower(), "/sys/bus/pci/devices/%s" % pci_bus_id.upper()):
        for h in sorted(glob.glob(os.path.join(base, "hwmon", "hwmon*"))):
            if os.path.exists(os.path.join(h, "power1_input")) or os.path.exists(os.path.join(h, "power1_average")):
                return h
    return None


def _read_int(path: str) -> Optional[int]:
    try:
        with open(path) as fh:
            return int(fh.read().strip())
    except (OSError, ValueError):
        return None


class PowerLog:
    """with PowerLog(bus_id) as log: ...; log.summary() -> {"socket_power_w": {...}, "sclk_mhz": {...}, ...}"""

    def __init__(self, pci_bus_id: Optional[str], period_s: float = 0.001):
        self.bus_id = pci_bus_id
        self.dir = hwmon_dir(pci_bus_id) if pci_bus_id else None
        self.period = period_s
        self.samples = []          # (t, watts, mhz)
        self._stop = threading.Event()
        self._thread = None
        self.power_file = self.freq_file = None
        if self.dir:
            for f in ("power1_input", "power1_average"):
                if os.path.exists(os.path.join(self.dir, f)):
                    self.power_file = os.path.join(self.dir, f)
                    break
            if os.path.exists(os.path.join(self.dir, "freq1_input")):
                self.freq_file = os.path.join(self.dir, "freq1_input")

    @property
    def available(self) -> bool:
        return self.power_file is not None

    def cap_w(self) -> Optional[float]:
        v = _read_int(os.path.join(self.dir, "power1_cap")) if self.dir else None
        return v / 1e6 if v else None

    def _run(self):
        while not self._stop.is_set():
            t = time.perf_counter()
            p = _read_int(self.power_file)
            f = _read_int(self.freq_file) if self.freq_file else None
            if p is not None:
                self.samples.append((t, p / 1e6, f / 1e6 if f is not None else None))
            self._stop.wait(self.period)

    def start(self):
        if self.available and self._thread is None:
            self._stop.clear()
            self._thread = threading.Thread(target=self._run, name="powerlog", daemon=True)
            self._thread.start()
        return self

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join()
            self._thread = None
        return self

    __enter__ = start

    def __exit__(self, *exc):
        self.stop()

    def summary(self, t0: Optional[float] = None, t1: Optional[float] = None, skip_s: float = 0.0) -> dict:
        """Median / min / max over the samples in [t0 + skip_s, t1] (perf_counter times; default: all)."""
        if not self.available:
            return {"available": False, "why": "no amdgpu hwmon power file for PCI device %r" % (self.bus_id,)}
        s = [x for x in self.samples if (t0 is None or x[0] >= t0 + skip_s) and (t1 is None or x[0] <= t1)]
        out = {"available": True, "source": self.power_file, "pci_bus_id": self.bus_id, "samples": len(s), "socket_power_cap_w": self.cap_w(),
               "sclk_mhz_max": 2400, "window_s": (s[-1][0] - s[0][0]) if len(s) > 1 else 0.0}
        if s:
            med = lambda v: sorted(v)[len(v) // 2]
            pw = [x[1] for x in s]
            out["socket_power_w"] = {"median": med(pw), "min": min(pw), "max": max(pw)}
            ck = [x[2] for x in s if x[2] is not None]
            if ck:
                out["sclk_mhz"] = {"median": med(ck), "min": min(ck), "max": max(ck)}
        return out

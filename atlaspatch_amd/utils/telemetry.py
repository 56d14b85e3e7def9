"""Measurement aids for bench.py: what clock and power the chip ran at during a timed region.

MI355X runs every MFMA-heavy kernel at its package power cap (1 400 W), so the shader clock the governor grants -- not the
code -- explains run-to-run and box-to-box differences of a few percent.  Two independent readings:

* ``ClockProbe``: in-stream stamps of ``s_memtime`` (shader-clock ticks) and ``s_memrealtime`` (100 MHz) per XCD through
  ``ap_clock_probe`` before and after the region -> the AVERAGE shader clock over exactly the timed kernels.
* ``PowerSampler``: a host thread polling the amdgpu hwmon files of the device (package power in microwatts, sclk in Hz)
  while the region runs; ``amdsmi`` when sysfs is not visible.  Coarse (one sample per few milliseconds), independent.

No reference counterpart (the reference has no measurement code).
"""
from __future__ import annotations

import glob
import os
import threading
import time

import torch

from .. import _lib


class ClockProbe:
    SLOTS = 2048            # AP_CLOCK_PROBE_SLOTS: xcc << 8 | se << 5 | sh << 4 | cu

    def __init__(self, device) -> None:
        self.device = torch.device(device)
        self.lib = _lib.load()
        self.a = torch.zeros(2 * self.SLOTS, dtype=torch.int64, device=self.device)
        self.b = torch.zeros(2 * self.SLOTS, dtype=torch.int64, device=self.device)

    def _stamp(self, buf) -> None:
        with torch.cuda.device(self.device):
            _lib.check(self.lib.ap_clock_probe(buf.data_ptr(), _lib.current_stream_ptr(self.device)), "ap_clock_probe")

    def start(self) -> None:
        self.a.zero_(); self.b.zero_()
        self._stamp(self.a)

    def stop(self) -> None:
        self._stamp(self.b)

    def read(self) -> dict:
        """Call after a synchronize.  Per compute unit that both probes reached: GHz = 0.1 * d(memtime) / d(memrealtime) (the
        s_memtime counters of different compute units are not aligned, so only same-unit stamps are differenced); per XCD the
        median over its units, overall the median over the XCDs."""
        import numpy as np
        a = self.a.cpu().numpy().reshape(self.SLOTS, 2); b = self.b.cpu().numpy().reshape(self.SLOTS, 2)
        ok = (a[:, 1] > 0) & (b[:, 1] > a[:, 1]) & (b[:, 0] > a[:, 0])
        if not ok.any():
            return {"shader_clock_GHz": None, "xcds": 0}
        ghz = np.where(ok, 0.1 * (b[:, 0] - a[:, 0]) / np.maximum(b[:, 1] - a[:, 1], 1), np.nan)
        per = {}
        for x in range(8):
            v = ghz[x * 256:(x + 1) * 256]
            v = v[~np.isnan(v)]
            if v.size:
                per[x] = float(np.median(v))
        vals = sorted(per.values())
        allv = ghz[~np.isnan(ghz)]
        return {"shader_clock_GHz": round(vals[len(vals) // 2], 4), "min_GHz": round(vals[0], 4), "max_GHz": round(vals[-1], 4),
                "mean_GHz": round(sum(vals) / len(vals), 4), "per_xcd_GHz": [round(per[x], 4) if x in per else None for x in range(8)],
                "xcds": len(vals), "compute_units": int(ok.sum()),
                "cu_spread_GHz": [round(float(np.percentile(allv, 5)), 4), round(float(np.percentile(allv, 95)), 4)],
                "region_ms": round(float((b[ok, 1] - a[ok, 1]).max()) / 1e5, 3),
                "method": "s_memtime / s_memrealtime stamps per compute unit (ap_clock_probe) on the launch stream before the first "
                          "and after the last timed step; per XCD the median over its compute units, then the median over XCDs"}


def _device_bdf(device):
    try:
        p = torch.cuda.get_device_properties(device)
        return f"{int(p.pci_domain_id):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}.0"
    except (AttributeError, RuntimeError, AssertionError, TypeError, ValueError):
        return None


def hwmon_files(device, sysfs: str = "/sys"):
    """(power file, sclk file) of the amdgpu hwmon node behind a torch device, or (None, None)."""
    bdf = _device_bdf(device)
    roots = []
    if bdf:
        roots += glob.glob(f"{sysfs}/bus/pci/devices/{bdf}/hwmon/hwmon*")
    if not roots:                                   # one visible GPU: take the only amdgpu hwmon node there is
        for h in glob.glob(f"{sysfs}/class/hwmon/hwmon*"):
            try:
                if open(os.path.join(h, "name")).read().strip() == "amdgpu":
                    roots.append(h)
            except OSError:
                pass
        if len(roots) != 1:
            roots = []
    for root in roots:
        power = next((p for p in (os.path.join(root, n) for n in ("power1_average", "power1_input")) if os.path.exists(p)), None)
        sclk = os.path.join(root, "freq1_input")
        if power:
            return power, (sclk if os.path.exists(sclk) else None)
    return None, None


class PowerSampler:
    """``with PowerSampler(device) as ps: ...timed region...`` then ``ps.summary()``."""

    def __init__(self, device, period_s: float = 0.004) -> None:
        self.device = torch.device(device)
        self.period = period_s
        self.samples: list = []
        self._stop = threading.Event()
        self._thread = None
        self.source = None
        self._power_file, self._sclk_file = hwmon_files(self.device)
        self._smi = None
        if self._power_file:
            self.source = "sysfs hwmon " + os.path.basename(self._power_file)
        else:
            try:
                import amdsmi
                amdsmi.amdsmi_init()
                handles = amdsmi.amdsmi_get_processor_handles()
                idx = self.device.index or 0
                self._smi = (amdsmi, handles[idx if idx < len(handles) else 0])
                self.source = "amdsmi_get_power_info"
            except Exception:  # noqa: BLE001
                self._smi = None

    def _read(self):
        if self._power_file:
            try:
                w = int(open(self._power_file).read()) / 1e6
                mhz = int(open(self._sclk_file).read()) / 1e6 if self._sclk_file else None
                return w, mhz
            except (OSError, ValueError):
                return None
        if self._smi:
            amdsmi, h = self._smi
            try:
                info = amdsmi.amdsmi_get_power_info(h)
                w = info.get("current_socket_power") or info.get("average_socket_power")
                mhz = None
                try:
                    mhz = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX).get("clk")
                except Exception:  # noqa: BLE001
                    pass
                return (float(w), float(mhz) if mhz not in (None, "N/A") else None) if w not in (None, "N/A") else None
            except Exception:  # noqa: BLE001
                return None
        return None

    def _run(self):
        while not self._stop.is_set():
            r = self._read()
            if r is not None:
                self.samples.append((time.perf_counter(),) + r)
            self._stop.wait(self.period)

    def __enter__(self):
        if self.source:
            self._thread = threading.Thread(target=self._run, name="power-sampler", daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2.0)
        return False

    def summary(self) -> dict:
        if not self.samples:
            return {"package_W_mean": None, "samples": 0, "source": self.source or "unavailable (no amdgpu hwmon node, no amdsmi)"}
        w = [s[1] for s in self.samples]
        clk = [s[2] for s in self.samples if s[2] is not None]
        out = {"package_W_mean": round(sum(w) / len(w), 1), "package_W_max": round(max(w), 1), "package_W_min": round(min(w), 1),
               "samples": len(w), "period_ms": round(self.period * 1e3, 1), "source": self.source}
        if clk:
            out["sclk_MHz_mean"] = round(sum(clk) / len(clk), 1)
        return out

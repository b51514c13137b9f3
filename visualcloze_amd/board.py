"""Board state during a measurement: socket power, the power cap and the shader clock of one GPU, sampled from the amdgpu
hwmon / sysfs nodes by a background thread while bench.py's timed steps run.

Why it is part of the bench line: on MI355X this path is POWER-bound (~1.4 kW of random-operand MFMA work holds
the clock at 1.65-2.0 GHz of 2.4), and the same build reads 17.6-19.7 steps/s on different boxes; the driver-run record should
say whether its box sat at the cap.  Read-only, best effort: a box that exposes none of the nodes yields {"source": "unavailable"}
and never fails the benchmark.  Nothing here touches the product path."""
from __future__ import annotations

import glob
import os
import re
import threading
import time
from typing import Dict, List, Optional


def _read_int(path: str) -> Optional[int]:
    try:
        with open(path) as f:
            return int(f.read().strip().split()[0])
    except (OSError, ValueError, IndexError):
        return None


def _pci_of(card_dir: str) -> str:
    """'0000:c1:00.0' of /sys/class/drm/cardN (its `device` link), lower case; '' when unknown"""
    try:
        return os.path.basename(os.path.realpath(os.path.join(card_dir, "device"))).lower()
    except OSError:
        return ""


def find_hwmon(pci_bus_id: Optional[str] = None, index: int = 0, root: str = "/sys/class/drm") -> Optional[Dict[str, str]]:
    """Node paths {power, cap, sclk, pp_sclk, pci} of the amdgpu card with PCI address `pci_bus_id` (as torch reports it,
    'domain:bus:device.function'), else of the `index`-th card that has an amdgpu hwmon directory."""
    cards = []
    for card in sorted(glob.glob(os.path.join(root, "card[0-9]*")), key=lambda p: int(re.sub(r"\D", "", os.path.basename(p)) or 0)):
        if "-" in os.path.basename(card):
            continue                                   # connectors (card0-DP-1)
        hw = sorted(glob.glob(os.path.join(card, "device", "hwmon", "hwmon*")))
        if not hw:
            continue
        h = hw[0]
        power = next((p for p in (os.path.join(h, "power1_average"), os.path.join(h, "power1_input")) if os.path.exists(p)), None)
        nodes = {"power": power, "cap": os.path.join(h, "power1_cap"), "sclk": os.path.join(h, "freq1_input"),
                 "pp_sclk": os.path.join(card, "device", "pp_dpm_sclk"), "pci": _pci_of(card)}
        cards.append(nodes)
    if not cards:
        return None
    if pci_bus_id:
        want = pci_bus_id.lower()
        for c in cards:
            if c["pci"] == want or c["pci"].endswith(want) or want.endswith(c["pci"]):
                return c
    return cards[index] if index < len(cards) else cards[0]


def _pp_sclk_mhz(path: str) -> Optional[float]:
    """the active level of pp_dpm_sclk ('1: 2400Mhz *')"""
    try:
        with open(path) as f:
            for ln in f:
                if "*" in ln:
                    m = re.search(r"(\d+(?:\.\d+)?)\s*mhz", ln.lower())
                    if m:
                        return float(m.group(1))
    except OSError:
        pass
    return None


class BoardSampler:
    """with BoardSampler(pci) as b: <timed region>;  b.summary() -> {"power_w_avg", "power_w_max", "power_cap_w",
    "sclk_mhz_avg", "sclk_mhz_min", "samples", "hz", "source"}"""

    def __init__(self, pci_bus_id: Optional[str] = None, index: int = 0, hz: float = 50.0, root: str = "/sys/class/drm"):
        self.nodes = find_hwmon(pci_bus_id, index, root)
        self.period = 1.0 / hz
        self.power: List[float] = []
        self.sclk: List[float] = []
        self._stop = threading.Event()
        self._thread: Optional[threading.Thread] = None
        self.t0 = self.t1 = 0.0

    def _sample(self) -> None:
        n = self.nodes
        if n["power"]:
            v = _read_int(n["power"])
            if v is not None:
                self.power.append(v * 1e-6)            # microwatts
        v = _read_int(n["sclk"])
        if v is not None:
            self.sclk.append(v * 1e-6)                 # Hz
        else:
            mhz = _pp_sclk_mhz(n["pp_sclk"])
            if mhz is not None:
                self.sclk.append(mhz)

    def _run(self) -> None:
        while not self._stop.is_set():
            self._sample()
            self._stop.wait(self.period)

    def __enter__(self) -> "BoardSampler":
        self.t0 = time.perf_counter()
        if self.nodes:
            self._thread = threading.Thread(target=self._run, name="board-sampler", daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc) -> None:
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2.0)
            self._sample()                              # one more at the closing edge
        self.t1 = time.perf_counter()

    def summary(self) -> dict:
        if not self.nodes:
            return {"source": "unavailable", "note": "no amdgpu hwmon node under /sys/class/drm"}
        cap = _read_int(self.nodes["cap"])
        out = {"source": "sysfs hwmon " + (self.nodes["pci"] or "?"), "samples": max(len(self.power), len(self.sclk)),
               "hz": round(max(len(self.power), len(self.sclk)) / max(self.t1 - self.t0, 1e-9), 1),
               "power_cap_w": round(cap * 1e-6, 1) if cap is not None else None}
        out["power_w_avg"] = round(sum(self.power) / len(self.power), 1) if self.power else None
        out["power_w_max"] = round(max(self.power), 1) if self.power else None
        out["sclk_mhz_avg"] = round(sum(self.sclk) / len(self.sclk), 1) if self.sclk else None
        out["sclk_mhz_min"] = round(min(self.sclk), 1) if self.sclk else None
        return out


def pci_bus_id_of(device_index: int = 0) -> Optional[str]:
    """'0000:c1:00.0'-style address of a torch device (None when torch does not expose it)"""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        dom, bus, dev = getattr(p, "pci_domain_id", None), getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None)
        if bus is None or dev is None:
            return None
        return f"{(dom or 0):04x}:{bus:02x}:{dev:02x}.0"
    except Exception:
        return None

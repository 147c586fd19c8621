"""Board power / engine clock sampled in a background thread while a timed loop runs (bench.py's `roofline.power_w` /
`roofline.sclk_mhz`, tools/*_probe.py).  Measurement plumbing only: nothing on the product path imports it.

Sources, first one that answers: the amdsmi Python binding shipped under /opt/rocm/share/amd_smi, the hwmon / pp_dpm files of
the amdgpu driver in sysfs, the `rocm-smi` command line (what profiles/r02s_power.log was taken with)."""
import glob
import os
import re
import subprocess
import sys
import threading
import time


class _AmdSmi:
    name = "amdsmi"

    def __init__(self, index):
        try:
            import amdsmi  # noqa: F401
        except Exception:
            sys.path.insert(0, "/opt/rocm/share/amd_smi")
            import amdsmi  # noqa: F401
        self.m = amdsmi
        amdsmi.amdsmi_init()
        hs = amdsmi.amdsmi_get_processor_handles()
        self.h = hs[index if index < len(hs) else 0]
        self.read()  # raises when the calls below are not there

    def read(self):
        m = self.m
        p = m.amdsmi_get_power_info(self.h)
        w = None
        for key in ("current_socket_power", "average_socket_power", "socket_power"):
            v = p.get(key)
            if isinstance(v, (int, float)) and v > 0:
                w = float(v)
                break
        c = m.amdsmi_get_clock_info(self.h, m.AmdSmiClkType.GFX)
        mhz = c.get("clk", c.get("cur_clk"))
        return w, float(mhz) if isinstance(mhz, (int, float)) else None


class _Sysfs:
    name = "sysfs"

    def __init__(self, index):
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
        if not cards:
            raise RuntimeError("no amdgpu card in sysfs")
        dev = os.path.dirname(cards[index if index < len(cards) else 0])
        self.sclk = os.path.join(dev, "pp_dpm_sclk")
        self.power = None
        for pat in ("hwmon/hwmon*/power1_average", "hwmon/hwmon*/power1_input"):
            hits = glob.glob(os.path.join(dev, pat))
            if hits:
                self.power = hits[0]
                break
        self.freq = (glob.glob(os.path.join(dev, "hwmon/hwmon*/freq1_input")) or [None])[0]
        w, mhz = self.read()
        if w is None and mhz is None:
            raise RuntimeError("sysfs answers neither power nor clock")

    def read(self):
        w = mhz = None
        try:
            if self.power:
                w = int(open(self.power).read()) / 1e6
        except Exception:
            pass
        try:
            if self.freq:
                mhz = int(open(self.freq).read()) / 1e6
            else:
                for line in open(self.sclk):
                    if "*" in line:
                        mhz = float(re.search(r"(\d+)\s*[Mm][Hh]z", line).group(1))
        except Exception:
            pass
        return w, mhz


class _RocmSmi:
    name = "rocm-smi"

    def __init__(self, index):
        self.index = index
        w, mhz = self.read()
        if w is None and mhz is None:
            raise RuntimeError("rocm-smi answers neither power nor clock")

    def read(self):
        out = subprocess.run(["rocm-smi", "-d", str(self.index), "--showpower", "--showclocks"], capture_output=True, text=True,
                             timeout=10).stdout
        w = re.search(r"Power \(W\):\s*([0-9.]+)", out)
        s = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", out)
        return (float(w.group(1)) if w else None), (float(s.group(1)) if s else None)


def open_source(index=0):
    errs = []
    for cls in (_AmdSmi, _Sysfs, _RocmSmi):
        try:
            return cls(index)
        except Exception as e:  # noqa: BLE001 - any failure means "try the next source"
            errs.append(f"{cls.name}: {type(e).__name__}: {e}")
    raise RuntimeError("no power / clock source: " + "; ".join(errs))


class PowerMonitor:
    """with PowerMonitor() as pm: <timed loop>; pm.summary() -> {"power_w": mean, "sclk_mhz": mean, "samples": n, "source": ...}.
    Never raises on a box without a source: the summary then says {"source": None, "error": ...}."""

    def __init__(self, index=0, period=0.05, skip=0.25):
        self.period, self.skip = period, skip
        self.samples = []
        self.err = None
        try:
            self.src = open_source(index)
        except Exception as e:  # noqa: BLE001
            self.src, self.err = None, str(e)
        self._stop = threading.Event()
        self._th = None

    def _run(self):
        t0 = time.perf_counter()
        while not self._stop.is_set():
            try:
                w, mhz = self.src.read()
                self.samples.append((time.perf_counter() - t0, w, mhz))
            except Exception as e:  # noqa: BLE001
                self.err = str(e)
            self._stop.wait(self.period)

    def __enter__(self):
        if self.src is not None:
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._th is not None:
            self._th.join(timeout=15)
        return False

    def summary(self):
        if self.src is None:
            return {"source": None, "error": self.err}
        s = [x for x in self.samples if x[0] >= self.skip] or self.samples  # the first samples still see the idle clock
        ws = [x[1] for x in s if x[1] is not None]
        cs = [x[2] for x in s if x[2] is not None]
        mean = lambda v: round(sum(v) / len(v), 1) if v else None  # noqa: E731
        return {"source": self.src.name, "samples": len(s), "power_w": mean(ws), "sclk_mhz": mean(cs),
                "power_w_max": max(ws) if ws else None, "sclk_mhz_min": min(cs) if cs else None}


if __name__ == "__main__":
    with PowerMonitor() as pm:
        time.sleep(1.0)
    print(pm.summary())

"""Dev measurement (round 5): which shader clock and board power does the MI355X sustain UNDER the headline kernel?
The fp64 roofline of bench.py is priced at the peak clock (78.6 TFLOP/s = 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz). This script steps the
4096^2 grid (fused planar RK2 step, FAST, `riemann`) for `seconds` while a second thread samples the clock and power files the amdgpu driver
exposes to an ordinary user (sysfs: hwmon freq*_input / power*_average|input, pp_dpm_sclk) - whatever of them exists on the box; it changes nothing.
Workloads: blast (mostly quiescent operands), wave (every cell busy), idle (no launches).
usage: python scripts/clock_under_load.py [seconds=8] [riemann=hllc]   -> one JSON line per workload
       python scripts/clock_under_load.py [seconds=6] configs          -> the kernels of BASELINE configs 3, 4 and 5 (FAST and STRICT) the same way"""
import glob, json, os, re, sys, threading, time
sys.path.insert(0, ".")
import numpy as np
from mara3_amd import setups
from mara3_amd.slab import NativeSlabStepper

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
riemann = sys.argv[2] if len(sys.argv) > 2 else "hllc"


def own_pci_bus_id():
    """the PCI address of HIP device 0 of this process (sysfs lists every board of the host, other tenants' included)"""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
            return buf.value.decode().lower()
    except OSError:
        pass
    return None


def sources():
    out = {}
    own = own_pci_bus_id()
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
    mine = [c for c in cards if own and os.path.realpath(c).lower().endswith(own)]
    for card in (mine or cards):
        for f in sorted(glob.glob(card + "/hwmon/hwmon*/freq*_input")):
            label = f.replace("_input", "_label")
            name = open(label).read().strip() if os.path.exists(label) else os.path.basename(f)
            out["%s:%s" % (card.split("/")[4], name)] = ("hz", f)
        for f in sorted(glob.glob(card + "/hwmon/hwmon*/power*_average")) + sorted(glob.glob(card + "/hwmon/hwmon*/power*_input")):
            out["%s:%s" % (card.split("/")[4], os.path.basename(f))] = ("uw", f)
        if os.path.exists(card + "/pp_dpm_sclk"):
            out["%s:pp_dpm_sclk" % card.split("/")[4]] = ("dpm", card + "/pp_dpm_sclk")
    return out


def read(kind, path):
    try:
        text = open(path).read()
    except OSError:
        return None
    if kind == "hz":
        return float(text) / 1e6                     # MHz
    if kind == "uw":
        return float(text) / 1e6                     # W
    m = [l for l in text.splitlines() if l.strip().endswith("*")]
    if not m:
        return None
    g = re.search(r"(\d+)\s*Mhz", m[0], re.I)
    return float(g.group(1)) if g else None


class Sampler(threading.Thread):
    def __init__(self, src):
        super().__init__(daemon=True)
        self.src, self.stop, self.samples = src, False, {k: [] for k in src}
    def run(self):
        while not self.stop:
            for k, (kind, path) in self.src.items():
                v = read(kind, path)
                if v is not None:
                    self.samples[k].append(v)
            time.sleep(0.02)
    def summary(self, skip=0.25):
        out = {}
        for k, v in self.samples.items():
            v = v[int(len(v) * skip):]               # the first quarter: the clock is still settling
            if v:
                s = sorted(v)
                out[k] = {"mean": round(sum(v) / len(v), 1), "min": s[0], "median": s[len(s) // 2], "max": s[-1], "n": len(v)}
        return out


def run_configs(src):
    """configs 5 (3-D Euler 512^3), 3 (`binary` 2048^2) in this process; config 4 (`cloud` 4096^2) through the compiled host as bench_configs.py runs it:
    its set-up and file output are idle time, so its figures are taken over the samples within 10 % of the highest power seen"""
    import subprocess, tempfile
    from mara3_amd import binary
    from mara3_amd.engine import EulerCartSolver
    n5, gamma = 512, 5.0 / 3
    only = sys.argv[3] if len(sys.argv) > 3 else ""
    for arith in ("fast", "strict") if only in ("", "c5") else ():
        s5 = EulerCartSolver((n5,) * 3, (1.0 / n5,) * 3, gamma, 1.5, "hlle", 2, "outflow", arith=arith)
        s5.upload(setups.blast_ic((n5,) * 3, gamma))
        s5.step(setups.baseline_dt(n5), 5); s5.synchronize()
        sm = Sampler(src); sm.start()
        t0 = time.perf_counter(); steps = 0
        while time.perf_counter() - t0 < seconds:
            s5.step(setups.baseline_dt(n5), 20); s5.synchronize(); steps += 20
        el = time.perf_counter() - t0
        sm.stop = True; sm.join()
        print(json.dumps({"workload": "c5 euler3d 512^3 hlle", "arith": arith, "seconds": round(el, 2), "steps": steps, "ms_per_step": round(el / steps * 1e3, 4),
                          "Mzones_per_s": round(n5 ** 3 * steps / el / 1e6, 1), "pci_bus_id": own_pci_bus_id(), "sensors": sm.summary()}), flush=True)
        s5.close()
    cfg = binary.config(depth=5, block_size=64, fixed_dt=1, rk_order=2, plm_theta=1.8)
    n3 = binary.grid_size(cfg)
    for arith in ("fast", "strict") if only in ("", "c3") else ():
        s3 = binary.BinarySolver(cfg, arith=arith)
        s3.next(20)
        sm = Sampler(src); sm.start()
        t0 = time.perf_counter(); steps = 0
        while time.perf_counter() - t0 < seconds:
            s3.next(500); steps += 500
        el = time.perf_counter() - t0
        sm.stop = True; sm.join()
        print(json.dumps({"workload": "c3 binary 2048^2", "arith": arith, "seconds": round(el, 2), "steps": steps, "ms_per_step": round(el / steps * 1e3, 4),
                          "Mzones_per_s": round(n3 * n3 * steps / el / 1e6, 1), "pci_bus_id": own_pci_bus_id(), "sensors": sm.summary()}), flush=True)
        s3.close()
    # config 4's kernels (`cloud`, SRHD 4096^2): the reference's own problem throws after 335 steps at this size (tests/test_gpu_long_runs_vs_reference.py),
    # too short for the clock to settle - so a smooth relativistic flow on the same grid shape, fixed nozzle row, the same kernels
    from mara3_amd import engine
    nr = nq = 4096
    rv, qv = np.logspace(0.0, 0.6, nr + 1), np.linspace(0.0, np.pi, nq + 1)
    rc, qc = 0.5 * (rv[1:] + rv[:-1])[:, None], 0.5 * (qv[1:] + qv[:-1])[None, :]
    P = np.zeros((nr, nq, 5))
    P[..., 0] = (1.0 + 0.3 * np.sin(2.0 * qc) * np.cos(1.5 * np.log(rc))) / rc ** 2
    P[..., 1] = 0.8 + 0.5 * np.cos(2.5 * qc) * np.sin(2.0 * np.log(rc))
    P[..., 2] = 0.1 * np.sin(2.0 * qc) * np.cos(2.0 * np.log(rc))
    P[..., 4] = 0.05 * P[..., 0] * (1.0 + 0.5 * np.sin(1.5 * qc))
    U = engine.srhd_to_conserved(P.reshape(-1, 5)).reshape(nr, nq, 5)
    dmu = -np.cos(qv[1:]) - -np.cos(qv[:-1])
    U *= (((rv[1:] ** 3 - rv[:-1] ** 3)[:, None] * dmu[None, :] * 2 * np.pi) / 3)[..., None]
    row = np.ascontiguousarray(P[0])
    dtc = 0.3 * (rv[1] - rv[0])
    for arith in ("fast", "strict") if only in ("", "c4") else ():
        s4 = engine.CloudSolver(rv, qv, 2, 1.2, 0.0, arith=arith)
        s4.upload(U); s4.set_inflow(row)
        s4.step(dtc, 5)
        sm = Sampler(src); sm.start()
        t0 = time.perf_counter(); steps = 0; note = None
        try:
            while time.perf_counter() - t0 < seconds:
                s4.step(dtc, 50); steps += 50
                if s4.status():
                    note = "status word %#x after %d steps" % (s4.status(), steps); break
        except Exception as e:
            note = repr(e)[:200]
        el = time.perf_counter() - t0
        sm.stop = True; sm.join()
        print(json.dumps({"workload": "c4 cloud kernels 4096^2 (smooth relativistic flow, fixed nozzle row)", "arith": arith, "seconds": round(el, 2), "steps": steps,
                          "ms_per_step": round(el / max(1, steps) * 1e3, 4), "Mzones_per_s": round(nr * nq * steps / el / 1e6, 1), "planar": s4.is_planar(), "note": note,
                          "pci_bus_id": own_pci_bus_id(), "sensors": sm.summary()}), flush=True)
        s4.close()


src = sources()
if not src:
    print(json.dumps({"error": "no clock or power file of the amdgpu driver is readable here"}))
if riemann == "configs":
    run_configs(src)
    sys.exit(0)
n, gamma = 4096, 5.0 / 3
dt = setups.baseline_dt(n)
for workload in ("idle", "blast", "wave", "idle_after"):
    st = None
    if workload in ("blast", "wave"):
        bc = "outflow" if workload == "blast" else "periodic"
        st = NativeSlabStepper((n, n), (1.0 / n, 1.0 / n), gamma, 1.5, riemann, 2, bc, arith="fast")
        st.load_slab(setups.blast_ic((n, n), gamma) if workload == "blast" else setups.smooth_wave_ic((n, n), gamma))
        st.step(dt, 50); st.synchronize()
    s = Sampler(src); s.start()
    t0 = time.perf_counter(); steps = 0
    while time.perf_counter() - t0 < (seconds if st else 2.0):
        if st:
            st.step(dt, 200); st.synchronize(); steps += 200
        else:
            time.sleep(0.1)
    elapsed = time.perf_counter() - t0
    s.stop = True; s.join()
    rec = {"workload": workload, "riemann": riemann, "seconds": round(elapsed, 2), "steps": steps, "pci_bus_id": own_pci_bus_id(), "sensors": s.summary()}
    if st:
        rec["us_per_step"] = round(elapsed / steps * 1e6, 2)
        rec["Mcells_per_s"] = round(n * n * steps / elapsed / 1e6, 1)
        rec["planar"] = bool(st.is_planar())
        st.close()
    print(json.dumps(rec), flush=True)

"""Is the headline kernel power-bound?  Samples the GPU's power / clock sensors (sysfs hwmon, amdgpu) every ~20 ms in a child
process while the timed workload runs in this one, and prints the distribution per phase: idle, exact-fp32 kernel, f16x3 kernel,
two-term kernel.  (DESIGN section 5 "power": three builds with different cycle counts took the same time; this is the direct look.)

    python scripts/power_probe.py [tiles=1024] [seconds_per_phase=4] [grad]        (grad: time log_prob_grad instead of log_prob)
"""
import glob
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sensors():
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "temp2_input", "power1_cap"):
            p = os.path.join(hw, name)
            if os.path.exists(p):
                out.setdefault(hw, {})[name] = p
    for dev in glob.glob("/sys/class/drm/card*/device"):
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "gpu_busy_percent"):
            p = os.path.join(dev, name)
            if os.path.exists(p):
                out.setdefault(dev, {})[name] = p
    return out


def read(p):
    try:
        return open(p).read().strip()
    except OSError:
        return None


def sampler(stop, q, period):
    s = sensors()
    only = os.environ.get("PROBE_CARD")          # e.g. card16: sample that card's sensors only
    rows = []
    while not stop.is_set():
        t = time.time()
        row = {"t": t}
        for hw, d in s.items():
            if only and only not in hw:
                continue
            for name, p in d.items():
                v = read(p)
                if v is None:
                    continue
                if name.startswith("pp_dpm"):
                    cur = [l for l in v.splitlines() if l.endswith("*")]
                    v = cur[0].split(":")[1].strip(" *") if cur else None
                row[os.path.basename(os.path.dirname(hw)) + "/" + os.path.basename(hw) + "/" + name if "hwmon" in hw else name] = v
        rows.append(row)
        time.sleep(period)
    q.put(rows)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    grad = len(sys.argv) > 3 and sys.argv[3] == "grad"
    print("sensors:", json.dumps({k: sorted(v) for k, v in sensors().items()}))
    stop, q = mp.Event(), mp.Queue()
    proc = mp.Process(target=sampler, args=(stop, q, 0.02))
    proc.start()                      # (forked before this process touches the GPU)
    import torch
    from audiosourcesep_amd import _lib
    from audiosourcesep_amd.config import CONFIG_B
    from audiosourcesep_amd.synthetic import calibrated_engine, synthetic_mel_tiles
    eng, _ = calibrated_engine(CONFIG_B, device=0, init_tiles=n)
    eng.set_range_policy("error")
    x = torch.from_numpy(synthetic_mel_tiles(n, CONFIG_B, seed=1234)).cuda()
    lp = torch.empty(n, device="cuda")
    phases = []
    torch.cuda.synchronize()
    t0 = time.time(); time.sleep(1.0); phases.append(("idle", t0, time.time(), None))
    for name, prec in (("f32", _lib.PREC_F32), ("f16x3", _lib.PREC_F16X3), ("f16x2", _lib.PREC_F16X2), ("f16x3_again", _lib.PREC_F16X3)):
        eng.set_precision(prec)
        call = (lambda: eng.log_prob_grad(x)) if grad else (lambda: eng.log_prob(x, out=lp))
        call(); torch.cuda.synchronize()
        t0 = time.time(); k = 0
        while time.time() - t0 < secs:
            for _ in range(4):
                call()
            torch.cuda.synchronize(); k += 4
        t1 = time.time()
        phases.append((name, t0, t1, n * k / (t1 - t0)))
        time.sleep(0.5)
    stop.set()
    rows = q.get(timeout=20)
    proc.join(timeout=5)
    keys = sorted({k for r in rows for k in r if k != "t"})
    print("samples:", len(rows))
    for name, a, b, rate in phases:
        sel = [r for r in rows if a + 0.3 <= r["t"] <= b - 0.1]
        line = {"phase": name, "passes_per_s": rate, "samples": len(sel)}
        for k in keys:
            vals = []
            for r in sel:
                try:
                    vals.append(float(str(r.get(k)).rstrip("Mhz")))
                except (TypeError, ValueError):
                    pass
            if vals:
                vals.sort()
                line[k] = {"min": vals[0], "med": vals[len(vals) // 2], "max": vals[-1]}
        print(json.dumps(line))


if __name__ == "__main__":
    main()

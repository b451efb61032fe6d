"""Clock / power trace of the prefill loop (is the GEMM-dense prefill power-limited?).

A sampler thread reads the GPU's hwmon files (shader clock, socket power) every ~20 ms while the main thread runs, in turn:
idle -> the big-tile GEMM alone (gate_up shape) -> the whole prefill in a loop -> the decode graph in a loop.
Writes profiles/<tag>_power_clock.txt (per-phase mean / min / max clock and power + the achieved rate of each phase).

    python tools/power_trace.py [tag]            (GPU box; no root needed: hwmon is world-readable)
"""
import glob
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]


def find_sensors():
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name, key in (("freq1_input", "sclk_hz"), ("freq2_input", "mclk_hz"), ("power1_average", "power_uw"), ("power1_input", "power_uw"),
                          ("temp1_input", "temp_mc")):
            f = os.path.join(hw, name)
            if os.path.exists(f) and key not in out:
                out[key] = f
        if out:
            break
    return out


class Sampler(threading.Thread):
    def __init__(self, sensors, period=0.02):
        super().__init__(daemon=True)
        self.sensors, self.period, self.rows, self.stop = sensors, period, [], False

    def run(self):
        while not self.stop:
            row = {"t": time.perf_counter()}
            for k, f in self.sensors.items():
                try:
                    row[k] = float(open(f).read().strip())
                except Exception:
                    row[k] = float("nan")
            self.rows.append(row)
            time.sleep(self.period)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    sensors = find_sensors()
    sm = Sampler(sensors)
    sm.start()
    import numpy as np
    import torch
    from phi_3_vision_mlx_amd import ops
    from phi_3_vision_mlx_amd.api import load_synthetic
    from phi_3_vision_mlx_amd.workloads import vqa_request
    model, proc = load_synthetic(seed=0, device="cuda:0")
    inp = vqa_request(proc.img_processor, 0, device="cuda:0")
    S = inp["input_ids"].shape[1]
    phases = []

    def phase(name, fn, seconds, work_per_call, unit):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds:
            fn()
            n += 1
            if n % 8 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        phases.append((name, t0, t1, n * work_per_call / (t1 - t0), unit, (t1 - t0) / n * 1e3))

    time.sleep(2.0)
    phases.append(("idle", time.perf_counter() - 2.0, time.perf_counter(), 0.0, "", 0.0))
    A = torch.randn(4096, 3072, device="cuda").bfloat16()
    W = torch.randn(16384, 3072, device="cuda").bfloat16() * 0.02
    phase("gemm 4096x16384x3072 (big tiles, SiLU*up)", lambda: ops.gemm(A, W, ops.EPI_SILU_MUL), 6.0, 2.0 * 4096 * 16384 * 3072 / 1e12, "TFLOP/s")
    pf = 25.94                                                    # algorithmic TFLOP of the bench prefill (bench.py)
    phase("prefill (bench request, 2531 tokens)", lambda: model(**inp, max_tokens=16), 10.0, pf, "TFLOP/s")
    tok, cache = model.greedy_prefill(4000, **inp)
    state = {"tok": tok}

    def step():
        if cache[0].state.offset + 2 > cache[0].state.T:
            cache[0].state.offset = S
        _, state["tok"] = model.greedy_step(state["tok"], cache)
    phase("decode (graph-replayed steps)", step, 6.0, 1.0, "tokens/s")
    sm.stop = True
    sm.join()
    lines = [f"sensors: {sensors}", f"{len(sm.rows)} samples at ~{sm.period * 1e3:.0f} ms"]
    for name, t0, t1, rate, unit, ms in phases:
        rows = [r for r in sm.rows if t0 + 0.3 <= r["t"] <= t1]           # skip the ramp of the first 300 ms
        def stat(k, scale):
            v = np.array([r.get(k, float("nan")) for r in rows]) * scale
            v = v[~np.isnan(v)]
            return "n/a" if v.size == 0 else f"mean {v.mean():8.1f}  min {v.min():8.1f}  max {v.max():8.1f}"
        lines.append(f"{name:48s} {rate:9.1f} {unit:9s} {ms:8.3f} ms/call | sclk MHz {stat('sclk_hz', 1e-6)} | power W {stat('power_uw', 1e-6)} | "
                     f"temp C {stat('temp_mc', 1e-3)}")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    txt = "\n".join(lines)
    open(os.path.join(ROOT, "gpurun_out", f"{tag}_power_clock.txt"), "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()

"""Config 5 decode (fp8 weights + int8 KV): every kernel of the step against its own streaming floor.  The floor of a launch that
reads `bytes` once inside a hipGraph on this chip is 2.4 us + bytes / 7.2 TB/s (tools/stream_floor.hip).  Per-kernel in-graph times come
from tools/trace_decode.sh run with P3V_FP8=1 P3V_QCACHE=1 (its summary file is argv[1]); bytes are algorithmic (weights e4m3 + row
scales; int8 K / V + fp32 scales per token and head).  python tools/c5_floor.py gpurun_out/trace_r5c5_summary.txt [ctx]"""
import re
import sys
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2531
H, I, V, nh, hd = 3072, 8192, 32064, 32, 96
BYTES = {"qkv": 9216 * H + 9216 * 4, "o_proj": H * H + H * 4, "gate_up": 2 * I * H + 2 * I * 4, "down": H * I + H * 4, "lm_head": V * H + V * 4,
         "attn": 2 * ctx * nh * hd + 2 * ctx * nh * 4}
BYTES["attn+o_proj"] = BYTES["attn"] + BYTES["o_proj"]
rows = []
for line in open(sys.argv[1]):
    m = re.match(r"\s+(\S.*?)\s+grid\s+(\d+)\s+calls/step\s+([\d.]+)\s+avg\s+([\d.]+) us\s+total/step\s+([\d.]+) us", line)
    if not m:
        if line.startswith("decode step"): print(line.strip())
        continue
    name, calls, avg, tot = m.group(1), float(m.group(3)), float(m.group(4)), float(m.group(5))
    role = re.search(r"\[(\w+)\]", name)
    key = role.group(1) if role else ("attn+o_proj" if "q8" in name and "attn" in name else None)
    rows.append((name, calls, avg, tot, key))
print(f"{'kernel':58s} {'calls':>5s} {'avg us':>8s} {'MB':>8s} {'floor us':>9s} {'avg/floor':>9s} {'TB/s':>6s}")
tot_t = tot_f = 0.0
for name, calls, avg, tot, key in rows:
    if key in BYTES:
        b = BYTES[key]
        floor = 2.4 + b / 7.2e6
        print(f"{name:58s} {calls:5.0f} {avg:8.2f} {b / 1e6:8.2f} {floor:9.2f} {avg / floor:9.2f} {b / avg / 1e6:6.2f}")
        tot_f += floor * calls
    else:
        print(f"{name:58s} {calls:5.0f} {avg:8.2f} {'-':>8s} {'-':>9s} {'-':>9s} {'-':>6s}")
        tot_f += avg * calls
    tot_t += tot
print(f"sum of kernels {tot_t:.1f} us/step; with every streaming kernel at its floor {tot_f:.1f} us/step")

#!/bin/bash
# SQ counters of the three prompt-sized attention kernels (one counter group per pass, kernel-trace only, as
# MI355X_MICROARCH.md prescribes): tools/pmc_attn_prefill.sh [L] -> gpurun_out/pmc_attn/summary.txt
L=${1:-8192}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_attn
rm -rf $OUT; mkdir -p $OUT
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD"
P3="SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM"
cd /tmp
for pp in ${PMC_KERNELS:-0 1 2}; do
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pp${pp}_p$i -o c -- python3 $GRAFT_REPO_ROOT/tools/pmc_attn_prefill.py $L $pp > $OUT/pp${pp}_p$i.log 2>&1
  done
done
cd - > /dev/null
python3 - "$OUT" "$L" <<'PY'
import csv, sys, glob, collections
out, L = sys.argv[1], int(sys.argv[2])
lines = [f"prompt-sized attention, B = 1, {L} tokens, 32 heads x 96, causal, q pre-scaled; per launch (mean of the last 2 of 4 launches)"]
for pp, key in ((0, "k_attn_prefill_dma"), (1, "k_attn_prefill_pp"), (2, "k_attn_prefill_il")):
    acc = collections.defaultdict(list)
    dur = []
    for f in glob.glob(f"{out}/pp{pp}_p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"{out}/pp{pp}_p1/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    c = {k: sum(v[-2:]) / max(1, len(v[-2:])) for k, v in acc.items()}
    lines.append(f"{key}: launch {sum(dur[-2:]) / max(1, len(dur[-2:])):.1f} us under the counters")
    for k in sorted(c):
        lines.append(f"   {k:32s} {c[k]:16.0f}")
    w = c.get("SQ_WAVES", 0)
    if not dur: continue
    if w:
        tiles = {0: L / 128 * (L / 128 + 1) / 2 * 2 * 32 * 4, 1: L / 256 * (L / 256 + 1) / 2 * 4 * 32 * 8, 2: L / 256 * (L / 256 + 1) / 2 * 4 * 32 * 8}[pp]   # wave-tiles (64 keys x 32 queries)
        lines.append(f"   per wave-tile (64 keys x 32 queries; {tiles:.0f} of them): VALU insts {c.get('SQ_INSTS_VALU', 0) / tiles:.1f}, "
                     f"of which transcendental {c.get('SQ_INSTS_VALU_TRANS_F32', 0) / tiles:.1f}, MFMA {c.get('SQ_INSTS_MFMA', 0) / tiles:.1f}, "
                     f"LDS {c.get('SQ_INSTS_LDS', 0) / tiles:.1f}, SALU {c.get('SQ_INSTS_SALU', 0) / tiles:.1f}, VMEM {c.get('SQ_INSTS_VMEM', 0) / tiles:.1f}; "
                     f"wave quad-cycles {c.get('SQ_WAVE_CYCLES', 0) / tiles:.0f} (wait_any {c.get('SQ_WAIT_ANY', 0) / tiles:.0f}, "
                     f"wait_inst_any {c.get('SQ_WAIT_INST_ANY', 0) / tiles:.0f}, active_any {c.get('SQ_ACTIVE_INST_ANY', 0) / tiles:.0f}, "
                     f"active_valu {c.get('SQ_ACTIVE_INST_VALU', 0) / tiles:.0f})")
        if c.get("SQ_BUSY_CYCLES"):
            lines.append(f"   MFMA busy / SQ busy cycles: {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / c['SQ_BUSY_CYCLES']:.3f}   "
                         f"LDS bank-conflict / active: {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, c.get('SQ_LDS_IDX_ACTIVE', 0)):.3f}")
open(f"{out}/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY

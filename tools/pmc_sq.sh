#!/bin/bash
# SQ counter passes for the synthesis / pre-pass kernels (own runs, kernel-trace only).
#   bash tools/pmc_sq.sh <tag> [kbench args, default "--blocks 40": 1e8 samples of the 25 MS/s workload per launch]
#   M1 geometry (k_synth_ev_dense): bash tools/pmc_sq.sh <tag> --fs 2.6e6 --nsamp 300000 --nch 12 --blocks 333
set -u
TAG="${1:-sq}"
shift || true
KARGS="${*:---blocks 40}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d "$OUT/sq1" -o pmc -- python /root/repo/tools/kbench.py --steps 2 --warmup 1 --no-cpu $KARGS > "$OUT/sq1.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d "$OUT/sq2" -o pmc -- python /root/repo/tools/kbench.py --steps 2 --warmup 1 --no-cpu $KARGS > "$OUT/sq2.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64 -d "$OUT/sq3" -o pmc -- python /root/repo/tools/kbench.py --steps 2 --warmup 1 --no-cpu $KARGS > "$OUT/sq3.log" 2>&1
cd /root/repo
python - "$OUT" <<'PY'
import sqlite3, glob, sys, os
for sub in ("sq1","sq2","sq3"):
    for db in glob.glob(os.path.join(sys.argv[1], sub, "*.db")):
        c = sqlite3.connect(db)
        try:
            rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
        except Exception as e:
            print(sub, "ERR", e); continue
        for k, n, v, cnt in rows:
            if "gpsbb" in k:
                print("%-24s %-28s %18.1f  (n=%d)" % (k.split("::")[1].split("(")[0][:24], n, v, cnt))
PY
tail -3 "$OUT"/sq*.log | grep -i -E "error|fail|invalid" | head

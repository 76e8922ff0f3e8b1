#!/bin/bash
# SQ counters of the score-volume kernels (two passes), printed per kernel.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O; export TMPDIR=/tmp; cd /tmp
python $R/tools/micro/score_bench.py 8 5
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $O/pmc_score1 -o s -- python $R/tools/micro/score_bench.py 8 2 > $O/pmc_score1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM -d $O/pmc_score2 -o s -- python $R/tools/micro/score_bench.py 8 2 > $O/pmc_score2.log 2>&1
python - <<PY
import sqlite3, glob
for d in ("$O/pmc_score1", "$O/pmc_score2"):
    for f in glob.glob(d + "/**/*.db", recursive=True):
        db = sqlite3.connect(f)
        rows = db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%score_sweep%' group by kernel_name, counter_name").fetchall()
        for r in rows:
            print(r[0][60:110], r[1], f"{r[2]:.4g}", r[3])
PY
tail -3 $O/pmc_score1.log $O/pmc_score2.log

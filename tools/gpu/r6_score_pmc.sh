#!/bin/bash
# Round 6 (round-5 verdict, next #6): where do the score-volume kernel's extra HBM bytes come from?  L2 <-> fabric request counters of pass B
# (score_sweep_kernel<1>), split by request size, and the L2 hit rate; separate --pmc passes (TCC has 4 counters per pass).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O; export TMPDIR=/tmp; cd /tmp
python -c 'import torch' 2> /dev/null
rocprofv3 -L 2> /dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $O/r06_tcc_counters.txt
python $R/tools/micro/score_bench.py 8 5 2>&1 | grep -v amdgpu
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $O/pmc_scoreB$i -o s -- python $R/tools/micro/score_bench.py 8 2 > $O/pmc_scoreB$i.log 2>&1 || tail -3 $O/pmc_scoreB$i.log
done
python - <<PY > $O/r06_score_pmc.txt
import sqlite3, glob
print("# rocprofv3 --pmc, tools/micro/score_bench.py 8 2 (8 pairs, L = S = 4800, C = 256): averages per launch")
print("# algorithmic bytes of pass B: descriptors 8 x (4800 + 4800) x 256 x 4 = 78.6 MB read, conf_matrix 8 x 4800 x 4800 x 4 = 737.3 MB written")
for i in range(1, 7):
    for f in glob.glob("$O/pmc_scoreB%d/**/*.db" % i, recursive=True):
        db = sqlite3.connect(f)
        tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
        t = next((x for x in tabs if x.startswith("counters_collection")), None)
        if not t: print("no counters table in", f, tabs[:8]); continue
        rows = db.execute(f"select kernel_name, counter_name, avg(value), count(*) from {t} where kernel_name like '%score_sweep%' group by kernel_name, counter_name").fetchall()
        for r in rows:
            k = r[0]; k = k[k.find("score_sweep"):][:44]
            print(f"{k:46s} {r[1]:24s} {r[2]:16.6g}  ({r[3]} launches)")
PY
cat $O/r06_score_pmc.txt
rm -rf $O/pmc_scoreB*

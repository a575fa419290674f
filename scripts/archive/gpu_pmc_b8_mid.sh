#!/bin/bash
# HBM traffic of the two 8-row launches AT the bench's cache length (kv 3858): PMC passes over profile_decode steps only (graph off), so that
# every rows_attn dispatch of the pass streams the same 134.8 MB.  -> profiles/r05_pmc_decode_traffic_b8_kv3858.json (imported by bench.py --batch 8).
mkdir -p gpurun_out/r5pmc; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5pmc
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -o p --output-format csv -- python $R/scripts/prof_step.py --batch 8 --options "use_graph=0" --steps 4 --lens 3858 > $O/pmc_${C}.log 2>&1
done
python $R/scripts/pmc_summary.py $O/pmc_decode_raw.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $O/pmc_decode_summary.log 2>&1
echo '{}' > $O/empty.json
python $R/scripts/pmc_r2_report.py $O/pmc_decode_raw.json $O/empty.json $O r05_b8_kv3858
python - <<PY
import json
d = json.load(open("$O/r05_b8_kv3858_pmc_decode_traffic.json"))
print(d["hbm_bytes_per_launch"], d["dispatches"]); print({k: (v["dispatches"], v["hbm_bytes_per_launch"]) for k, v in d["per_kernel"].items()})
PY
tail -3 $O/pmc_FETCH_SIZE.log

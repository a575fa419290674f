#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5j; mkdir -p $O
R=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1 || { tail -20 $O/build.txt; exit 1; }


cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -o p --output-format csv -- python $R/scripts/prof_step.py --batch 8 --options "use_graph=0" --steps 2 --gen 256 --no-profile > $R/$O/pmc_${C}_b8.log 2>&1
done
python $R/scripts/pmc_summary.py $R/$O/pmc_decode_raw_b8.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $R/$O/pmc_decode_summary_b8.log 2>&1
echo '{}' > $R/$O/empty.json
python $R/scripts/pmc_r2_report.py $R/$O/pmc_decode_raw_b8.json $R/$O/empty.json $R/$O r05_b8
python - <<PY
import json
d = json.load(open("$R/$O/r05_b8_pmc_decode_traffic.json"))
print(d["hbm_bytes_per_launch"]); print({k: v["hbm_bytes_per_launch"] for k, v in d["per_kernel"].items()})
PY

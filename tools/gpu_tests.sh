set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp; cd /tmp
for t in run_iss run_knn run_n3 run_n4 run_p2plane run_c4; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_rows/$t --output-format csv -- python3 $R/tools/$t.py > $R/gpurun_out/prof_rows_$t.log 2>&1; echo "$t rc=$?"
done

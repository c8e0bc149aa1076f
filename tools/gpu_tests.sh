set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -3 || exit 1
timeout -k 10 200 python tools/run_hw9.py 4000 800 0 || exit 1
timeout -k 10 200 python tools/run_hw9.py 120000 40 0 || exit 1
export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_hw9x --output-format csv -- python3 $R/tools/run_hw9.py 4000 200 0 > $R/gpurun_out/prof_hw9x.log 2>&1; echo rc=$?

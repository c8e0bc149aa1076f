set -o pipefail
for t in "" "prof=0" "" "prof=0"; do PCR_TUNE=$t timeout -k 10 200 python tools/run_hw9.py 4000 800 0 || exit 1; done
for t in "" "prof=0"; do PCR_TUNE=$t timeout -k 10 200 python tools/run_hw9.py 120000 40 0 || exit 1; done
for t in "" "prof=0"; do PCR_TUNE=$t timeout -k 10 200 python tools/run_hw9.py 120000 40 1 || exit 1; done

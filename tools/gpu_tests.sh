set -o pipefail
for t in "nn1_etrack_blocks=32768" "nn1_etrack_blocks=16384" "nn1_etrack_blocks=8192" "nn1_etrack_blocks=4096" "nn1_etrack_blocks=8192,nn1_etrack_qpl=2" "nn1_etrack_blocks=16384,nn1_etrack_qpl=2" "nn1_etrack_blocks=32768"; do PCR_TUNE=$t timeout -k 10 200 python tools/run_nn1.py 120000 10 4 | grep "n=120000" | cut -c60-140 | sed "s/^/$t /"; done

# A/B arms of the 10 M search, one gpurun call: tools/c5_ab.sh "<tunes of arm 1>" "<tunes of arm 2>" ...   (each arm: KEY=VALUE words for run_c5_iters.py)
# env LIBS="a.so b.so" runs every arm with each library (PCR_LIB_PATH); output -> gpurun_out/c5_ab.txt
mkdir -p gpurun_out; : > gpurun_out/c5_ab.txt
for arm in "$@"; do
  for lib in ${LIBS:-default}; do
    echo "=== arm: $arm   lib: $lib" >> gpurun_out/c5_ab.txt
    if [ "$lib" = default ]; then STATS=1 python tools/run_c5_iters.py ${N:-10000000} ${ITERS:-20} $arm >> gpurun_out/c5_ab.txt 2>&1
    else PCR_LIB_PATH=$PWD/$lib STATS=1 python tools/run_c5_iters.py ${N:-10000000} ${ITERS:-20} $arm >> gpurun_out/c5_ab.txt 2>&1; fi
  done
done
grep -v "^ICP\|^tunes" gpurun_out/c5_ab.txt

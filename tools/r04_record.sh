# The measurement record of round 4 with the final library, one gpurun call: gpurun --timeout 1150 -- 'bash tools/r04_record.sh'  -> gpurun_out/r04_*.txt
mkdir -p gpurun_out
run() { local out=$1; shift; echo "\$ $*" >> gpurun_out/$out; timeout -k 10 400 "$@" >> gpurun_out/$out 2>&1; echo "rc=$?" >> gpurun_out/$out; echo >> gpurun_out/$out; date +%T >> gpurun_out/heartbeat.log; }
rm -f gpurun_out/r04_*.txt
run r04_strack3.txt python tools/run_sphere.py 120000
run r04_strack3.txt python tools/s2_sweep.py 120000 - nn1_sphere_qg=2 nn1_sphere_qg=4 nn1_sphere=2
run r04_strack3.txt python tools/run_sphere.py 60000
run r04_strack3.txt python tools/run_sphere.py 250000
run r04_cold.txt python tools/run_cold.py 120000
STATS=1 run r04_c5_stile.txt python tools/run_c5_iters.py 10000000 20
run r04_c5_stile.txt python tools/run_c5_iters.py 10000000 20 grid_stile_cold=2
for r in 0 3 7; do run r04_shard.txt python tools/run_shard.py 10000000 8 $r spatial; done
run r04_shard.txt python tools/run_shard.py 10000000 8 0 contiguous
run r04_shard.txt python tools/run_shard.py 10000000 2 1 spatial
run r04_plane.txt python tools/run_plane.py
grep -h "nn1 ms\|avg first\|ICP\|STRACK\|search last3" gpurun_out/r04_*.txt | cut -c1-250

set -x
D=gpurun_out/r04g; mkdir -p $D
bash tools/profile_gpu.sh r04 > $D/profile.log 2>&1; echo "profile rc $?"
tail -3 $D/profile.log
python tools/summarize_pmc.py gpurun_out/prof_r04 > $D/summary.txt 2>&1; echo "summarize rc $?"
cat gpurun_out/prof_r04/bench.json | cut -c1-3000
( OLMC_BENCH_REHEARSAL=1 timeout -k 10 400 python3 bench.py --gpus 3 --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --paths-per-gpu 1000000 > $D/rehearsal_line.json 2> $D/rehearsal.err ); echo "rehearsal rc $?"
cat $D/rehearsal_line.json | cut -c1-1500; tail -5 $D/rehearsal.err | cut -c1-300

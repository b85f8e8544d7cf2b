set -x
D=gpurun_out/r04j; mkdir -p $D
bash tools/profile_gpu.sh r04c > $D/profile.log 2>&1; echo "profile rc $?"
tail -3 $D/profile.log
python tools/summarize_pmc.py gpurun_out/prof_r04c > $D/summary.txt 2>&1; echo "summarize rc $?"
cat gpurun_out/prof_r04c/bench.json | cut -c1-3100

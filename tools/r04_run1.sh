set -x
D=gpurun_out/r04k; mkdir -p $D
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $D/pytest.log 2>&1; echo "pytest rc $?" | tee -a $D/pytest.log
tail -4 $D/pytest.log | cut -c1-300
for c in "european 10000 50" "european 100000 100" "european 1000000 252" "european 8000000 252" "american 50000 50" "american 1000000 50" "greeks14_lean 1000000 252" "asian 1000000 1024" "heston 1000000 252" "qmc 131072 252" "barrier 1000000 252"; do
  set -- $c
  echo "== $c" >> $D/ab.txt
  timeout -k 10 200 python tools/ab_libs.py tools/ab/libolmc_r04_agent_acquire.so optionslab_amd/libolmc.so --case $1 --n $2 --m $3 --rounds 5 >> $D/ab.txt 2>&1
done
cat $D/ab.txt

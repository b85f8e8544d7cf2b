set -x
D=gpurun_out/r04h; mkdir -p $D
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $D/pytest.log 2>&1; echo "pytest rc $?" | tee -a $D/pytest.log
tail -8 $D/pytest.log | cut -c1-300
for c in "qmc 131072 252" "qmc 16384 16" "qmc 4096 8" "qmc 262144 64" "qmc 524288 64" "qmc 4194304 252" "qmc_greeks8 131072 252" "qmc_greeks14 131072 252" "qmc_greeks14 16384 16" "qmc_cv 131072 252"; do
  set -- $c
  echo "== $c" >> $D/ab.txt
  timeout -k 10 200 python tools/ab_libs.py tools/ab/libolmc_r03.so optionslab_amd/libolmc.so --case $1 --n $2 --m $3 --rounds 3 >> $D/ab.txt 2>&1
done
cat $D/ab.txt
gcc -O2 -pthread -Iinclude examples/threads_from_c.c -o /tmp/threads_from_c -Loptionslab_amd -lolmc -Wl,-rpath,$PWD/optionslab_amd
for sz in "10000 50" "100000 100" "1000000 252"; do /tmp/threads_from_c $sz 1.0 >> $D/threads_c.jsonl; done
cat $D/threads_c.jsonl

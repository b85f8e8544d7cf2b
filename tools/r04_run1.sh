set -x
D=gpurun_out/r04c; mkdir -p $D
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $D/pytest.log 2>&1; echo "pytest rc $?" | tee -a $D/pytest.log
tail -5 $D/pytest.log
for c in "qmc 131072 252" "qmc 16384 16" "qmc 4194304 252" "qmc_greeks8 131072 252" "qmc_greeks14 131072 252" "qmc_greeks14 1048576 64" "qmc_cv 131072 252"; do
  set -- $c
  echo "== $c" >> $D/ab.txt
  timeout -k 10 200 python tools/ab_libs.py tools/ab/libolmc_r03.so optionslab_amd/libolmc.so --case $1 --n $2 --m $3 --rounds 3 >> $D/ab.txt 2>&1
done
cat $D/ab.txt

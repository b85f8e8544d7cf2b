set -x
D=gpurun_out/r04e; mkdir -p $D
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $D/pytest.log 2>&1; echo "pytest rc $?" | tee -a $D/pytest.log
tail -15 $D/pytest.log | cut -c1-300

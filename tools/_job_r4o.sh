mkdir -p gpurun_out/r4o; O=gpurun_out/r4o
python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "upconv or conv_gemm or attention" 2>&1 | tail -15 > $O/pytest_kernels.txt
tail -3 $O/pytest_kernels.txt
python -m pytest tests/test_configs_gpu.py -x -q -m gpu -s 2>&1 | grep -E "passed|failed|Error|error|parity|c1_|smoke|reference_windowing|assert" | head -40 > $O/pytest_configs.txt
cat $O/pytest_configs.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > $O/smoke.txt; cat $O/smoke.txt
python tools/ab_schedule.py 3 2s 2s > $O/ab.txt 2>&1; grep -E "s/chunk|Error|error" $O/ab.txt

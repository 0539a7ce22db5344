# full GPU suite with the parity report and its wall time (the driver's step limit is 1200 s; the round's own bar: <= 900 s)
O=gpurun_out/r6_gpu_tests; mkdir -p $O; rm -f $O/parity.txt
S=$(date +%s)
VV_PARITY_REPORT=$O/parity.txt python -m pytest tests -q -m gpu -x --durations=15 2>&1 | tail -40 > $O/pytest.txt; cat $O/pytest.txt
echo "suite wall time: $(( $(date +%s) - S )) s" | tee -a $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/parity.txt

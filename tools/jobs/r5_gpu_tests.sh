# full GPU suite with the parity report
O=gpurun_out/r5_gpu_tests; mkdir -p $O; rm -f $O/parity.txt
VV_PARITY_REPORT=$O/parity.txt python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $O/pytest.txt; cat $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/parity.txt

mkdir -p gpurun_out/r4r; O=gpurun_out/r4r
bash tools/power_trace.sh $O/power_2s.txt python tools/ab_schedule.py 2 2s > $O/ab2s.txt 2>&1; grep -E "s/chunk|power trace" $O/ab2s.txt
bash tools/power_trace.sh $O/power_1s.txt python tools/ab_schedule.py 2 1s > $O/ab1s.txt 2>&1; grep -E "s/chunk|power trace" $O/ab1s.txt
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > $O/pytest.txt; cat $O/pytest.txt

# Round 6: the block front with the HEADS weight layout (each wave of a pair owns four whole heads of q / k / v) + one store burst per q / k / v (ab/front_burst.so = the tree with
# -DVV_FRONT_BURST) against the product (block-by-block stores, BLOCKS layout): correctness, steady-loop timing, WRITE_SIZE of both; one box, interleaved.
O=gpurun_out/r6_front_heads; mkdir -p $O
echo -n "front_burst correctness: " | tee $O/pytest.txt; VV_LIB_PATH=videovanish_amd/csrc/ab/front_burst.so python tools/pytest_with_lib.py tests/test_chain_gpu.py -m gpu -x -q 2>&1 | tail -1 | tee -a $O/pytest.txt
for r in 1 2 3; do
  for v in tree front_burst; do
    L=videovanish_amd/csrc/ab/$v.so; [ $v == tree ] && L=videovanish_amd/csrc/libvvhip.so
    echo -n "round $r $v: "; VV_LIB_PATH=$L python tools/bench_chain.py fp16 2>&1 | grep -E "\(front\)" | sed 's/fp16 spatial chain front level 0//; s/of the MFMA peak//' | tr '\n' ' '; echo
  done
done | tee $O/front_ab.txt
export TMPDIR=/tmp; ROOT=$(pwd); cd /tmp
for v in tree front_burst; do
  L=$ROOT/videovanish_amd/csrc/ab/$v.so; [ $v == tree ] && L=$ROOT/videovanish_amd/csrc/libvvhip.so
  export VV_LIB_PATH=$L; W=/tmp/pmcf_$v; rm -rf $W
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $W -o w -- python3 $ROOT/tools/bench_chain.py fp16 > /dev/null 2>&1
  echo "== $v"; python3 $ROOT/tools/pmc_sum.py $W chain_front | head -4
done | tee $ROOT/$O/front_write_size.txt

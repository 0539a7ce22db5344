bash tools/jobs/r5_chain_ab.sh $1 1
bash tools/jobs/r5_pmc_chain_lib.sh libvvhip.so $1 | grep "SQ_\|GRBM"

cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_a; mkdir -p $O
B="--no-cpu-baseline --no-extra-legs --docs 1250000 --fields 16 --dtype bf16 --steps 8 --warmup 2"
MFAR_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_exp/mt/libmfar_hip.so timeout -k 10 300 python bench.py $B > $O/mt.json 2> $O/mt.err
grep "^TR" $O/mt.err > $O/trace.txt; wc -l $O/trace.txt

cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_a; mkdir -p $O
A="--dtype bf16 --docs 1250000 --fields 16 --queries 128 --iters 10"
: > $O/exp.txt
echo "default" >> $O/exp.txt; timeout -k 10 200 python tools/s1_bench.py $A >> $O/exp.txt 2>&1
echo "default dbg=1" >> $O/exp.txt; MFAR_S1_DEBUG=1 timeout -k 10 200 python tools/s1_bench.py $A >> $O/exp.txt 2>&1
for e in 1 2 4 3; do
echo "exp $e dbg=1" >> $O/exp.txt; MFAR_S1_DEBUG=1 MFAR_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_exp/e$e/libmfar_hip.so timeout -k 10 200 python tools/s1_bench.py $A >> $O/exp.txt 2>&1
done
echo "64-col certified bf16s" >> $O/exp.txt; timeout -k 10 200 python tools/s1_bench.py --dtype bf16 --docs 1250000 --fields 16 --queries 64 --iters 10 >> $O/exp.txt 2>&1
echo "plain bf16r" >> $O/exp.txt; timeout -k 10 200 python tools/s1_bench.py --dtype bf16 --docs 1250000 --fields 16 --queries 64 --iters 10 --screen 0 >> $O/exp.txt 2>&1
grep -v amdgpu.ids $O/exp.txt | cut -c1-200

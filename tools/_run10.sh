cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_a; mkdir -p $O
export MFAR_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_exp/mt/libmfar_hip.so
timeout -k 10 300 python tools/trace_run.py --docs 1250000 --fields 16 --dtype bf16 --out $O/trace_bf16.npz 2>&1 | grep -v amdgpu.ids > $O/trace_bf16.txt
MFAR_S1_DYN=0 timeout -k 10 300 python tools/trace_run.py --docs 1250000 --fields 16 --dtype bf16 --out $O/trace_bf16_static.npz 2>&1 | grep -v amdgpu.ids > $O/trace_bf16_static.txt
cat $O/trace_bf16.txt

cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_a; mkdir -p $O
B="--no-cpu-baseline --no-extra-legs --docs 1250000 --fields 16 --dtype bf16 --steps 8 --warmup 2"
MFAR_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_exp/mt/libmfar_hip.so timeout -k 10 300 python bench.py $B > $O/mt.json 2> $O/mt.err
grep mixtrace $O/mt.err | tail -300 | awk '{print $3, $5, $7}' > $O/mt.txt
python - <<'PY'
import numpy as np
a=np.loadtxt('gpurun_out/r04_a/mt.txt')
print(a.shape)
# group by launches: 128 blocks each
n=len(a)//128
for L in range(max(0,n-2), n):
    b=a[L*128:(L+1)*128]
    t0=b[:,1].min()
    print('launch',L,'start spread us %.1f'%((b[:,1].max()-t0)/100), 'dur us min/med/max %.1f %.1f %.1f'%(b[:,2].min(), np.median(b[:,2]), b[:,2].max()))
PY

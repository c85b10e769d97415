cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_a; mkdir -p $O
B="--no-cpu-baseline --no-extra-legs --docs 1250000 --fields 16 --dtype bf16"
show() { python -c "
import sys,json
d=json.loads(open('$1').read()); r=d['roofline']
print('$2', 'q/s=%.0f'%d['value'], 'ms/step=%.3f'%d['ms_per_step'], r['kernel'], 'launch_ms=%.3f'%r['avg_launch_ms'], 'frac=%.3f'%r['frac'])
"; }
MFAR_PIPE_DEPTH=2 timeout -k 10 300 python bench.py $B > $O/x1.json 2>/dev/null; show $O/x1.json depth2
MFAR_PIPE_DEPTH=4 timeout -k 10 300 python bench.py $B > $O/x2.json 2>/dev/null; show $O/x2.json depth4
MFAR_TAIL_PRIORITY=1 timeout -k 10 300 python bench.py $B > $O/x3.json 2>/dev/null; show $O/x3.json tailprio
timeout -k 10 300 python bench.py $B --coalesce 1 > $O/x4.json 2>/dev/null; show $O/x4.json coalesce1
timeout -k 10 300 python bench.py $B --wgs-per-cu 1 > $O/x5.json 2>/dev/null; show $O/x5.json wgs1
timeout -k 10 300 python bench.py $B --wgs-per-cu 3 > $O/x6.json 2>/dev/null; show $O/x6.json wgs3

cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_a; mkdir -p $O
show() { python -c "
import sys,json
d=json.loads(open('$1').read()); r=d['roofline']
print('$2', 'q/s=%.0f'%d['value'], 'ms/step=%.3f'%d['ms_per_step'], r['kernel'], 'launch_ms=%.3f'%r['avg_launch_ms'], 'frac=%.3f'%r['frac'])
"; }
B="--no-cpu-baseline --no-extra-legs"
MFAR_PIPE_SERIAL=1 timeout -k 10 300 python bench.py $B --docs 1250000 --fields 16 --dtype bf16 > $O/y1.json 2>/dev/null; show $O/y1.json bf16share_serial
MFAR_PIPE_SERIAL=1 timeout -k 10 300 python bench.py $B > $O/y2.json 2>/dev/null; show $O/y2.json headline_serial
timeout -k 10 300 python bench.py $B > $O/y3.json 2>/dev/null; show $O/y3.json headline_default
MFAR_PIPE_SERIAL=1 timeout -k 10 300 python bench.py $B --docs 129375 --fields 22 > $O/y4.json 2>/dev/null; show $O/y4.json prime_serial
timeout -k 10 300 python bench.py $B --docs 129375 --fields 22 > $O/y5.json 2>/dev/null; show $O/y5.json prime_default

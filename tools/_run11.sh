cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_a; mkdir -p $O
show() { python -c "
import sys,json
d=json.loads(open('$1').read()); r=d['roofline']
print('$2', 'q/s=%.0f'%d['value'], 'ms/step=%.3f'%d['ms_per_step'], r['kernel'], 'launch_ms=%.3f'%r['avg_launch_ms'], 'frac=%.3f'%r['frac'], 'redone', d['screen']['lists_redone_exactly'] if d.get('screen') else None)
"; }
B="--no-cpu-baseline --no-extra-legs --docs 1250000 --fields 16 --dtype bf16"
MFAR_BF16W_RING=4 timeout -k 10 300 python bench.py $B > $O/w1.json 2>/dev/null; show $O/w1.json ring4_scap16_dyn
MFAR_BF16W_RING=4 MFAR_S1_DYN=0 timeout -k 10 300 python bench.py $B > $O/w2.json 2>/dev/null; show $O/w2.json ring4_scap16_static
timeout -k 10 300 python bench.py $B > $O/w3.json 2>/dev/null; show $O/w3.json ring6_scap16_dyn

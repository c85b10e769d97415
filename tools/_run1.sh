set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_a; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bf16" > $O/pytest_bf16.log 2>&1; echo "pytest rc=$?" >> $O/pytest_bf16.log
tail -5 $O/pytest_bf16.log
B="--no-cpu-baseline --no-extra-legs"
timeout -k 10 300 python bench.py $B --docs 1250000 --fields 16 --dtype bf16 > $O/b_bf16_share_default.json 2> $O/b1.err; tail -c 300 $O/b1.err
MFAR_BF16W_RING=6 timeout -k 10 300 python bench.py $B --docs 1250000 --fields 16 --dtype bf16 > $O/b_bf16_share_ring6.json 2> $O/b2.err
timeout -k 10 300 python bench.py $B --docs 1250000 --fields 16 --dtype bf16 --screen off > $O/b_bf16_share_off.json 2> $O/b3.err
timeout -k 10 300 python bench.py $B --dtype bf16 > $O/b_bf16_1m8.json 2> $O/b4.err
for f in $O/b_*.json; do python -c "
import sys,json
d=json.loads(open('$f').read()); r=d['roofline']
print('$f', 'q/s=%.0f'%d['value'], 'ms/step=%.3f'%d['ms_per_step'], r['kernel'], 'launch_ms=%.3f'%r['avg_launch_ms'], 'frac=%.3f'%r['frac'], d.get('screen'), 'recall=%.3f'%d['recall_at_20'])
"; done

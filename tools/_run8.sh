cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_a; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bf16" > $O/pytest_bf16.log 2>&1; echo "pytest rc=$?" >> $O/pytest_bf16.log
tail -3 $O/pytest_bf16.log
show() { python -c "
import sys,json
d=json.loads(open('$1').read()); r=d['roofline']
print('$2', 'q/s=%.0f'%d['value'], 'ms/step=%.3f'%d['ms_per_step'], r['kernel'], 'launch_ms=%.3f'%r['avg_launch_ms'], 'frac=%.3f'%r['frac'], 'redone', d['screen']['lists_redone_exactly'] if d.get('screen') else None)
"; }
B="--no-cpu-baseline --no-extra-legs --docs 1250000 --fields 16 --dtype bf16"
timeout -k 10 300 python bench.py $B > $O/z1.json 2>$O/z1.err; show $O/z1.json dyn_U2
MFAR_UNIT_TILES=4 timeout -k 10 300 python bench.py $B > $O/z2.json 2>/dev/null; show $O/z2.json dyn_U4
MFAR_UNIT_TILES=8 timeout -k 10 300 python bench.py $B > $O/z3.json 2>/dev/null; show $O/z3.json dyn_U8
MFAR_S1_DYN=0 timeout -k 10 300 python bench.py $B > $O/z4.json 2>/dev/null; show $O/z4.json static
MFAR_PIPE_SERIAL=1 timeout -k 10 300 python bench.py $B > $O/z5.json 2>/dev/null; show $O/z5.json dyn_serial
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs --dtype bf16 > $O/z6.json 2>/dev/null; show $O/z6.json bf16_1m8_dyn
timeout -k 10 200 python tools/s1_bench.py --dtype bf16 --docs 1250000 --fields 16 --queries 128 --iters 10 2>&1 | grep queries | cut -c1-120

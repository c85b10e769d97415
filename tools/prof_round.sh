# on the GPU box: rocprofv3 evidence for the default bench -> gpurun_out/prof_round/   (copy what is to be judged into profiles/)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_round; mkdir -p $O
B="--no-cpu-baseline --no-extra-legs"
python $R/bench.py > $O/bench.json 2> $O/bench.err
python $R/bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json 2>> $O/bench.err
python $R/bench.py --dtype bf16 --screen on --no-cpu-baseline --no-extra-legs > $O/bench_bf16_screened.json 2>> $O/bench.err
python $R/bench.py --coalesce 1 $B > $O/bench_coalesce1.json 2>> $O/bench.err
# BASELINE.json's config shapes and other embedding widths (one line each)
: > $O/shapes.txt
for cfg in "129375 22 768" "700244 5 768" "957192 8 768" "125000 8 768" "2000000 8 384" "1500000 8 512" "750000 8 1024" "500000 8 1536"; do
  set -- $cfg
  python $R/bench.py $B --docs $1 --fields $2 --dim $3 2>> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['config']['docs'], d['config']['fields'], d['config']['dim'], 'q/s=%.0f' % d['value'], 'ms/step=%.3f' % d['ms_per_step'], r['kernel'], 'launch_ms=%.3f' % r['avg_launch_ms'], 'hbm_frac=%.3f' % r['frac'], 'redone=%d' % d['screen']['lists_redone_exactly'], 'recall20=%.3f' % d['recall_at_20'])" >> $O/shapes.txt
done
# BASELINE.json configs[4]'s per-GPU share (10 M x 16 bf16 over 8 GPUs = 1.25 M rows per rank), exact bf16 pass and opt-in screen
for scr in auto on; do
  python $R/bench.py $B --docs 1250000 --fields 16 --dtype bf16 --screen $scr 2>> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['config']['docs'], d['config']['fields'], d['config']['dim'], 'bf16 screen=$scr', 'q/s=%.0f' % d['value'], 'ms/step=%.3f' % d['ms_per_step'], r['kernel'], 'launch_ms=%.3f' % r['avg_launch_ms'], 'hbm_frac=%.3f' % r['frac'], 'recall20=%.3f' % d['recall_at_20'])" >> $O/shapes.txt
done
# the STaRK-prime shape with sparse fields (most documents lack most of its 22 fields) and with the structured field kinds
for extra in "--empty-frac 0.7" "--empty-frac 0.9" "--corpus structured"; do
  python $R/bench.py $B --docs 129375 --fields 22 $extra 2>> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['config']['docs'], d['config']['fields'], d['config']['dim'], '$extra', 'q/s=%.0f' % d['value'], 'ms/step=%.3f' % d['ms_per_step'], r['kernel'], 'launch_ms=%.3f' % r['avg_launch_ms'], 'redone=%d' % d['screen']['lists_redone_exactly'], 'recall20=%.3f' % d['recall_at_20'])" >> $O/shapes.txt
done
# kernel trace + stats of the default leg alone (the extra legs launch the same kernels on other shapes and would skew the averages)
rocprofv3 --kernel-trace --stats -d /tmp/kt -o r1 --output-format csv -- python $R/bench.py --steps 16 --warmup 2 $B > $O/bench_under_rocprof.json 2>/dev/null
# counters: one pass each; default leg, then the exact fp32 pass (--screen off) into a sub-directory of the same pass
for c in "pf FETCH_SIZE" "pw WRITE_SIZE" "ps SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  set -- $c; d=$1; shift
  rocprofv3 --kernel-trace --pmc $* -d /tmp/$d -o r1 --output-format csv -- python $R/bench.py --steps 4 --warmup 1 $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $* -d /tmp/$d/off -o r1 --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --screen off $B > /dev/null 2>&1
done
# stage-2 bytes per launch at the 22-field shape, certified two-level stage 2 vs full gather (one FETCH_SIZE pass each), and its kernel trace
for m in auto full; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/s2_$m -o r1 --output-format csv -- python $R/bench.py --steps 4 --warmup 1 $B --docs 129375 --fields 22 --stage2 $m > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats -d /tmp/ktp -o r1 --output-format csv -- python $R/bench.py --steps 16 --warmup 2 $B --docs 129375 --fields 22 > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/s2_auto /tmp/s2_full /tmp/ktp > $O/stage2_traffic_prime.txt
python $R/tools/prof_summary.py /tmp/kt /tmp/pf /tmp/pw /tmp/ps > $O/rocprofv3_summary.txt
python $R/tools/prof_summary.py --counters-json $O/counters.json /tmp/pf /tmp/pw /tmp/ps $(cd $R && python -c "import bench; print(bench.source_hash())") 1000000 8 768 64 1
python $R/tools/trace_timeline.py /tmp/kt > $O/timeline.txt 2>/dev/null
python $R/tools/trace_timeline.py /tmp/ktp > $O/timeline_prime.txt 2>/dev/null
# what v_mfma_f32_32x32x2_f32 sustains with constant / fresh / LDS / streamed operands
hipcc -O3 --offload-arch=gfx950 $R/tools/probes/mfma_f32_clock_probe.hip -o /tmp/mp 2>/dev/null && timeout -k 10 240 /tmp/mp > $O/mfma_f32_clock_probe.txt 2>&1
tail -c 600 $O/bench.json

# on the GPU box: rocprofv3 evidence for the default bench -> gpurun_out/prof_round/   (copy what is to be judged into profiles/)
#   bash tools/prof_round.sh            (~15 GPU-minutes)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_round; mkdir -p $O
B="--no-cpu-baseline --no-extra-legs"
# the driver-shaped run (every leg, BASELINE config legs included) and the default 64-step run
python $R/bench.py --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench.err      # the driver's command, every leg
python $R/bench.py --no-cpu-baseline --no-config-legs > $O/bench.json 2>> $O/bench.err
python $R/bench.py --no-encode-leg > $O/bench_default.json 2>> $O/bench.err        # `python bench.py` (64 steps) without the 60 s encode leg
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print(c['docs'], c['fields'], c['dim'], '$1', 'q/s=%.0f' % d['value'], 'ms/step=%.3f' % d['ms_per_step'], r['kernel'], 'launch_ms=%.3f' % r['avg_launch_ms'], 'hbm_frac=%.3f' % r['frac'],
      'redone=%s' % (d['screen']['lists_redone_exactly'] if d.get('screen') else '-'), 'tier2=%s/%s' % (d['adaptive']['tier2']['lists'], d['adaptive']['tier2']['passed_on_to_exact']),
      'off=%s' % d['adaptive']['off'], 'recall20=%.3f' % d['recall_at_20'], 'resident=%.2f' % d['resident_bytes']['ratio'])"; }
# BASELINE.json's config shapes and other embedding widths (one line each)
: > $O/shapes.txt
for cfg in "129375 22 768" "700244 5 768" "957192 8 768" "125000 8 768" "2000000 8 384" "1500000 8 512" "750000 8 1024" "500000 8 1536"; do
  set -- $cfg
  python $R/bench.py $B --docs $1 --fields $2 --dim $3 2>> $O/bench.err | line "" >> $O/shapes.txt
done
# bf16 indexes: BASELINE.json configs[4]'s per-GPU share (10 M x 16 over 8 GPUs = 1.25 M rows per rank) and the headline shape; default = the
# certified pass over the slab itself, off = the plain three-term pass
for scr in auto off; do
  python $R/bench.py $B --docs 1250000 --fields 16 --dtype bf16 --screen $scr 2>> $O/bench.err | line "bf16 screen=$scr" >> $O/shapes.txt
  python $R/bench.py $B --dtype bf16 --screen $scr 2>> $O/bench.err | line "bf16 screen=$scr" >> $O/shapes.txt
done
# the STaRK-prime shape with sparse fields (most documents lack most of its 22 fields), the structured field kinds, no score dump
for extra in "--empty-frac 0.7" "--empty-frac 0.9" "--corpus structured" "--corpus clustered"; do
  python $R/bench.py $B --docs 129375 --fields 22 $extra 2>> $O/bench.err | line "$extra" >> $O/shapes.txt
done
MFAR_S2_DUMP=0 python $R/bench.py $B --docs 129375 --fields 22 2>> $O/bench.err | line "MFAR_S2_DUMP=0" >> $O/shapes.txt
python $R/bench.py $B --screen off 2>> $O/bench.err | line "--screen off" >> $O/shapes.txt
python $R/bench.py $B --coalesce 1 2>> $O/bench.err | line "--coalesce 1" >> $O/shapes.txt
python $R/bench.py $B --pipeline python 2>> $O/bench.err | line "--pipeline python" >> $O/shapes.txt
python $R/bench.py $B --corpus clustered --steps 256 2>> $O/bench.err | line "--corpus clustered" >> $O/shapes.txt
python $R/bench.py $B --corpus clustered --steps 256 --screen off 2>> $O/bench.err | line "--corpus clustered --screen off" >> $O/shapes.txt
# round 6: tier 2 off (the round-5 policy alone: fields switched off -> exact pass), moderate cluster noise, the deep scan (opt-in), a narrow cone,
# the kernels of rounds 3-5 in the tail
MFAR_SCREEN_TIER2=0 python $R/bench.py $B --corpus clustered --steps 256 2>> $O/bench.err | line "--corpus clustered MFAR_SCREEN_TIER2=0" >> $O/shapes.txt
python $R/bench.py $B --corpus clustered --cluster-noise 1e-2 --steps 256 2>> $O/bench.err | line "--corpus clustered --cluster-noise 1e-2" >> $O/shapes.txt
MFAR_SCREEN_DEEP=1 python $R/bench.py $B --corpus clustered --steps 256 2>> $O/bench.err | line "--corpus clustered MFAR_SCREEN_DEEP=1" >> $O/shapes.txt
MFAR_SCREEN_DEEP=2 MFAR_SCREEN_TIER2=2 python $R/bench.py $B --steps 256 2>> $O/bench.err | line "MFAR_SCREEN_DEEP=2 (every field, plain corpus)" >> $O/shapes.txt
python $R/bench.py $B --mu-scale 2.7 --steps 256 2>> $O/bench.err | line "--mu-scale 2.7 (narrow cone, isotropic spread)" >> $O/shapes.txt
MFAR_S2_FUSED=0 python $R/bench.py $B --steps 256 2>> $O/bench.err | line "MFAR_S2_FUSED=0" >> $O/shapes.txt
MFAR_S2_FUSED=0 python $R/bench.py $B --docs 129375 --fields 22 --steps 256 2>> $O/bench.err | line "MFAR_S2_FUSED=0" >> $O/shapes.txt
python $R/bench.py $B --docs 129375 --fields 22 --steps 256 2>> $O/bench.err | line "(256 steps)" >> $O/shapes.txt
# kernel trace + stats of the default leg alone (the extra legs launch the same kernels on other shapes and would skew the averages)
rocprofv3 --kernel-trace --stats -d /tmp/kt -o r1 --output-format csv -- python $R/bench.py --steps 16 --warmup 2 $B > $O/bench_under_rocprof.json 2>/dev/null
# counters: one pass each; default leg, then the exact fp32 pass (--screen off) into a sub-directory of the same pass
for c in "pf FETCH_SIZE" "pw WRITE_SIZE" "ps SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  set -- $c; d=$1; shift
  rocprofv3 --kernel-trace --pmc $* -d /tmp/$d -o r1 --output-format csv -- python $R/bench.py --steps 4 --warmup 1 $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc $* -d /tmp/$d/off -o r1 --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --screen off $B > /dev/null 2>&1
done
python $R/tools/prof_summary.py /tmp/kt /tmp/pf /tmp/pw /tmp/ps > $O/rocprofv3_summary.txt
python $R/tools/prof_summary.py --counters-json $O/counters.json /tmp/pf /tmp/pw /tmp/ps $(cd $R && python -c "import bench; print(bench.source_hash())") 1000000 8 768 64 1
python $R/tools/trace_timeline.py /tmp/kt > $O/timeline.txt 2>/dev/null
# per shape: kernel trace + FETCH_SIZE / WRITE_SIZE passes of BASELINE configs[1], [2] and the configs[4] share (HBM bytes per launch of every kernel)
for cfg in "prime 129375 22 f32" "mag 700244 5 f32" "bf16share 1250000 16 bf16"; do
  set -- $cfg
  A="--steps 24 --warmup 4 $B --docs $2 --fields $3 --dtype $4"
  rocprofv3 --kernel-trace --stats -d /tmp/kt_$1 -o r1 --output-format csv -- python $R/bench.py $A > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf_$1 -o r1 --output-format csv -- python $R/bench.py $A > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw_$1 -o r1 --output-format csv -- python $R/bench.py $A > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/kt_$1 /tmp/pf_$1 /tmp/pw_$1 > $O/traffic_$1.txt
  python $R/tools/trace_timeline.py /tmp/kt_$1 > $O/timeline_$1.txt 2>/dev/null
done
python $R/tools/build_timing.py > $O/build_timing.txt 2>/dev/null
python $R/tools/build_timing.py --dtype bf16 --docs 1250000 --fields 16 >> $O/build_timing.txt 2>/dev/null
tail -c 600 $O/bench.json
cat $O/shapes.txt
# round 6: the kept encoder run at the STaRK-prime row count (>= 129 375 x 22), bf16 autocast rows
python $R/tools/encode_bench.py --docs 129375 --modes bf16 --no-sweep > $O/encode_prime_129k.json 2>> $O/bench.err

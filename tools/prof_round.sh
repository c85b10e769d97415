# on the GPU box: rocprofv3 evidence for the screened (default) bench -> gpurun_out/prof_round/
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_round; mkdir -p $O
python $R/bench.py > $O/bench.json 2> $O/bench.err
python $R/bench.py --screen off --no-cpu-baseline > $O/bench_screen_off.json 2>> $O/bench.err
python $R/bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats -d /tmp/kt -o r1 --output-format csv -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o r1 --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o r1 --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT -d /tmp/ps -o r1 --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/kt /tmp/pf /tmp/pw /tmp/ps > $O/rocprofv3_summary.txt
python $R/tools/prof_summary.py --traffic-json $O/stage1_f16_traffic.json /tmp/pf /tmp/pw mfar_stage1_f16r_kernel
python $R/tools/trace_timeline.py /tmp/kt > $O/timeline.txt 2>/dev/null
# the bf16 slab: traffic of its scan kernel
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/bf -o r1 --output-format csv -- python $R/bench.py --dtype bf16 --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/bw -o r1 --output-format csv -- python $R/bench.py --dtype bf16 --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/prof_summary.py --traffic-json $O/stage1_bf16_traffic.json /tmp/bf /tmp/bw mfar_stage1_bf16r_kernel
tail -c 600 $O/bench.json

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_a; mkdir -p $O
B="--no-cpu-baseline --no-extra-legs"
rocprofv3 --kernel-trace --stats -d /tmp/kt -o r1 --output-format csv -- python $R/bench.py --steps 16 --warmup 2 $B --docs 1250000 --fields 16 --dtype bf16 > $O/prof_bench.json 2>/dev/null
python $R/tools/prof_summary.py /tmp/kt > $O/prof_bf16_share.txt 2>&1
python $R/tools/trace_timeline.py /tmp/kt > $O/timeline_bf16_share.txt 2>/dev/null
head -40 $O/prof_bf16_share.txt

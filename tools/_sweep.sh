python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -o r1 --output-format csv -- python $GRAFT_REPO_ROOT/tools/s1_bench.py > /tmp/s1.log 2>&1
tail -1 /tmp/s1.log
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/kt | grep -E "avg_us" | grep -E "merge|stage1|score|certify|sample|queries"
python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('1M', round(j['value']), round(j['ms_per_step'],3), j['roofline']['avg_launch_ms'])"
python $GRAFT_REPO_ROOT/bench.py --docs 125000 --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('125k', round(j['value']), round(j['ms_per_step'],3), j['roofline']['avg_launch_ms'])"

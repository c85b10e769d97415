L=multifield-adaptive-retrieval_amd/lib
cp $L/var_st6/libmfar_hip.so $L/libmfar_hip.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -o r1 --output-format csv -- python $GRAFT_REPO_ROOT/tools/s1_bench.py > /tmp/s1.log 2>&1
tail -1 /tmp/s1.log
python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/kt | grep -v "^==" | head -70
python $GRAFT_REPO_ROOT/tools/trace_timeline.py /tmp/kt 2>/dev/null | tail -40

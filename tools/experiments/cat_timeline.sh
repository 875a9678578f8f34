cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cp; CAT_BENCH_OPTS=nm_cat_one=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/cp -o cp -- python $GRAFT_REPO_ROOT/tools/categorical_bench.py 5000 2>/dev/null | tail -1 | cut -c1-300
python $GRAFT_REPO_ROOT/tools/experiments/cat_timeline.py /tmp/cp 30

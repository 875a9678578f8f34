cd /tmp && export TMPDIR=/tmp
for o in nm_cat_one=0 nm_cat_one=1; do
rm -rf /tmp/cp; CAT_BENCH_OPTS=$o CAT_BENCH_STEPS=6 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/cp -o cp -- python $GRAFT_REPO_ROOT/tools/categorical_bench.py ${1:-5000} > /dev/null 2>&1; echo "== $o"; python $GRAFT_REPO_ROOT/tools/kernel_table.py /tmp/cp 2>&1 | head -10
done

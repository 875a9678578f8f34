cd /tmp && export TMPDIR=/tmp
for o in nm_subset=0 nm_subset=4; do
rm -rf /tmp/cp; CAT_BENCH_OPTS=$o CAT_BENCH_STEPS=10 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/cp -o cp -- python $GRAFT_REPO_ROOT/tools/categorical_bench.py ${1:-1000} > /dev/null 2>&1; echo "== $o"; python $GRAFT_REPO_ROOT/tools/kernel_table.py /tmp/cp 2>&1 | head -9
done

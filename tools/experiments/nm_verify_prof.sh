cd /tmp && export TMPDIR=/tmp
for o in "" "nm_verify_rows=1" "nm_verify_rows=100"; do
rm -rf /tmp/cp; NM_BENCH_OPTS=$o timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/cp -o cp -- python $GRAFT_REPO_ROOT/tools/nonmetric_bench.py 2>/dev/null | tail -1 | cut -c150-420; echo "== $o"; python $GRAFT_REPO_ROOT/tools/kernel_table.py /tmp/cp 2>&1 | grep -E "nm_conv_dense|nm_v|nmwave"
done

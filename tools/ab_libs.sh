# A/B of two builds of the library on one box: tools/ab_libs.sh <old.so> <command...>   (the command reads PLSPM_HIP_LIB)
OLD=$1; shift
for r in 1 2; do
  echo "== lib $OLD"; PLSPM_HIP_LIB=$OLD "$@" 2>/dev/null | tail -1 | cut -c1-600
  echo "== lib in-tree"; env -u PLSPM_HIP_LIB "$@" 2>/dev/null | tail -1 | cut -c1-600
done

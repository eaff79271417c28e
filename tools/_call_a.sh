set -u
export TMPDIR=/tmp
for r in 1 2 3; do
for v in 0 1; do echo "NO_REL=$v"; HGMM_TREE_NO_REL=$v timeout 100 python tools/c4prof.py both 6 2>&1 | tail -2; done
done

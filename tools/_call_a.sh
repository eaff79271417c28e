set -u
export TMPDIR=/tmp
for r in 1 2; do
for a in 0 1 2 3 4 6; do echo -n "ahead $a: "; HGMM_TREE_AHEAD=$a timeout 100 python tools/c4prof.py c4 8 2>&1 | grep "C4 build"; done
done

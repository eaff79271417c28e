set -u
for r in 1 2 3; do
echo -n "live: "; timeout 100 python tools/c4prof.py both 8 2>&1 | grep "build ms" | cut -c1-100
echo -n "no live: "; HGMM_TREE_NO_LIVE=1 timeout 100 python tools/c4prof.py both 8 2>&1 | grep "build ms" | cut -c1-100
done

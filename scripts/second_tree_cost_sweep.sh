# Frame time of the two-tree mode against the inner-node price its second tree is built with (diag build reads
# MRT_SECOND_TREE_COST, in tenths of a triangle test; a tree that does not fit the LDS is built again at the next price
# of the library's list).   gpurun -- 'bash scripts/second_tree_cost_sweep.sh'
cd $GRAFT_REPO_ROOT
export MRT_LIB=raytracing_amd/libmrt_diag.so
for cost in 15 30 35 40 50 60 80 120; do
  echo "== first price $cost"
  MRT_SECOND_TREE_COST=$cost MRT_TIMING=1 timeout -k 10 200 python3 scripts/two_trees_probe.py 2>&1 | grep -E "second tree at|both trees  " | grep -v against | tail -4 || exit 1
done

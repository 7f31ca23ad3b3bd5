# The moving-camera tile order (radius in thousandths of the frame's width, share, split budget in hundredths) on a scene
# with two trees, whose tiles cost less than those the rule was tuned on.   gpurun -- 'bash scripts/spread_sweep_two_trees.sh'
cd $GRAFT_REPO_ROOT
export MRT_LIB=raytracing_amd/libmrt_diag.so
for cfg in "40 75 50" "40 0 50" "20 75 50" "80 75 50" "40 100 50" "40 75 100" "40 75 25" "20 100 100"; do set -- $cfg
  for size in "1920 1080" "800 600"; do
    echo -n "permille $1 share $2 budget $3: "; MRT_SPREAD_PERMILLE=$1 MRT_SPREAD_SHARE=$2 MRT_SPLIT_BUDGET=$3 timeout -k 10 100 python3 scripts/walk_loss_probe.py $size 2>&1 | tail -1 || exit 1
  done
done

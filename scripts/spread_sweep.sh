cd $GRAFT_REPO_ROOT
for cfg in "40 0" "20 50" "40 50" "80 50" "40 75" "40 100" "80 100"; do set -- $cfg
  for size in "1920 1080" "800 600"; do
    echo -n "permille $1 share $2: "; MRT_LIB=raytracing_amd/libmrt_diag.so MRT_SPREAD_PERMILLE=$1 MRT_SPREAD_SHARE=$2 timeout -k 10 100 python3 scripts/walk_loss_probe.py $size 2>&1 | tail -1
  done
done

cd $GRAFT_REPO_ROOT
for cfg in "50 40 75" "100 40 75" "200 40 75" "100 20 90" "200 20 90" "100 40 0" "200 40 0"; do set -- $cfg
  echo -n "budget $1 permille $2 share $3: "; MRT_LIB=raytracing_amd/libmrt_diag.so MRT_SPLIT_BUDGET=$1 MRT_SPREAD_PERMILLE=$2 MRT_SPREAD_SHARE=$3 timeout -k 10 100 python3 scripts/walk_loss_probe.py 800 600 2>&1 | tail -1
done

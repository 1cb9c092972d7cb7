mkdir -p gpurun_out/r2m
export WIS_LIB_PATH=$PWD/willow-inference-server_amd/lib/libwis_hip_taps.so
for cfg in "fold defer" "nofold defer" "fold nodefer" "nofold nodefer"; do
  set -- $cfg
  unset WIS_NO_CQFOLD WIS_NO_DEFER
  [ $1 = nofold ] && export WIS_NO_CQFOLD=1
  [ $2 = nodefer ] && export WIS_NO_DEFER=1
  echo "#### $cfg"
  python tools/timeline.py large 5 10 1 2>&1 | tail -12
done > gpurun_out/r2m/timeline.txt 2>&1
cat gpurun_out/r2m/timeline.txt

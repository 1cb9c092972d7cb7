mkdir -p gpurun_out/full
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/full/gputest.log 2>&1; echo rc=$? >> gpurun_out/full/gputest.log
tail -12 gpurun_out/full/gputest.log
python __graft_entry__.py --smoke 2>&1 | tail -2

export WIS_LIB_PATH=$PWD/willow-inference-server_amd/lib/libwis_hip_taps.so
python tools/sampling_cycles.py 2>&1 | tail -4

PHK_LIB=$PWD/phlash_amd/csrc/exp/libphk_trc.so python3 scripts/dev/debug_trace.py 2>&1 | grep -v amdgpu.ids

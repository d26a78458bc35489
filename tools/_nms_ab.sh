set -u
python -m pytest tests/test_gpu_rbox.py tests/test_gpu_configs.py tests/test_gpu_center_infer.py tests/test_gpu_anchor_infer.py tests/test_gpu_pvrcnn_infer.py tests/test_gpu_center_end_to_end.py -m gpu -x -q > gpurun_out/t_rbox.log 2>&1; tail -3 gpurun_out/t_rbox.log
for v in 1 2; do python tests/perf/nms_time.py 2>&1 | grep -v amdgpu; done
GD3D_LIB=tools/variants/libgd3d_prof.so python tools/scan_profile.py 2>&1 | grep -v amdgpu

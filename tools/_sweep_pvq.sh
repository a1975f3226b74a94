export PYTHONPATH=/root/repo; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_quant_decode_gpu.py tests/test_decode_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 120 python tools/time_pvq_loop.py 3 384 65536
timeout 120 python tools/time_pvq_loop.py 4 192 131072
PALU_HIP_LIB=/root/repo/gpurun_in/lib/libpalu_hip_old.so timeout 120 python tools/time_pvq_loop.py 3 384 65536
PALU_HIP_LIB=/root/repo/gpurun_in/lib/libpalu_hip_old.so timeout 120 python tools/time_pvq_loop.py 4 192 131072
} > gpurun_out/pvq_sweep.txt 2>&1
grep -v amdgpu.ids gpurun_out/pvq_sweep.txt

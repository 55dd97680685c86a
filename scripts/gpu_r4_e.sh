#!/bin/bash
# device-noise generator + non-blocking status look: tests, same-box A/B
O=gpurun_out/r4e; mkdir -p $O
python -m pytest tests/test_noise.py tests/test_gpu_pipeline.py -x -q -m gpu -s -k "noise or aborted or window_timing or ddpm" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
python scripts/ab_check_noise.py > $O/ab.log 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids $O/ab.log | tail -14

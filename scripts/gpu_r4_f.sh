#!/bin/bash
# decoder kernels: parity + speed
O=gpurun_out/r4f; mkdir -p $O
python scripts/stress_decode.py 40 > $O/stress_decode.log 2>&1; echo "stress rc=$?"; tail -3 $O/stress_decode.log
python scripts/decode_speed.py > $O/decode_speed.log 2>&1; echo "speed rc=$?"; grep -v amdgpu.ids $O/decode_speed.log | head -3; grep "B=8" $O/decode_speed.log | head -2
python -m pytest tests/ -x -q -m gpu -k "decode or vae or golden or c5 or ragged or fusion" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
bash scripts/gpu_r4_g.sh

#!/bin/bash
# fused out_proj + cross-attention decoder kernel: parity + speed
O=gpurun_out/r4f; mkdir -p $O
python scripts/stress_decode.py 40 > $O/stress_decode.log 2>&1; echo "stress rc=$?"; tail -4 $O/stress_decode.log
python scripts/oc_stamps.py 2>&1 | grep -v amdgpu.ids | tail -6
python scripts/decode_speed.py > $O/decode_speed.log 2>&1; echo "speed rc=$?"; grep -v amdgpu.ids $O/decode_speed.log | head -3
python -m pytest tests/ -x -q -m gpu -k "decode or vae or golden or c5 or ragged" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log

#!/bin/bash
for k in test_single_launch_equals_two_launch_long test_fused_step_vs_oracle_pipeline test_random_fused_replay test_ring_history_folded test_l2_fused_step_vs_oracle test_hand_off_timeout test_fused_step_differential_fuzz; do
  echo -n "$k: "; python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_recovery.py -q -m gpu -k "$k or (co_tenant and recent_global)" 2>&1 | tail -n 1
done
echo -n "recovery all but strict: "; python -m pytest tests/test_gpu_recovery.py -q -m gpu -k "not strict" 2>&1 | tail -n 1

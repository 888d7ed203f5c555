#!/bin/bash
# end-of-round evidence: the new GPU form tests, then the round-3 evidence pass (tools/r3_profiles.sh)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_attention_forms_gpu.py -x -q -m gpu > gpurun_out/forms_tests.log 2>&1; grep "passed\|failed" gpurun_out/forms_tests.log | tail -1
bash tools/r3_profiles.sh

#!/bin/bash
# re-run one randomised configuration: fuzz_one.sh <seed-offset> <k>
WMBUS_FUZZ_SEED=$1 WMBUS_FUZZ_N=$(($2+1)) timeout 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "test_random_configuration_matches_oracle and $2-" 2>&1 | grep -v "^\s*$" | tail -25

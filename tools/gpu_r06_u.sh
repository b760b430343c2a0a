#!/bin/bash
# round 6, pass U: the optimiser tests on the device
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "diodemix or update_step or training_step" 2>&1 | tail -30

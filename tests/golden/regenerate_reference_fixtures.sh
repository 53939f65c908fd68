#!/bin/bash
# Re-create every fixture that is produced by EXECUTING reference code (needs the reference checkout at /root/reference;
# the GPU box and CI only read the committed *.pt files).  The oracle-made fixtures come from make_golden.py.
set -e
cd "$(dirname "$0")/../.."
for s in make_reference_fixtures make_reference_driver_fixtures make_reference_forward_fixture make_reference_unet_fixture \
         make_reference_train_fixture make_reference_condition_fixtures make_reference_checkpoint_fixture \
         make_reference_scheduler_fixture make_reference_train_unet_fixture; do
  echo "== $s"
  python tests/golden/$s.py | tail -2
done
python -m pytest tests/test_reference_fixtures_cpu.py -q | tail -1

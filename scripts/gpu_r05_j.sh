#!/bin/bash
# round 5, call J: the switch point of the default mapping again (joint fetches), the renderer's rules on the gallery, the shadow-order A/B
mkdir -p gpurun_out/r05; export TMPDIR=/tmp
O=gpurun_out/r05
timeout 900 python scripts/threshold_sweep.py --scenes atrium 2>&1 | grep -v amdgpu.ids > $O/threshold_sweep.txt; cat $O/threshold_sweep.txt
timeout 2400 python scripts/render_rules_check.py 2>&1 | grep -v amdgpu.ids > $O/render_rules_check.txt; cat $O/render_rules_check.txt

#!/bin/bash
# scratch: whatever is being measured right now
mkdir -p gpurun_out/r03
for ORDER in bfs area; do
echo "== image order $ORDER"
RODENT_HIP_IMAGE_ORDER=$ORDER timeout 900 python scripts/refill_sweep.py --idle 0,40 --scenes atrium,atrium/2,atrium/8,cornell --frames 5 2>&1 | tail -4 | cut -c1-75
RODENT_HIP_IMAGE_ORDER=$ORDER timeout 900 python scripts/refill_sweep.py --idle 40 --scenes atrium --size 3840x2160 --spp 32 --frames 3 2>&1 | tail -1 | cut -c1-75
done > gpurun_out/r03/image_order.txt 2>&1
cat gpurun_out/r03/image_order.txt

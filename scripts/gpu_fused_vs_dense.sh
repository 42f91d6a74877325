python -c "from rodent_amd import scenes; scenes.scene_bvh('atrium')" 
A="--scene data/atrium.obj --bench 3 --eye -1150 350 30 --dir 1 0.12 -0.05 --up 0 1 0 --width 1920 --height 1080 --spp 16 --max-path-len 8"
C="--scene tests/golden/cornell_box.obj --bench 5 --eye 0 1 2.7 --dir 0 0 -1 --up 0 1 0 --width 1920 --height 1080 --spp 64 --max-path-len 4"
for k in 1 2; do for f in 0 1; do echo "FUSED=$f atrium: $(RODENT_HIP_FUSED_SORT=$f rodent_amd/bin/rodent $A 2>&1 | tail -1)"; echo "FUSED=$f cfg4: $(RODENT_HIP_FUSED_SORT=$f rodent_amd/bin/rodent $C 2>&1 | tail -1)"; done; done

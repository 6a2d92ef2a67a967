# round 3, call 26: does the HALO form pay on the narrow maps of layer3 / layer4 (image-row ends every 24 / 12 pixels: bank conflicts in the fragment reads)?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 16"
for i in 1 2 3; do
  for mw in 0 24 48; do echo -n "minw=$mw "; LP_CONV_HALO_MINW=$mw $B 2>&1 | tail -1 | cut -c88-110; done
done > gpurun_out/r03y_halo_minw.txt 2>&1; cat gpurun_out/r03y_halo_minw.txt

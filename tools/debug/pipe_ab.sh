for nb in 32 64 128; do echo "bg blocks $nb"; SAM_UPDATE_BG_BLOCKS=$nb python tools/debug/step_stamps.py 40 2>&1 | grep -v amdgpu.ids | egrep "ms per step|adam|textbert|mmt fwd|sumsq|after"; done
echo "no pipeline"; SAM_PIPELINE_UPDATE=0 python tools/debug/step_stamps.py 40 2>&1 | grep -v amdgpu.ids | egrep "ms per step|adam|textbert|mmt fwd|sumsq|after"

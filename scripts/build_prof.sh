#!/bin/bash
# A PROFILING build of the library beside the shipping one: the named translation units (default: roi_pool.hip) compiled
# with -DD2AMD_PROFILE (d2_prof_env() reads the A/B switches), linked with the shipping objects of the others ->
# detectron2_amd/lib/libd2amd_prof.so (git-ignored, travels to the GPU box; select it with D2AMD_LIB_PATH).
#   bash scripts/build_prof.sh [unit.hip ...]
set -e
cd "$(dirname "$0")/.."
python -m detectron2_amd.build > /dev/null
UNITS=${@:-roi_pool.hip}
mkdir -p detectron2_amd/lib/obj_prof
OBJS=""
for o in detectron2_amd/lib/obj/*.o; do
  b=$(basename $o .o)
  if [[ " $UNITS " == *" $b.hip "* ]]; then
    EXTRA=""; grep -q "\"$b.hip\": \[\"-ffp-contract=off\"\]" detectron2_amd/build.py && EXTRA="-ffp-contract=off"
    /opt/rocm/bin/hipcc -c detectron2_amd/csrc/$b.hip -o detectron2_amd/lib/obj_prof/$b.o -O3 -std=c++17 -fPIC --offload-arch=gfx950 \
      -fno-gpu-rdc -w -fhip-fp32-correctly-rounded-divide-sqrt -Xclang -target-feature -Xclang -packed-fp32-ops -DD2AMD_PROFILE $EXTRA 2>&1 | grep -v "not a recognized feature" || true
    OBJS="$OBJS detectron2_amd/lib/obj_prof/$b.o"
  else
    OBJS="$OBJS $o"
  fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o detectron2_amd/lib/libd2amd_prof.so $OBJS
echo detectron2_amd/lib/libd2amd_prof.so

#!/bin/bash
# Development variant of the WIDE per-instance kernels only: recompile csrc/admm_pinst_g2.hip with extra -D flags and link it with the
# objects of the last full build:   tools/pinst_variant.sh <name> [-D flags...]  -> variants/libadmm_hip_<name>.so (ADMM_HIP_LIB selects it)
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p variants
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Xarch_device -fno-honor-nans -Xarch_device -Wno-nan-infinity-disabled -Wall \
  -Wno-unused-function -Iinclude "$@" -c admm-library_amd/csrc/admm_pinst_g2.hip -o "variants/${name}_g2.o" \
  -Rpass-analysis=kernel-resource-usage 2> "variants/${name}.build.log" || { grep error "variants/${name}.build.log" | head; exit 1; }
objs=$(ls build/obj/*.o | grep -v admm_pinst_g2.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "variants/libadmm_hip_${name}.so" $objs "variants/${name}_g2.o" && rm "variants/${name}_g2.o"
ls -la "variants/libadmm_hip_${name}.so"

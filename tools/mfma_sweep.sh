#!/bin/bash
# A/B of admm_mfma.hip variants on one GPU box: only that translation unit is recompiled (seconds), the rest of
# the library is taken from build/obj of the last full build (prepared here, on the CPU container, with --prepare).
#   tools/mfma_sweep.sh --prepare "<name>:<-D flags>" ...      (here: builds variants/libadmm_<name>.so)
#   tools/mfma_sweep.sh --run [mode] [workload]                 (on the GPU box: times every variants/*.so)
cd "$(dirname "$0")/.."
if [ "$1" == "--prepare" ]; then
  shift; mkdir -p variants; rm -f variants/*.so
  FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -Xarch_device -fno-honor-nans -Xarch_device -Wno-nan-infinity-disabled"
  for spec in "$@"; do
    name="${spec%%:*}"; flags="${spec#*:}"; [ "$flags" == "$spec" ] && flags=""
    /opt/rocm/bin/hipcc $FLAGS $flags -c admm-library_amd/csrc/admm_mfma.hip -o variants/$name.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libadmm_$name.so variants/$name.o \
      $(ls build/obj/*.o | grep -v admm_mfma) || exit 1
    rm variants/$name.o; echo "built variants/libadmm_$name.so ($flags)"
  done
else
  shift
  for lib in variants/libadmm_*.so; do
    echo "== $lib"
    ADMM_HIP_LIB="$PWD/$lib" timeout -k 10 200 python tools/mfma_time.py ${1:-2} ${2:-cw_formation} 2>&1 | tail -2
  done
fi

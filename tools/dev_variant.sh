#!/bin/bash
# Build a DEVELOPMENT variant of libadmm_hip.so here (no GPU needed) for A/B runs on the GPU box:
#   tools/dev_variant.sh <name> [-D flags...]     -> variants/libadmm_hip_<name>.so   (ADMM_HIP_LIB=... selects it)
# -DADMM_DEV_DIMS compiles one (n, m) pair per group -- (2,1), (6,3), (8,4), (12,6) -- so a variant builds in well under a
# minute and is a few MB; variants/ is git-ignored but travels with gpurun.
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p variants
python __graft_entry__.py variant "$PWD/variants/libadmm_hip_${name}.so" -DADMM_DEV_DIMS "$@" > "variants/${name}.build.log" 2>&1 || { tail -20 "variants/${name}.build.log"; exit 1; }
rm -rf "variants/libadmm_hip_${name}.so.obj"
ls -la "variants/libadmm_hip_${name}.so"

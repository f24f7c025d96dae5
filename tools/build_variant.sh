#!/bin/bash
# Build a library variant for in-run A/B (tools/ab_lib.sh, tools/pmc_ab.sh): tools/_bin/libvsg_<name>.so with extra
# -D flags; the in-tree library is not touched.   Usage: tools/build_variant.sh <name> [-DVSG_...=..] ...
set -e
name=$1; shift
cd "$(dirname "$0")/.."
C=visual_sgraphs_amd/csrc
mkdir -p tools/_bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -w -mllvm -amdgpu-mfma-vgpr-form "$@" \
  -o tools/_bin/libvsg_$name.so $C/vsg_kernels.hip $C/vsg_orb.hip $C/vsg_match.hip $C/vsg_grid.hip $C/vsg_bow.hip \
  $C/vsg_frame.hip $C/vsg_ctx.hip $C/vsg_shard.hip -ldl -lpthread
echo "built tools/_bin/libvsg_$name.so"

#!/bin/bash
# Build a named A/B variant of the library for XNACK-off devices (what the loader picks on this pool):
#   bash tools/build_variant.sh NAME [SRC_DIR] [EXTRA_FLAGS...]   ->  gymwipe_amd/lib/libgymwipe_amd_NAME.so
# SRC_DIR defaults to gymwipe_amd/csrc; pass a checkout of another revision's csrc (git worktree / git archive) to compare
# revisions on one box with tools/ab.sh.  Never the product library.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=${2:-$REPO/gymwipe_amd/csrc}; shift; shift || true
FLAGS=$(make -s -C $REPO/gymwipe_amd/csrc print-flags | sed 's/--offload-arch=gfx950/--offload-arch=gfx950:xnack-/; s#-I../../include#-I'$REPO'/include#')
SRCS="gw_api.cpp gw_plant_api.cpp plant_mfma.hip gw_grid_api.cpp grid_phy.hip gw_tables.cpp ct_step.hip ct_step_sfx.hip ct_step_dyn.hip ct_rollout_sfx.hip feedback_pack.hip gw_ctrl_api.cpp ctrl_step.hip"
cd $SRC
/opt/rocm/bin/hipcc $FLAGS "$@" -shared -o $REPO/gymwipe_amd/lib/libgymwipe_amd_$NAME.so -x hip $SRCS
echo built $REPO/gymwipe_amd/lib/libgymwipe_amd_$NAME.so

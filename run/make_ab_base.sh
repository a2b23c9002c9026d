#!/bin/bash
# builds the library of the last commit into xmask3d_amd/ab/libxm3d_hip_base.so (git-ignored; travels to the GPU box) for run/ab_*.sh
set -e
rm -rf /tmp/xm3d_base && mkdir -p /tmp/xm3d_base
git -C "$(dirname "$0")/.." archive HEAD xmask3d_amd/csrc include | tar -x -C /tmp/xm3d_base
make -C /tmp/xm3d_base/xmask3d_amd/csrc -j8 > /dev/null
mkdir -p "$(dirname "$0")/../xmask3d_amd/ab"
cp /tmp/xm3d_base/xmask3d_amd/libxm3d_hip.so "$(dirname "$0")/../xmask3d_amd/ab/libxm3d_hip_base.so"

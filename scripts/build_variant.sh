#!/bin/bash
# A/B builds of the solver kernels: compiles csrc/w8pt16.hip and csrc/loss_tail.hip (the objects that hold the step's kernels) with extra -D switches and
# links it with the product's other cached objects into ab_libs/<name>.so (git-ignored; travels to the GPU box).  Runs here (hipcc
# cross-compiles).   usage: bash scripts/build_variant.sh <name> [-DSWITCH=1 ...]      then on the GPU box:
#   python scripts/ab_time.py ab_libs/base.so ab_libs/<name>.so ; bash scripts/ab_fit.sh ab_libs/base.so ab_libs/<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python pytorch-deepfepe_amd/build.py > /dev/null
mkdir -p ab_libs/obj_$name
C=pytorch-deepfepe_amd/csrc
for f in w8pt16 loss_tail; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -mllvm -amdgpu-kernarg-preload-count=16 "$@" \
        -Iinclude -I$C -c $C/$f.hip -o ab_libs/obj_$name/$f.o &
done
wait
objs=$(ls $C/build/*.o | grep -v "/w8pt16.o\|/loss_tail.o")
hipcc --offload-arch=gfx950 -shared -fPIC $objs ab_libs/obj_$name/w8pt16.o ab_libs/obj_$name/loss_tail.o -o ab_libs/$name.so
echo ab_libs/$name.so

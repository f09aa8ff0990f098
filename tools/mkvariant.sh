#!/bin/bash
# usage: tools/mkvariant.sh name "-DFLAG=.. ..."   -> build/variants/lib_<name>.so   (all HIP translation units rebuilt with the flags)
set -e
V=/root/repo/build/variants; mkdir -p $V/obj_$1
cd /root/repo/mujoco_mpc_amd/csrc
for f in engine.hip rollout_*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -Wno-unused-value $2 -c -o $V/obj_$1/${f%.hip}.o $f &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o $V/lib_$1.so $V/obj_$1/*.o _obj/planner.o _obj/testspeed.o _obj/multi.o -lpthread
rm -rf $V/obj_$1
echo built lib_$1.so

#!/bin/bash
# usage: tools/mkvariant.sh name "-DFLAG=.. ..."   -> build/variants/lib_<name>.so
set -e
mkdir -p /root/repo/build/variants
cd /root/repo/mujoco_mpc_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -Wno-unused-value $2 -c -o /root/repo/build/variants/eng_$1.o engine.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o /root/repo/build/variants/lib_$1.so /root/repo/build/variants/eng_$1.o _obj/planner.o _obj/testspeed.o -lpthread
rm -f /root/repo/build/variants/eng_$1.o
echo built lib_$1.so

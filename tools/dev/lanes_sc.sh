#!/bin/bash
# build + run tools/dev/lanes_sc.cpp on the GPU box
set -e
cd "$(dirname "$0")/../.."
g++ -O2 -std=c++17 -I include tools/dev/lanes_sc.cpp -L ceno_amd -lceno_hip -Wl,-rpath,$PWD/ceno_amd -lpthread -o /tmp/lanes_sc
for nv in ${NVS:-7 12 16}; do /tmp/lanes_sc $nv ${M:-32} ${T:-8} ${MINT:-1}; done

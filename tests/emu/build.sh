#!/bin/sh
# builds the host lock-step emulation of the product kernels (test infrastructure only)
set -e
cd "$(dirname "$0")"
g++ -O1 -g -std=c++17 -DOSOT_EMULATION -fPIC -shared -fvisibility=hidden -Wl,-Bsymbolic -I. -I../../opensot_amd/csrc -I../../include \
    -Wno-unused-parameter emu_driver.cpp -o libosot_emu.so
# the wide-QP solver (opensot_amd/csrc/osot_qp_big.h) with a team of one thread: the same source the product runs as a 256-thread workgroup
g++ -O2 -g -std=c++17 -fPIC -shared -fvisibility=hidden -I../../opensot_amd/csrc -I../../include big_host.cpp -o libosot_big_host.so

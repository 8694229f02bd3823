#!/bin/sh
# builds the host lock-step emulation of the product kernels (test infrastructure only)
set -e
cd "$(dirname "$0")"
g++ -O1 -g -std=c++17 -DOSOT_EMULATION -fPIC -shared -fvisibility=hidden -Wl,-Bsymbolic -I. -I../../opensot_amd/csrc -I../../include \
    -Wno-unused-parameter emu_driver.cpp -o libosot_emu.so

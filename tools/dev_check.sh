#!/bin/bash
# Developer loop on the CPU box: rebuild the SIMT emulation of the solver kernels and the product library, run the SIMT tests.
# usage: bash tools/dev_check.sh [pytest -k expression]
R=$(cd $(dirname $0)/.. && pwd)
g++ -O2 -std=c++17 -fPIC -ffp-contract=off -Wno-psabi -DVIO_SIMT -I$R/include -I$R/vins-mobile_amd/csrc -I$R/tests/emul -shared \
  -o $R/tests/emul/libvio_simt.so $R/tests/emul/simt_backend.cpp 2>&1 | grep -E "error" | head
(cd $R && timeout 2400 python -m pytest tests/test_simt_backend.py -x -q ${1:+-k "$1"} 2>&1 | tail -2)
make -C $R/vins-mobile_amd/csrc -j4 2>&1 | grep -E "error|Error"
true

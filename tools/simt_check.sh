# bash tools/simt_check.sh [pytest -k expression]: the device sections of the solver on the SIMT emulator (CPU), rebuilt first
cd "$(dirname "$0")/.."
( cd tests/emul && g++ -O2 -std=c++17 -fPIC -ffp-contract=off -Wno-psabi -DVIO_SIMT -I../../include -I../../vins-mobile_amd/csrc -I. -shared -o libvio_simt.so simt_backend.cpp 2>&1 | grep -E "error" -A3 | head -30 )
timeout 1500 python -m pytest tests/test_simt_backend.py -x -q ${1:+-k "$1"} 2>&1 | tail -4

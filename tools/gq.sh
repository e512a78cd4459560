#!/bin/bash
# tools/gq.sh [tests]: rebuild the library, (SIMT check), then one GPU call: stage profile of one window + timings at 1 / 256 / 512
cd "$(dirname "$0")/.."
make -j8 -C vins-mobile_amd/csrc 2>&1 | grep -E "error|Error" -A5 | head -30
if [ "$1" = "simt" ] || [ "$2" = "simt" ]; then bash tools/simt_check.sh 2>&1 | tail -1; fi
T=""
if [ "$1" = "tests" ] || [ "$2" = "tests" ]; then T='; timeout 600 python -m pytest tests/test_backend_gpu.py tests/test_closed_loop.py -x -q -m gpu 2>&1 | grep -v "marginaliz\|release" | tail -2'; fi
timeout 1500 /usr/local/graft/bin/gpurun --timeout 900 -- "python tools/time_backend.py --prof-batch=1 1 256 512 2>&1 | grep 'path=\|stage' $T" 2>&1 | tail -6 | tr ',' '\n' | grep -v "=0(" | tr '\n' ',' | sed 's/path=single/\n/g; s/(wall[^>]*>//g; s/us\/solve[^;]*;//g'
echo

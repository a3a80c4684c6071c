#!/bin/bash
# ASan + UBSan fuzz of the blob parser on the CPU build: tools/fuzz_blob.sh [SEED] [N_PER_MODEL]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d)
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -w -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include \
    -I$R/include -I$R/infiniteexamodels.jl_amd/csrc $R/tools/fuzz_blob.cpp $R/infiniteexamodels.jl_amd/csrc/iem_api.cpp \
    $R/infiniteexamodels.jl_amd/csrc/iem_codegen.cpp -L/opt/rocm/lib -lamdhip64 -lhiprtc -ldl -Wl,-rpath,/opt/rocm/lib -o $D/fuzz_blob
python3 $R/tools/fuzz_blob_gen.py ${1:-1} ${2:-100} $D/blobs
ASAN_OPTIONS=detect_leaks=0 $D/fuzz_blob $D/blobs/*.bin 2> $D/err.log || { grep -n "ERROR\|runtime error" -A12 $D/err.log | head -40; exit 1; }
grep -c "runtime error" $D/err.log && exit 1
rm -rf $D

#!/bin/bash
# ORACLE / TEST INFRASTRUCTURE.  Builds oracle/_ref/libexl3_ref_mul1.so from the reference's own
# host-only C++ (exllamav3_ext/cpu/moe_mul1.cpp) compiled FROM WHERE IT LIES under /root/reference,
# plus our extern "C" harness.  No reference source is copied into the repo; outputs only into oracle/_ref/.
# Skips quietly when /root/reference is absent (GPU box: the prebuilt .so travels with the snapshot).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=/root/reference/exllamav3/exllamav3_ext
OUT="$HERE/_ref"
if [ ! -f "$REF/cpu/moe_mul1.cpp" ]; then
    echo "build_ref: $REF not present, skipping"; exit 0
fi
mkdir -p "$OUT"
if [ "$OUT/libexl3_ref_mul1.so" -nt "$HERE/ref_harness.cpp" ] && [ "$OUT/libexl3_ref_mul1.so" -nt "$REF/cpu/moe_mul1.cpp" ]; then
    echo "build_ref: up to date"; exit 0
fi
TORCH_DIR=$(python3 -c "import torch, os; print(os.path.dirname(torch.__file__))")
PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
g++ -O2 -std=c++17 -fPIC -shared -mavx2 -mfma -mf16c \
    -D_GLIBCXX_USE_CXX11_ABI=$(python3 -c "import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))") \
    -I"$REF" -I"$TORCH_DIR/include" -I"$TORCH_DIR/include/torch/csrc/api/include" -I"$PYINC" \
    "$HERE/ref_harness.cpp" "$REF/cpu/moe_mul1.cpp" \
    -L"$TORCH_DIR/lib" -ltorch_cpu -lc10 -Wl,-rpath,"$TORCH_DIR/lib" -lpthread \
    -o "$OUT/libexl3_ref_mul1.so"
echo "build_ref: built $OUT/libexl3_ref_mul1.so"

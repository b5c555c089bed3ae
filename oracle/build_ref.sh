#!/bin/bash
# ORACLE / TEST INFRASTRUCTURE.  Builds, FROM WHERE THE SOURCES LIE under /root/reference (no reference source is copied into the
# repo; outputs only into oracle/_ref/):
#   oracle/_ref/libexl3_ref_cuda.so  the reference's DEVICE headers for the codebooks, the trellis window readers and the KV-cache
#                                    quantizer (quant/codebook.cuh, quant/exl3_dq.cuh, cache/lmq.cuh, cache/q_cache_kernels.cuh) compiled for
#                                    the host on top of oracle/cuda_host_shim.h, behind the extern "C" harness oracle/ref_cuda_harness.cpp;
#   oracle/_ref/libexl3_ref_act.so   the reference's activation kernels (activation_kernels.cuh act_mul_kernel_h / _f: silu / gelu / relu2 / relu / silu_oai
#                                    times up, with the act_limit clamps) the same way, behind oracle/ref_act_harness.cpp;
#   oracle/_ref/libexl3_ref_mul1.so  the reference's own host-only C++ (exllamav3_ext/cpu/moe_mul1.cpp) + oracle/ref_harness.cpp.
# Skips quietly when /root/reference is absent (GPU box: the prebuilt .so travels with the snapshot).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=/root/reference/exllamav3/exllamav3_ext
OUT="$HERE/_ref"
if [ ! -f "$REF/cpu/moe_mul1.cpp" ]; then
    echo "build_ref: $REF not present, skipping"; exit 0
fi
mkdir -p "$OUT"
if [ "$OUT/libexl3_ref_cuda.so" -nt "$HERE/ref_cuda_harness.cpp" ] && [ "$OUT/libexl3_ref_cuda.so" -nt "$HERE/cuda_host_shim.h" ] \
   && [ "$OUT/libexl3_ref_cuda.so" -nt "$REF/quant/codebook.cuh" ] && [ "$OUT/libexl3_ref_cuda.so" -nt "$REF/cache/q_cache_kernels.cuh" ]; then
    echo "build_ref: libexl3_ref_cuda.so up to date"
else
    g++ -O1 -std=c++17 -fPIC -shared -Wno-attributes -I"$REF" -I"$HERE" "$HERE/ref_cuda_harness.cpp" -lpthread -o "$OUT/libexl3_ref_cuda.so"
    echo "build_ref: built $OUT/libexl3_ref_cuda.so"
fi
if [ "$OUT/libexl3_ref_act.so" -nt "$HERE/ref_act_harness.cpp" ] && [ "$OUT/libexl3_ref_act.so" -nt "$HERE/cuda_host_shim.h" ] \
   && [ "$OUT/libexl3_ref_act.so" -nt "$REF/activation_kernels.cuh" ]; then
    echo "build_ref: libexl3_ref_act.so up to date"
else
    g++ -O1 -std=c++17 -fPIC -shared -Wno-attributes -DUSE_ROCM -I"$REF" -I"$HERE" "$HERE/ref_act_harness.cpp" -lpthread -o "$OUT/libexl3_ref_act.so"
    echo "build_ref: built $OUT/libexl3_ref_act.so"
fi
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

#!/bin/bash
# diagnostics: build/alt_gn2var/libexl3_hip.so = the library with timing-only variants of the generation-2 GEMM K-loop (EXL3_GN2_VARIANT=1..N picks one; 0 = the product loop)
#   tools/r6_gn2_variants.sh "noglds" "noreads" "nobar" "noglds noreads nobar" "spread" "@noepi"      (@noepi: product loop, no global stores)
cd /root/repo; D=build/gn2_variants; mkdir -p $D; : > $D/gn2_variants.inc; n=0; tbl=""
for v in "$@"; do n=$((n+1))
  case "$v" in k64*) python tools/gen_gemm_nt3_loop.py ${v#k64} > $D/v$n.inc; echo "#define GN2_K64" >> $D/gn2_variants.inc;; esac
  if [[ "$v" == k64* ]]; then :; elif [ "$v" = "@noepi" ]; then python tools/gen_gemm_nt2_loop.py > $D/v$n.inc; echo "#define GN2_NOEPI" >> $D/gn2_variants.inc; else python tools/gen_gemm_nt2_loop.py $v > $D/v$n.inc; fi
  case "$v" in *src128*) echo "#define GN2_SRC128" >> $D/gn2_variants.inc;; esac
  case "$v" in *n128*) printf '#define GN2_TILE_N 128\n#define GN2_NB 4\n' >> $D/gn2_variants.inc;; *) printf '#define GN2_TILE_N 256\n#define GN2_NB 8\n' >> $D/gn2_variants.inc;; esac
  printf '#define GN2_NAME exl3_gemm_nt2_kernel_v%d\n#define GN2_LOOP_FILE "v%d.inc"\n#include "exl3_gemm_nt2_body.inc"\n#undef GN2_NAME\n#undef GN2_LOOP_FILE\n#undef GN2_NOEPI\n#undef GN2_SRC128\n#undef GN2_K64\n#undef GN2_TILE_N\n#undef GN2_NB\n' $n $n >> $D/gn2_variants.inc
  tbl="$tbl exl3_gemm_nt2_kernel_v$n,"; echo "variant $n: $v"
done
printf '#define GN2_NVARIANTS %d\nstatic gn2_kernel_t gn2_variant_table[] = {%s };\n' $n "$tbl" >> $D/gn2_variants.inc
python tools/alt_lib.py gn2var exl3_gemm_nt2.o -DGN2_VARIANTS -I$PWD/$D -I$PWD/exllamav3_amd/csrc

#!/bin/bash
# Builds the UNMODIFIED reference host (JiejunShi/BASAL: main.cpp, reads.cpp, refbase.cpp, param.cpp, pairs.cpp, utilities.cpp and
# align.cpp, plus its vendored samtools / gzstream) with SingleAlign::Do_Batch and PairAlign::Do_Batch provided by integration/*.inc on top of
# libbasal_amd.so -- the north star's "the C++ host stays and calls the GPU through a thin C-ABI", as a binary:
#     oracle/_ref_gpu/basal      (git-ignored like oracle/_ref/; travels to the GPU box with the snapshot)
# No reference file is edited or copied into the repository: every reference source is compiled where it lies; align.cpp and pairs.cpp get
# -DDo_Batch=Do_Batch_reference_cpu, which renames the reference's own Do_Batch functions (and their declarations as those two files see
# them), so that the symbols main.cpp calls, SingleAlign::Do_Batch and PairAlign::Do_Batch, are the ones integration/*.inc define.  Objects of the other reference
# files are shared with oracle/Makefile.ref.  Build container only (needs /root/reference).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REF=${REF:-/root/reference}
OUT=$ROOT/oracle/_ref_gpu
[ -d "$REF" ] || { echo "build_ref_with_core.sh: $REF not found (build container only)"; exit 1; }
make -s -C "$ROOT/basal_amd/csrc" -j8 all
make -s -f "$ROOT/oracle/Makefile.ref" -j8
mkdir -p "$OUT/obj"
CXXFLAGS="-DMAXHITS=1000 -DTHREAD -funroll-loops -O3 -m64 -w -I$REF -I$REF/samtools -I$REF/gzstream"
g++ $CXXFLAGS -DDo_Batch=Do_Batch_reference_cpu -c "$REF/align.cpp" -o "$OUT/obj/align_renamed.o"
g++ $CXXFLAGS -DDo_Batch=Do_Batch_reference_cpu -c "$REF/pairs.cpp" -o "$OUT/obj/pairs_renamed.o"
printf '#include "do_batch_gpu.inc"\n' > "$OUT/obj/do_batch_gpu.cpp"
printf '#include "pair_do_batch_gpu.inc"\n' > "$OUT/obj/pair_do_batch_gpu.cpp"
g++ $CXXFLAGS -std=c++11 -I"$ROOT/integration" -I"$ROOT/include" -c "$OUT/obj/do_batch_gpu.cpp" -o "$OUT/obj/do_batch_gpu.o"
g++ $CXXFLAGS -std=c++11 -I"$ROOT/integration" -I"$ROOT/include" -c "$OUT/obj/pair_do_batch_gpu.cpp" -o "$OUT/obj/pair_do_batch_gpu.o"
O=$ROOT/oracle/_ref/obj
g++ $CXXFLAGS "$OUT/obj/align_renamed.o" "$OUT/obj/pairs_renamed.o" "$OUT/obj/do_batch_gpu.o" "$OUT/obj/pair_do_batch_gpu.o" $O/refbase.o $O/main.o $O/param.o $O/reads.o $O/utilities.o $O/gzstream.o \
    $(ls $O/bam/*.o | grep -v -E "/(bam_tview|bam_plcmd|sam_view|bam_rmdup|bam_rmdupse|bam_mate|bam_stat|bam_color|bamtk|kaln|bam2bcf|bam2bcf_indel|errmod|sample|cut_target|phase|bam2depth)\.o") \
    -o "$OUT/basal" -L"$ROOT/basal_amd/lib" -lbasal_amd -Wl,-rpath,'$ORIGIN/../../basal_amd/lib' -lpthread -lz -lm
rm -rf "$OUT/obj"
echo "built $OUT/basal"

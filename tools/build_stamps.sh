#!/bin/bash
# The diagnostic build (-DZK_STAMPS: in-kernel s_memtime per phase of stream_pass0_kernel / dedupe_kernel / dedupe2_kernel) as a second
# library, build/libzotk_stamps.so, without touching the product build: tools/p0_phases.py reads it through ZOTK_LIB.
set -e
cd "$(dirname "$0")/.."
T=/tmp/zotk_stamps_build
rm -rf $T && mkdir -p $T/zotmer_amd $T/include build
cp -r zotmer_amd/csrc $T/zotmer_amd/ && cp include/zotk.h $T/include/
rm -f $T/zotmer_amd/csrc/*.o
make -C $T/zotmer_amd/csrc -j8 CXXFLAGS_EXTRA=-DZK_STAMPS 2>&1 | grep -v "^/opt\|warning: ignoring" | tail -3
cp $T/zotmer_amd/libzotk.so build/libzotk_stamps.so
ls -la build/libzotk_stamps.so

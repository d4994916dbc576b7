#!/bin/sh
# Build the reference's OWN V0 (namespace v0, /root/reference/core.cu:11-54) into
# oracle/_ref/libv0ref.so, compiling the lines where they lie: they are streamed
# from the read-only reference into g++'s stdin, so no reference source is ever
# written into this repository (oracle/_ref/ holds the .so only and is
# git-ignored; it still travels to the GPU box with the snapshot).
#
# V0 is self-contained host C++ (malloc + INFINITY), so a <stdlib.h>/<math.h>
# prelude is all it needs; the CUDA parts of core.cu are not buildable here (no
# nvcc / CUDA headers) and are not attempted.  Flags: the reference's README
# build line has no -O and no -march (README.md:20); results are identical at
# -O0 and -O2 as long as fp contraction is off, and -O2 is the kinder CPU
# baseline, so that is what is built.
#
# The appended extern "C" wrapper is ours (a plain C symbol for ctypes).
set -e
cd "$(dirname "$0")"
REF=${NNS_REFERENCE_DIR:-/root/reference}
if [ ! -r "$REF/core.cu" ]; then
    echo "oracle/build_ref.sh: $REF/core.cu not present - skipping reference build (using prebuilt _ref if any)"
    exit 0
fi
mkdir -p _ref
{
    printf '#include <stdlib.h>\n#include <math.h>\n'
    sed -n '11,54p' "$REF/core.cu"
    printf '\nextern "C" void v0_ref_cudaCall(int k, int m, int n, float *s, float *r, int **res)\n{ v0::cudaCall(k, m, n, s, r, res); }\n'
} | g++ -x c++ -O2 -ffp-contract=off -fno-fast-math -fPIC -shared -o _ref/libv0ref.so -
echo "oracle/build_ref.sh: built _ref/libv0ref.so from $REF/core.cu:11-54"

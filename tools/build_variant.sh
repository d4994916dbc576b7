#!/bin/sh
# build a variant library: tools/build_variant.sh <name> [FILTER="flags"] [EXACT="flags"]
# (plain extra args are filter_mfma flags).  -DNNS_DIAG among the filter flags is also passed to
# nns_api.hip (the NNS_DIAG_FILTER_ONLY switch lives there).
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=/tmp/nnsvar_$NAME
FILTERF="-fno-honor-nans"; EXACTF="-fno-slp-vectorize"; APIF=""; APIX=""
for a in "$@"; do
  case "$a" in
    EXACT=*) EXACTF="$EXACTF ${a#EXACT=}";;
    FILTER=*) FILTERF="$FILTERF ${a#FILTER=}";;
    API=*) APIX="$APIX ${a#API=}";;
    *) FILTERF="$FILTERF $a";;
  esac
done
case "$FILTERF" in *-DNNS_DIAG*) APIF="-DNNS_DIAG";; esac
rm -rf $T && mkdir -p $T/nns-cuda_amd && cp -r $ROOT/include $T/ && cp -r $ROOT/nns-cuda_amd/csrc $T/nns-cuda_amd/ && rm -f $T/nns-cuda_amd/csrc/*.o
make -C $T/nns-cuda_amd/csrc -j6 $T/nns-cuda_amd/libnns_mi355x.so FLAGS_filter_mfma="$FILTERF" FLAGS_exact_kernels="$EXACTF" FLAGS_nns_api="$APIF $APIX" > $T/build.log 2>&1 || { tail -20 $T/build.log; exit 1; }
cp $T/nns-cuda_amd/libnns_mi355x.so $ROOT/nns-cuda_amd/libnns_var_$NAME.so
echo built libnns_var_$NAME.so

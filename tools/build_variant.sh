#!/bin/sh
# build a filter variant library: tools/build_variant.sh <name> <extra hipcc flags...>
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=/tmp/nnsvar_$NAME
rm -rf $T && mkdir -p $T/nns-cuda_amd && cp -r $ROOT/include $T/ && cp -r $ROOT/nns-cuda_amd/csrc $T/nns-cuda_amd/ && rm -f $T/nns-cuda_amd/csrc/*.o
make -C $T/nns-cuda_amd/csrc -j6 $T/nns-cuda_amd/libnns_mi355x.so FLAGS_filter_mfma="-fno-honor-nans $*" > $T/build.log 2>&1 || { tail -20 $T/build.log; exit 1; }
cp $T/nns-cuda_amd/libnns_mi355x.so $ROOT/nns-cuda_amd/libnns_var_$NAME.so
echo built libnns_var_$NAME.so

#!/bin/bash
# Build a second copy of libboa_hip.so with extra compiler flags (kernel A/B experiments on ONE GPU box: run-to-run /
# box-to-box spread is +-5 %, so variants are compared inside the same gpurun call):
#   tools/build_alt.sh alt1 -DWS_DEFER_EPILOGUE=0      -> body-and-organ-analysis_amd/boa_hip/libboa_hip_alt1.so
#   BOA_HIP_LIB=$PWD/body-and-organ-analysis_amd/boa_hip/libboa_hip_alt1.so python bench.py ...
set -e
NAME=$1; shift
PKG=$(cd "$(dirname "$0")/../body-and-organ-analysis_amd" && pwd)
OBJ=/tmp/boa_alt_$NAME; mkdir -p $OBJ
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$PKG/../include -I$PKG/csrc -Wno-unused-result -Wno-unused-value -Wno-pass-failed $*"
for f in $PKG/csrc/*.hip; do
  o=$OBJ/$(basename ${f%.hip}).o
  hdr_new=0; for h in $PKG/csrc/*.h $PKG/../include/*.h; do [ -f $o ] && [ $h -nt $o ] && hdr_new=1; done   # (a changed header rebuilds everything: a stale object with an old struct layout links fine and runs wrong)
  if [ ! -f $o ] || [ $f -nt $o ] || [ $hdr_new = 1 ] || [ "$(cat $OBJ/.flags 2>/dev/null)" != "$*" ]; then /opt/rocm/bin/hipcc $FLAGS -c $f -o $o & fi
done
wait
echo "$*" > $OBJ/.flags
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $PKG/boa_hip/libboa_hip_$NAME.so $OBJ/*.o
ls -la $PKG/boa_hip/libboa_hip_$NAME.so

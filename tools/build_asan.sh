#!/bin/bash
# Host-side AddressSanitizer build of libboa_hip (device code is compiled as usual: ASan for gfx950 needs xnack+).
# Usage (on the GPU box, from the repo root):
#   bash tools/build_asan.sh && RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1) && \
#   BOA_HIP_LIB=$PWD/body-and-organ-analysis_amd/boa_hip/libboa_hip_asan.so LD_PRELOAD=$RT \
#   ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:protect_shadow_gap=0 python -m pytest tests -m gpu -x -q
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/body-and-organ-analysis_amd/csrc
OUT=${TMPDIR:-/tmp}/boa_asan
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -I$ROOT/include -I$SRC -Wno-pass-failed -Wno-unused-value -Wno-option-ignored -fsanitize=address -fno-omit-frame-pointer"
pids=""
for f in api seg conv conv_ws net net_f32 agg resample morph diag; do
  /opt/rocm/bin/hipcc $FLAGS -c $SRC/$f.hip -o $OUT/$f.o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -shared-libasan -o $ROOT/body-and-organ-analysis_amd/boa_hip/libboa_hip_asan.so $OUT/*.o
ls -la $ROOT/body-and-organ-analysis_amd/boa_hip/libboa_hip_asan.so

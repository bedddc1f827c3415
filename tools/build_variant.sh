# Build a variant of the library for same-box A/B measurements: tools/build_variant.sh <name> <extra hipcc flags...>
# -> gpurun_variants/libkzg_<name>.so (travels with the gpurun snapshot; select it with KZG_LIB_PATH)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
B=/tmp/kzg_variant_$NAME; mkdir -p $B $ROOT/gpurun_variants
for f in msm ntt poly lagrange srs g1fft capi blobstream multi ubench; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -c $ROOT/rust-kzg-bn254_amd/csrc/$f.hip -o $B/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/gpurun_variants/libkzg_$NAME.so $B/*.o
ls -la $ROOT/gpurun_variants/libkzg_$NAME.so

#!/bin/bash
# A/B builds: tools/build_variant.sh NAME "-DAPK_PRIO_SORT=3 ..."  ->  algoplonk_amd/libapk_NAME.so (use with APK_LIB / tools/ab_bench.sh)
# Only the two GPU backends are rebuilt with the extra flags; the host objects of the regular build are linked as they are.
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../algoplonk_amd/csrc"
B=/tmp/apk_variant_$NAME; mkdir -p $B
CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unknown-pragmas -Wno-unused-result -ffp-contract=off $FLAGS"
/opt/rocm/bin/hipcc $CXXFLAGS -c backend_bn254.hip -o $B/backend_bn254.o &
if [ -z "$BN254_ONLY" ]; then /opt/rocm/bin/hipcc $CXXFLAGS -c backend_bls12381.hip -o $B/backend_bls12381.o & else cp backend_bls12381.o $B/; fi
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libapk_$NAME.so $B/backend_bn254.o $B/backend_bls12381.o apk_api.o verify_api.o comm.o -lpthread -ldl
ls -la ../libapk_$NAME.so

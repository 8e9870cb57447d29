#!/bin/bash
# Build-time experiment: link a library whose ensemble kernels were compiled with extra flags (scratch_variants/ens_X.o)
# and run a command with it in place of libp7x.so (on the GPU box's scratch copy only).
set -e
if [ -z "$GRAFT_REPO_ROOT" ]; then echo "ens_variant.sh replaces pyhmmer_amd/libp7x.so: run it through gpurun" >&2; exit 2; fi
cd "$GRAFT_REPO_ROOT"
obj=$1; shift; shift
objs=$(ls pyhmmer_amd/csrc/build/*.o | grep -v p7x_ensemble.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pyhmmer_amd/libp7x.so $objs $obj -lpthread
"$@"

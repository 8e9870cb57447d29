#!/bin/bash
# Build-time experiment: a library whose p7x_pipeline.hip was compiled with extra flags, in place of libp7x.so (GPU box only)
set -e
if [ -z "$GRAFT_REPO_ROOT" ]; then echo "run through gpurun" >&2; exit 2; fi
cd "$GRAFT_REPO_ROOT"
obj=$1; shift; shift
objs=$(ls pyhmmer_amd/csrc/build/*.o | grep -v p7x_pipeline.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pyhmmer_amd/libp7x.so $objs $obj -lpthread
"$@"

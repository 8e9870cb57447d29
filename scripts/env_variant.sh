#!/bin/bash
# Build-time experiment: link a library whose envelope kernel is compiled with extra flags and run a command with it in
# place of libp7x.so (on the GPU box's scratch copy).  Usage: scripts/env_variant.sh scratch_variants/env_X.o -- cmd...
set -e
if [ -z "$GRAFT_REPO_ROOT" ] && [ -z "$P7X_VARIANT_HERE" ]; then
  echo "env_variant.sh replaces pyhmmer_amd/libp7x.so: run it on the GPU box's scratch copy (gpurun), or set P7X_VARIANT_HERE=1" >&2
  exit 2
fi
cd "${GRAFT_REPO_ROOT:-/root/repo}"
obj=$1; shift; shift
objs=$(ls pyhmmer_amd/csrc/build/*.o | grep -v p7x_envelope.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pyhmmer_amd/libp7x.so $objs $obj -lpthread
"$@"

#!/bin/bash
# Build-time experiment: link a library in which one object (csrc/build/<name>.o) is replaced by a variant compiled with
# other flags, in place of libp7x.so, and run a command with it (on the GPU box's scratch copy only).
# usage: obj_variant.sh <name, e.g. p7x_vitpk.hip> <variant.o | -> -- command...      ("-": the tree's own object again)
set -e
if [ -z "$GRAFT_REPO_ROOT" ]; then echo "obj_variant.sh replaces pyhmmer_amd/libp7x.so: run it through gpurun" >&2; exit 2; fi
cd "$GRAFT_REPO_ROOT"
name=$1; obj=$2; shift; shift; shift
if [ "$obj" = "-" ]; then obj=pyhmmer_amd/csrc/build/$name.o; fi
objs=$(ls pyhmmer_amd/csrc/build/*.o | grep -v "/$name.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pyhmmer_amd/libp7x.so $objs $obj -lpthread
"$@"

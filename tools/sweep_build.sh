#!/bin/bash
# usage: tools/sweep_build.sh "<EXTRA_HIPFLAGS>" -- rebuilds libhmrm.so with extra -D flags (GPU box or here)
set -e
cd "$(dirname "$0")/../heightmap-ray-marcher_amd/csrc"
rm -f _build/render.o _build/render_fast.o _build/api.o
make -s -j16 EXTRA_HIPFLAGS="$1" >/dev/null

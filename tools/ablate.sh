#!/bin/bash
# development: build timing-ablation variants of the attention kernel into tools/ablate/libaid_abl<N>.so
set -e
cd "$(dirname "$0")/../attention-interpolation-diffusion_amd/csrc"
mkdir -p ../../tools/ablate
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -mllvm -amdgpu-mfma-vgpr-form=1 -ffinite-math-only"
for n in "$@"; do
  /opt/rocm/bin/hipcc $F -DAID_ABL=$n -c aid_attn.hip -o /tmp/aid_attn_abl$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC aid_gemm.o /tmp/aid_attn_abl$n.o aid_abi.o -o ../../tools/ablate/libaid_abl$n.so
done

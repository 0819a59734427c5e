#!/bin/bash
# Build tools/probes/gemm_ladder (cross-compiles here; runs on the MI355X box): the harness + the two wide-tile kernels, tuning arms compiled in.
set -e
cd "$(dirname "$0")/../.."
CS=dnn-based-speech-enhancement-in-the-frequency-domain_amd/csrc
O=tools/probes/_obj; mkdir -p $O
F="--offload-arch=gfx950 -O3 -std=c++17 -DSEFD_TUNING -I$CS"
/opt/rocm/bin/hipcc $F -c -o $O/cgemm256.o $CS/cgemm256.hip &
/opt/rocm/bin/hipcc $F -c -o $O/cgemm8p.o tools/probes/ladder/cgemm8p.hip &
/opt/rocm/bin/hipcc $F -c -o $O/ladder.o tools/probes/gemm_ladder.hip &
/opt/rocm/bin/hipcc $F -c -o $O/cgemm128.o tools/probes/ladder/cgemm128.hip &
/opt/rocm/bin/hipcc $F -c -o $O/rungemm.o $CS/rungemm.hip &
/opt/rocm/bin/hipcc $F -c -o $O/thin.o $CS/thin.hip &
/opt/rocm/bin/hipcc $F -c -o $O/slabgemm.o $CS/slabgemm.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -o tools/probes/gemm_ladder $O/ladder.o $O/cgemm256.o $O/cgemm8p.o $O/cgemm128.o $O/rungemm.o $O/thin.o $O/slabgemm.o
ls -la tools/probes/gemm_ladder

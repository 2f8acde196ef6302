#!/bin/bash
# A scripted model in a libtorch C++ host (examples/torchscript_host.cpp), end to end on the GPU box:
#   1. python (no GPU use): script the C3 model, save it, write a batch of frames
#   2. the C++ host: dlopen libmolann_torch.so, torch::jit::load, forward + forces -> out.bin
#   3. python (GPU): the eager molann_amd model on the same frames, compared with out.bin
# Each step is its own process started by this shell.  Usage: gpurun -- bash tools/check_torchscript_host.sh
set -euo pipefail
cd "$(dirname "$0")/.."
D=gpurun_out/ts_host; mkdir -p $D
N=${N:-5000}
python3 tools/ts_host_io.py prepare $D $N
./examples/torchscript_host molann_amd/csrc/libmolann_torch.so $D/model.pt $D/frames.bin $N 22 $D/out.bin --forces
python3 tools/ts_host_io.py check $D $N

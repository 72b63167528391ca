#!/bin/bash
# ISA listing + register / scratch summary of one kernel source (developer tool): bash tools/isa.sh team_stream [pattern]
R=$(cd "$(dirname "$0")/.." && pwd)
src=${1:-team_stream}
mkdir -p $R/gpurun_out
(cd $R/epik_amd/csrc && hipcc -O3 -std=c++20 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wextra $EXTRA -I$R/include -I. -S --cuda-device-only -o $R/gpurun_out/$src.s $src.hip 2>&1 | grep -v "hip-link" | head -${ERRLINES:-30})
python3 $R/tools/kernel_regs.py $R/gpurun_out/$src.s "${2:-}"

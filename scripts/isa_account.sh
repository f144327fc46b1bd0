#!/bin/bash
# static instruction account of the headline kernel by phase (see scripts/isa_account.py)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math --cuda-device-only -gline-tables-only -S scripts/ubench/headline_isa.hip -o /tmp/headline.s 2>/dev/null
python scripts/isa_account.py /tmp/headline.s

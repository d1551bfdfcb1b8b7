#!/bin/bash
# Registers / scratch / LDS / occupancy of every kernel of the library as the compiler reports them
# (hipcc -Rpass-analysis=kernel-resource-usage).   bash profiles/kernel_resources.sh [-DKNOB=value ...]
cd "$(dirname "$0")/../vargeno_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Rpass-analysis=kernel-resource-usage "$@" -c -o /tmp/vg_res.o vargeno_hip.hip 2>&1 |
	sed -n 's/.*remark: *//p' | sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
	awk -F': ' '/^Function Name/ {n=$2} /^VGPRs:/ {v=$2} /^ScratchSize/ {s=$2} /^Occupancy/ {o=$2} /^SGPRs Spill/ {ss=$2} /^LDS Size/ {printf "%-100s vgpr %-4s scratch %-5s sgpr-spill %-4s lds %-6s waves/SIMD %s\n", n, v, s, ss, $2, o}' | sort -u

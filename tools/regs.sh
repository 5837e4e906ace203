#!/bin/bash
# register / scratch usage of one translation unit: tools/regs.sh wino_dw [kernel-name-filter]
cd /root/repo/neuralrgbd_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c -I ../../include $1.hip -o /tmp/$1.regs.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep "error\|Function Name\|VGPRs:\|VGPRs Spill\|ScratchSize" | sed 's/\[-Rpass.*//; s/remark: .*hip:[0-9]*:[0-9]*://' | grep -A3 "${2:-Function}\|error"

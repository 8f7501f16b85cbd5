#!/bin/bash
# Development build of csrc/ksvd_block.hip: only the config-2 instantiation of bksvd_step_kernel (-DLYS_BK_DEV), with the
# compiler's per-kernel resource report.  usage: tools/bk_dev_build.sh [link]   ("link" also relinks liblyssa_hip.so with it)
set -e
cd "$(dirname "$0")/.."
O=lyssandra_amd/build/ksvd_block.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable \
  -Wno-unused-but-set-variable -DLYS_BK_DEV=1 $BK_EXTRA -Rpass-analysis=kernel-resource-usage -c lyssandra_amd/csrc/ksvd_block.hip -o $O 2>&1 |
  grep -A9 "Function Name: _ZN3lys17bksvd_step_kernelILi1ELi3ELi1ELi64ELb1" | grep "VGPRs\|Scratch\|error" | sed 's/.*remark: [^ ]* *//;s/ \[-Rpass.*//'
if [ "$1" = link ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lyssandra_amd/liblyssa_hip.so lyssandra_amd/build/*.o
  echo linked
fi
